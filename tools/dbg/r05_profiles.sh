#!/bin/bash
# The three configurations' rocprofv3 passes on the final build of round 5:  bash tools/dbg/r05_profiles.sh  -> gpurun_out/prof5/{c16,c16_fric,c32}
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P=gpurun_out/prof5
mkdir -p $P
bash tools/prof_passes.sh $P/c16 sq -- --steps 10 --warmup 2 > $P.c16.log 2>&1
echo "c16 done"
bash tools/prof_passes.sh $P/c16_fric -- --friction-seed 1 --steps 10 --warmup 2 > $P.c16f.log 2>&1
echo "c16_fric done"
bash tools/prof_passes.sh $P/c32 sq -- --links 32 --steps 6 --warmup 2 > $P.c32.log 2>&1
echo "c32 done"
