#!/bin/bash
# The three configurations' rocprofv3 passes again on the round's final build (after the ring split and the in-place
# substep's fix):  bash tools/dbg/r04_profiles_final.sh      -> gpurun_out/prof4f/{c16,c16_fric,c32}
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P=gpurun_out/prof4f
mkdir -p $P
bash tools/prof_passes.sh $P/c16 sq -- --steps 10 --warmup 2 > $P.c16.log 2>&1
echo "c16 done"
bash tools/prof_passes.sh $P/c16_fric -- --friction-seed 1 --steps 10 --warmup 2 > $P.c16f.log 2>&1
echo "c16_fric done"
bash tools/prof_passes.sh $P/c32 sq -- --links 32 --steps 6 --warmup 2 > $P.c32.log 2>&1
echo "c32 done"
