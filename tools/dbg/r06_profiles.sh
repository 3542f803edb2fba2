#!/bin/bash
# The three configurations' rocprofv3 passes on a round-6 build, each with the W / K the default bench.py run times it
# with (headline 10 + 100; variants.configs4_c16_fric 10 + 50; variants.configs3_c32 6 + 20), so that the trace's timed
# launches carry the same substep counts as the bench line's:
#   bash tools/dbg/r06_profiles.sh  -> gpurun_out/prof6/{c16,c16_fric,c32}
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P=gpurun_out/prof6
mkdir -p $P
bash tools/prof_passes.sh $P/c16 sq -- --steps 100 --warmup 10 > $P.c16.log 2>&1
echo "c16 done"
bash tools/prof_passes.sh $P/c16_fric -- --friction-seed 1 --steps 50 --warmup 10 > $P.c16f.log 2>&1
echo "c16_fric done"
bash tools/prof_passes.sh $P/c32 sq -- --links 32 --steps 20 --warmup 6 > $P.c32.log 2>&1
echo "c32 done"
