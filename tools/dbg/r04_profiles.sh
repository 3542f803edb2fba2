#!/bin/bash
# Round 4's rocprofv3 evidence, on the GPU box:  bash tools/dbg/r04_profiles.sh
# Three configurations (each pass its own run, tools/prof_passes.sh) and, for VERDICT r3 item 8, the WRITE_SIZE of the
# 16-link kernel with its suspected writers switched off one at a time: SNK_QUANTUM=0 (the unscheduled kernel: no
# hand-offs at all) and SNK_HYST=64 (scheduled, but no env-step ever changes waves).
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P=gpurun_out/prof4
bash tools/prof_passes.sh $P/c16 sq -- --steps 10 --warmup 2 > $P.c16.log 2>&1
echo "c16 done"
bash tools/prof_passes.sh $P/c16_fric -- --friction-seed 1 --steps 10 --warmup 2 > $P.c16f.log 2>&1
echo "c16_fric done"
bash tools/prof_passes.sh $P/c32 sq -- --links 32 --steps 6 --warmup 2 > $P.c32.log 2>&1
echo "c32 done"
for v in "SNK_QUANTUM=0" "SNK_HYST=64"; do
  d=$P/w16_$(echo $v | tr '=' '_')
  mkdir -p $d
  export $v
  rocprofv3 --pmc WRITE_SIZE -d $d/pmc_write --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants > $d/w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $d/pmc_fetch --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants > $d/f.log 2>&1
  unset ${v%%=*}
  echo "$v done"
done
