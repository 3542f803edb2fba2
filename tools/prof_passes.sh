#!/bin/bash
# rocprofv3 passes of bench.py for one configuration, each in its own run (gpurun refuses --pmc combined with the
# trace domains): kernel trace + stats, FETCH_SIZE, WRITE_SIZE and, with "sq" as the second argument, the SQ set.
#   bash tools/prof_passes.sh gpurun_out/prof_c32 [sq] -- --links 32 --steps 6 --warmup 2
# `--profile`: W + K launches of the step kernel and nothing else (no CPU baseline, no variants, no histogram pass, no
# second timed region), so the trace's launches W .. W + K - 1 are the timed region and summarize_prof.py averages those.
# then, back in the build container:  python tools/summarize_prof.py gpurun_out/prof_c32 r02_c32 --links 32 ...
set -e
out=$1; shift
sq=0
if [ "$1" = "sq" ]; then sq=1; shift; fi
[ "$1" = "--" ] && shift
mkdir -p "$out"
rocprofv3 --kernel-trace --stats -d "$out/trace" --output-format csv -- python3 bench.py "$@" --profile > "$out/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" --output-format csv -- python3 bench.py "$@" --profile > "$out/f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" --output-format csv -- python3 bench.py "$@" --profile > "$out/w.log" 2>&1
if [ $sq = 1 ]; then
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES \
    -d "$out/pmc_sq" --output-format csv -- python3 bench.py "$@" --profile > "$out/sq.log" 2>&1
fi
echo "profiled: $out"
