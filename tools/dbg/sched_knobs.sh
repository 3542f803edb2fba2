#!/bin/bash
# The scheduler's two knobs on the current build: one bench line per (SNK_QUANTUM, SNK_HYST), headline configuration.
#   bash tools/dbg/sched_knobs.sh out.txt
out=$1; : > $out
for q in 1 2; do for h in 1 2 3 4 5; do
  SNK_QUANTUM=$q SNK_HYST=$h python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('quantum $q hyst $h  %9.0f env-steps/s  kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']))" >> $out
done; done
cat $out
