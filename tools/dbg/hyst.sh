#!/bin/bash
# scheduler hand-off hysteresis x slice length: speed of the default config
for qh in "1 1" "1 3" "1 4" "1 5" "1 6" "1 8" "2 4" "3 3"; do set -- $qh
  echo -n "quantum $1 hyst $2: "
  SNK_QUANTUM=$1 SNK_HYST=$2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel_ms'])"
done
