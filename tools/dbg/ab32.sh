#!/bin/bash
# A/B of library builds on configs[3] (32 links) and the 16-link streamed kernels: one bench line each, same box.
#   bash tools/dbg/ab32.sh out.txt lib1.so lib2.so ...
out=$1; shift
: > $out
for lib in "$@"; do
  for cfg in "--links 32 --steps 20 --warmup 5" "--streamed-rows --steps 30 --warmup 5"; do
    SNK_LIB=$PWD/$lib python bench.py $cfg --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s %-42s %9.0f env-steps/s  kernel %.3f ms' % ('$lib', '$cfg', d['value'], d['roofline']['kernel_ms']))" >> $out
  done
done
cat $out
