#!/bin/bash
# Instruction-cache and instruction-fetch counters of the 16-link step kernel (is the 300-KB kernel starved for code?)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/icache; mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1 || true
grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_WAIT_ANY\|SQ_WAIT_IFETCH" $out/counters.txt | sort -u | tr '\n' ' ' > $out/avail.txt; cat $out/avail.txt; echo
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES -d $out/p1 --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants > $out/p1.log 2>&1 || tail -3 $out/p1.log
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $out/p2 --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants > $out/p2.log 2>&1 || tail -3 $out/p2.log
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    d = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/icache/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "env_step_sched_kernel<16" in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(d.items()):
        print(p, k, "per launch %.4g (%d launches)" % (sum(v[2:]) / max(1, len(v[2:])), len(v)))
PY
