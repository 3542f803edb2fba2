#!/bin/bash
# FETCH_SIZE of the 32-link step kernel for the library named in SNK_LIB (default: the product's)
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/fetch_$1; mkdir -p $out
rocprofv3 --pmc FETCH_SIZE -d $out --output-format csv -- python3 bench.py --links 32 --steps 6 --warmup 2 --no-cpu-baseline --no-variants > $out/f.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
v = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "env_step_sched_kernel<32" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            v.append(float(r["Counter_Value"]))
v = v[2:]
print("FETCH_SIZE per launch: %.1f GB x 2 (gfx950 correction) = %.1f GB read" % (sum(v) / len(v) * 1024 / 1e9, 2 * sum(v) / len(v) * 1024 / 1e9))
PY
