#!/bin/bash
# one line per configuration: value, kernel ms, substeps per env-step, overflow
run() { echo -n "$1: "; shift; python bench.py "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'env-steps/s, kernel ms', round(d['roofline']['kernel_ms'],3), 'substeps/env-step', round(d['mean_substeps_per_env_step'],3), 'overflow', d['config']['contact_overflow'], 'cpu', (round(d['cpu_baseline']['one_thread']['value'],1), round(d['cpu_baseline']['value'],1), d['cpu_baseline']['cores']) if d['cpu_baseline'] else None)"; }
run "default c16" --steps 100 --warmup 10 --no-variants
run "c16 friction seed 1" --steps 100 --warmup 10 --no-variants --friction-seed 1 --no-cpu-baseline
run "c16 round1 model" --steps 100 --warmup 10 --no-variants --hull-sides 0 --contact-model 0 --no-cpu-baseline
run "c16 warm start" --steps 100 --warmup 10 --no-variants --warm-start 1 --no-cpu-baseline
run "c16 static box at 0.1" --steps 40 --warmup 5 --no-variants --obstacle 0.1 --no-cpu-baseline
run "c16 free box at 0.1" --steps 40 --warmup 5 --no-variants --obstacle 0.1 --obstacle-free --no-cpu-baseline
run "c16 streamed rows" --steps 40 --warmup 5 --no-variants --streamed-rows --no-cpu-baseline
run "c32" --links 32 --steps 20 --warmup 4 --cpu-steps 400
run "c32 no self-collision" --links 32 --steps 20 --warmup 4 --self-collision 0 --no-cpu-baseline
run "c16 policy" --steps 40 --warmup 5 --policy --no-cpu-baseline
python tools/host_api_rate.py
