#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out/sched
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-grouped"
run() { name=$1; shift; timeout 600 $B "$@" > gpurun_out/sched/$name.json 2> gpurun_out/sched/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/sched/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['ms_per_call'], d['config'].get('decode_scheduler'))
except Exception as e: print('ERR', e)
PY
)"; }
run ns_s2 --no-stagger --streams 2
run ns_s3 --no-stagger --streams 3
run ns_s4 --no-stagger --streams 4
run ns_s5 --no-stagger --streams 5
run ns_s3_stride4 --no-stagger --streams 3 --sched-stride 4
run ns_s3_stride6 --no-stagger --streams 3 --sched-stride 6
run ns_s3_slots128 --no-stagger --streams 3 --sched-slots 128
run ns_s2_b16 --no-stagger --streams 2 --batch 16
run c2 --preset c2
run c2_ns --preset c2 --no-stagger
