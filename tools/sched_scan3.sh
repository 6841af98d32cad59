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
run c1
run c2 --preset c2
export OAVSR_SCHED_PRIORITY=-1
run c1_prio
run c2_prio --preset c2
unset OAVSR_SCHED_PRIORITY
run c2_nosched --preset c2 --no-scheduler
run c2_b8 --preset c2 --batch 8
run c3 --preset c3
run c4 --preset c4
run c3_nosched --preset c3 --no-scheduler
