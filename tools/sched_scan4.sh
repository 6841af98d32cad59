#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out/sched
python -m pytest tests/test_scheduler_gpu.py -x -q > gpurun_out/sched/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/sched/pytest.log
tail -5 gpurun_out/sched/pytest.log
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-grouped"
run() { name=$1; shift; timeout 600 $B "$@" > gpurun_out/sched/$name.json 2> gpurun_out/sched/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/sched/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['ms_per_call'], d['config'].get('decode_scheduler'))
except Exception as e: print('ERR', e)
PY
)"; }
run c1_l1
run c1_l2 --sched-lanes 2 --sched-slots 128
run c1_l2_s4 --sched-lanes 2 --sched-slots 128 --streams 4
run c2_l1 --preset c2
run c2_l2 --preset c2 --sched-lanes 2 --sched-slots 128
run c2_l3 --preset c2 --sched-lanes 3 --sched-slots 192
run c2_l2_r16 --preset c2 --sched-lanes 2 --sched-slots 128 --sched-rows 16
run c4_l2 --preset c4 --sched-lanes 2 --sched-slots 128
