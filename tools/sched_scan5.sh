#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out/sched
python -m pytest tests/test_scheduler_gpu.py -x -q > gpurun_out/sched/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/sched/pytest.log
tail -3 gpurun_out/sched/pytest.log
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-grouped"
run() { name=$1; shift; timeout 600 $B "$@" > gpurun_out/sched/$name.json 2> gpurun_out/sched/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/sched/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['ms_per_call'], d['config'].get('decode_scheduler'))
except Exception as e: print('ERR', e)
PY
)"; }
run c1_l3 --sched-lanes 3 --sched-slots 192
run c1_l4 --sched-lanes 4 --sched-slots 256
run c2_l4 --preset c2 --sched-lanes 4 --sched-slots 256
run c2_l4_s4 --preset c2 --sched-lanes 4 --sched-slots 256 --streams 4
run c2_l6 --preset c2 --sched-lanes 6 --sched-slots 384
run c3_l3 --preset c3 --sched-lanes 3 --sched-slots 192
run c4_l3 --preset c4 --sched-lanes 3 --sched-slots 192
