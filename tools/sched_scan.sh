#!/bin/bash
# Decode-scheduler sweep on one GPU box: parity tests first, then the headline bench under different scheduler settings.
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
run sched_default
run nosched --no-scheduler
run sched_s1 --streams 1
run sched_s2 --streams 2
run sched_s4 --streams 4
run sched_r24 --sched-rows 24
run sched_r16 --sched-rows 16
run sched_stride4 --sched-stride 4
run sched_stride16 --sched-stride 16
run sched_nostagger --no-stagger
