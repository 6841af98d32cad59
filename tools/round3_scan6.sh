#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out/s6
python -m pytest tests/test_scheduler_gpu.py tests/test_harness_ingest_gpu.py tests/test_bf16_emulation_gpu.py -x -q -k "not fullsize_4s_clip_bf16_engine" > gpurun_out/s6/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s6/pytest.log
python -m pytest tests/test_kernels_gpu.py -x -q -k "stem" > gpurun_out/s6/pytest_stem.log 2>&1; echo "pytest stem rc=$?"; tail -3 gpurun_out/s6/pytest_stem.log
timeout 900 python tools/harness_throughput.py 96 > gpurun_out/s6/harness.log 2>&1; echo "harness rc=$?"; tail -8 gpurun_out/s6/harness.log
B="python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --no-grouped"
run() { name=$1; shift; timeout 600 $B "$@" > gpurun_out/s6/$name.json 2> gpurun_out/s6/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/s6/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['ms_per_call'])
except Exception as e: print('ERR', e)
PY
)"; }
run beam15_s3 --beams 15
run beam15_s4 --beams 15 --streams 4
run beam15_s5 --beams 15 --streams 5
export OAVSR_SKINNY_MAX_ROWS=128
run beam15_sk128_s3 --beams 15
run beam15_sk128_s5 --beams 15 --streams 5
unset OAVSR_SKINNY_MAX_ROWS
