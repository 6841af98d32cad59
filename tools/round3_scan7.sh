#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out/s7
B="python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --no-grouped"
run() { name=$1; shift; timeout 600 $B "$@" > gpurun_out/s7/$name.json 2> gpurun_out/s7/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/s7/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['ms_per_call'])
except Exception as e: print('ERR', e)
PY
)"; }
run beam15_s3 --beams 15
run beam15_s1 --beams 15 --streams 1
export OAVSR_BEAM_SKINNY=1
run beam15_sk_s3 --beams 15
run beam15_sk_s1 --beams 15 --streams 1
unset OAVSR_BEAM_SKINNY
bash tools/beam_trace.sh > gpurun_out/s7/beam_trace.txt 2>&1
head -40 gpurun_out/s7/beam_trace.txt
