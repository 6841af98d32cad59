#!/bin/bash
# Sweeps of the decode scheduler on one GPU box (results: one line per run; see profiles/r03_sched_scan.txt for the round-3 collection).
#   usage: sched_scan.sh [preset ...]        e.g.  sched_scan.sh c1 c2
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
for P in "${@:-c1}"; do
  run ${P}_default --preset $P
  run ${P}_nosched --preset $P --no-scheduler
  for L in 1 2 4; do run ${P}_lanes$L --preset $P --sched-lanes $L --sched-slots $((64 * L)); done
  run ${P}_rows24 --preset $P --sched-rows 24
  run ${P}_rows16 --preset $P --sched-rows 16
  run ${P}_stride4 --preset $P --sched-stride 4
  run ${P}_streams2 --preset $P --streams 2
  run ${P}_streams4 --preset $P --streams 4
  run ${P}_stagger --preset $P --no-stagger
done
