#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
# bench.py sets this for --beams itself, but under rocprofv3 the profiler's preloaded library brings the runtime up before the program runs:
# the in-process setdefault comes too late, so export it here (the 4-calls-in-flight beam configuration needs its own hardware queues)
export GPU_MAX_HW_QUEUES=8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/beam_trace -- python3 $R/bench.py --beams 15 --streams 1 --steps 1 --warmup 1 --calls-per-step 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/beam_trace.log 2>&1
python3 - $R/gpurun_out/beam_trace <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:28]:
        print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}% calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:110]}')
PY
