#!/bin/bash
# Round profile on the GPU box: rocprofv3 --kernel-trace --stats of the default bench (decode scheduler on) and of the per-call-decode
# mode; per-kernel summaries + a concurrency summary of the trace (tools/trace_phases.py) are left under gpurun_out/ (copy into profiles/).
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for tag in main nosched; do
  mode=""; [ "$tag" = nosched ] && mode="--no-scheduler"
  rm -rf "$R/gpurun_out/prof_$tag"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$tag" -- python3 "$R/bench.py" --no-cpu-baseline --no-grouped --steps 8 --warmup 2 $mode > "$R/gpurun_out/prof_$tag.json" 2> "$R/gpurun_out/prof_$tag.err" || echo "rocprofv3 ($tag) failed"
  tr=$(find "$R/gpurun_out/prof_$tag" -name "*kernel_trace.csv" | head -1)
  if [ -n "$tr" ]; then
    python3 "$R/tools/trace_phases.py" "$tr" 0.4 > "$R/gpurun_out/prof_${tag}_overlap.txt" 2>&1 || true
    rm -f "$tr"                                   # tens of MB; the summaries are what is kept
  fi
  st=$(find "$R/gpurun_out/prof_$tag" -name "*kernel_stats.csv" | head -1)
  [ -n "$st" ] && cp "$st" "$R/gpurun_out/prof_${tag}_kernel_stats.csv"
  tail -c 700 "$R/gpurun_out/prof_$tag.json" | head -c 300; echo
  cat "$R/gpurun_out/prof_${tag}_overlap.txt" 2>/dev/null | head -20
done
