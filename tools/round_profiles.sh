#!/bin/bash
# Everything profiles/rNN_* is made of, in one GPU call: the driver-format bench line, the presets, beam search, the caller-side loop with host
# inputs, rocprofv3 kernel stats (scheduler on / off) + the concurrency summary of the trace, the beam-search kernel stats.
#   usage: round_profiles.sh <tag>      (e.g. r03) -> gpurun_out/profiles_<tag>/  (copy into profiles/)
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O="$R/gpurun_out/profiles_$TAG"
mkdir -p "$O"
cd "$R"
python bench.py --steps 20 --warmup 2 > "$O/${TAG}_bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
python bench.py --preset c2 --steps 10 --warmup 2 --no-cpu-baseline > "$O/${TAG}_bench_configs2_vsr_b16.json" 2>> "$O/bench.err"; echo "c2 rc=$?"
python bench.py --preset c3 --steps 10 --warmup 2 --no-cpu-baseline > "$O/${TAG}_bench_configs3_qwen25_7b.json" 2>> "$O/bench.err"; echo "c3 rc=$?"
python bench.py --preset c4 --steps 10 --warmup 2 --no-cpu-baseline > "$O/${TAG}_bench_configs4_llama31_8b.json" 2>> "$O/bench.err"; echo "c4 rc=$?"
python bench.py --beams 15 --steps 8 --warmup 1 --no-cpu-baseline > "$O/${TAG}_bench_beam15.json" 2>> "$O/bench.err"; echo "beam rc=$?"
python bench.py --testset 1321 > "$O/${TAG}_bench_testset1321.json" 2>> "$O/bench.err"; echo "testset rc=$?"
timeout 1500 python tools/harness_throughput.py 960 > "$O/${TAG}_harness_throughput.txt" 2>&1; echo "harness rc=$?"
bash tools/prof_round.sh > "$O/prof_round.log" 2>&1
cp gpurun_out/prof_main_kernel_stats.csv "$O/${TAG}_kernel_stats.csv" 2>/dev/null
cp gpurun_out/prof_nosched_kernel_stats.csv "$O/${TAG}_kernel_stats_nosched.csv" 2>/dev/null
{ echo "# rocprofv3 --kernel-trace of the headline run (decode scheduler on), tools/trace_phases.py over the last 60 % of the trace";
  echo "# NOTE: under the profiler dispatches of different queues do not overlap (the profiled run takes ~1.6 x the un-profiled time per call,";
  echo "# see ${TAG}_kernel_stats_bench.json vs ${TAG}_bench.json): the enc / dec split below is a sum of kernel durations, not the un-profiled concurrency";
  echo "# (that is bench.py's schedule record: both_active_frac).";
  cat gpurun_out/prof_main_overlap.txt; echo; echo "# the same with --no-scheduler"; cat gpurun_out/prof_nosched_overlap.txt; } > "$O/${TAG}_trace_overlap.txt" 2>/dev/null
cp gpurun_out/prof_main.json "$O/${TAG}_kernel_stats_bench.json" 2>/dev/null
bash tools/beam_trace.sh > "$O/${TAG}_beam15_kernel_stats.txt" 2>&1
for f in "$O"/*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split("/")[-1], d["value"], d["config"]["ms_per_call"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
tail -8 "$O/${TAG}_harness_throughput.txt"
