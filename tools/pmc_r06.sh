#!/bin/bash
# Round-6 counter passes (MI355X_MICROARCH.md: FETCH_SIZE, WRITE_SIZE and the SQ counters in SEPARATE rocprofv3 runs, each with --kernel-trace
# only, the program itself behind `--`): (a) the JOB -- bench.py's own timed loop with one GPU (no child process, no launcher; 2 + 2 + 2 calls:
# under --pmc every dispatch costs ~3 ms, and a call is ~5 000 dispatches), so that
# roofline.traffic refers to the launches that were timed; (b) tools/decode_mix.py: the decode-shaped launches of a step as they ship, at 1 and 8 rows
# (the strips of decode_strip.hip) and at 97 rows (Llama-3.2-1B and Llama-3.1-8B widths).
# Aggregated per kernel family by tools/pmc_round.py into gpurun_out/pmc_r06/*.json (copy into profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O="$R/gpurun_out/pmc_r06"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run() {   # tag, script args...
  local tag=$1; shift
  for pass in fetch write sq; do
    case $pass in fetch) P="FETCH_SIZE";; write) P="WRITE_SIZE";; sq) P="$SQ";; esac
    rm -rf "$O/${tag}_$pass"
    timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$O/${tag}_$pass" -- python3 "$@" > "$O/${tag}_$pass.log" 2>&1 || echo "pass $tag/$pass failed"
  done
  python3 "$R/tools/pmc_round.py" "r06_$tag" "$O/${tag}_fetch" "$O/${tag}_write" "$O/${tag}_sq" "$O"
  rm -rf "$O/${tag}_fetch" "$O/${tag}_write" "$O/${tag}_sq"
}
run job "$R/bench.py" --gpus 1 --steps 1 --warmup 1 --calls-per-step 2 --no-cpu-baseline --no-roofline --no-grouped
run decode_r1 "$R/tools/decode_mix.py" 1 1b 3
run decode_r8 "$R/tools/decode_mix.py" 8 1b 3
run decode "$R/tools/decode_mix.py" 97 1b 2
run decode8b "$R/tools/decode_mix.py" 97 8b 1
ls "$O"
