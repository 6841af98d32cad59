#!/bin/bash
# Round-4 counter passes: FETCH_SIZE | WRITE_SIZE | SQ_* (separate rocprofv3 runs, each with --kernel-trace only) over the GEMM mix of one
# bench step (tools/gemm_mix.py: the persistent ping-pong kernel now takes its multi-round shapes) and over the decode GEMVs in the forms the
# decode scheduler's steps run (GemmArgs::shared_chip): 32 rows at the Llama-3.2-1B widths, 64 rows at the Llama-3.1-8B widths;
# per-family JSON files land in gpurun_out/pmc_out (copy into profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out/pmc_out"
cd /tmp && export TMPDIR=/tmp
for what in gemm skinny32s skinny64s8b; do
  prog="$R/tools/gemm_mix.py"; args="1"
  [ "$what" = skinny32s ] && { prog="$R/tools/skinny_mix.py"; args="32 1"; }
  [ "$what" = skinny64s8b ] && { prog="$R/tools/skinny_mix.py"; args="64 1 8b"; }
  i=0
  for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf "$R/gpurun_out/pmc_${what}_$i"
    timeout 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_${what}_$i" -- python3 "$prog" $args > "$R/gpurun_out/pmc_${what}_$i.log" 2>&1 || echo "pass $what $i failed"
  done
  python3 "$R/tools/pmc_round.py" "r04_$what" "$R/gpurun_out/pmc_${what}_1" "$R/gpurun_out/pmc_${what}_2" "$R/gpurun_out/pmc_${what}_3" "$R/gpurun_out/pmc_out"
done
find "$R/gpurun_out" -name "*kernel_trace.csv" -size +8M -delete
find "$R/gpurun_out" -name "*counter_collection.csv" -size +8M -delete
timeout 600 python3 "$R/tools/blas_ref.py" > "$R/gpurun_out/pmc_out/r04_vendor_yardstick.txt" 2>&1 || true
tail -30 "$R/gpurun_out/pmc_out/r04_vendor_yardstick.txt"
