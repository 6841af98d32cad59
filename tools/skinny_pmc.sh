#!/bin/bash
set -u
# three counter passes (FETCH_SIZE | WRITE_SIZE | SQ_*) over the decode GEMVs (tools/skinny_mix.py), aggregated per kernel family into
# profiles/<tag>_pmc_traffic_<family>.json by tools/pmc_round.py.   usage: skinny_pmc.sh [tag]   (default r02_skinny)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r02_skinny}
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/skpmc_$i -- python3 $R/tools/skinny_mix.py 8 > $R/gpurun_out/skpmc_$i.log 2>&1
done
mkdir -p $R/gpurun_out/pmc_out
python3 $R/tools/pmc_round.py $TAG $R/gpurun_out/skpmc_1 $R/gpurun_out/skpmc_2 $R/gpurun_out/skpmc_3 $R/gpurun_out/pmc_out
find $R/gpurun_out -name "*kernel_trace.csv" -size +8M -delete
ls $R/gpurun_out/pmc_out
