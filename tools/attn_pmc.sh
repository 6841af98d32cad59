#!/bin/bash
set -u
# two SQ counter passes over the Whisper-shaped attention (tools/attn_mix.py); summaries land in gpurun_out/attn_pmc_*.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc_$i -- python3 $R/tools/attn_mix.py 5 > $R/gpurun_out/attn_pmc_$i.log 2>&1
  python3 - $R/gpurun_out/attn_pmc_$i <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "attn_dma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print(k, acc[k] / n[k], "per launch over", n[k])
PY
done
