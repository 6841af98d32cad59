#!/bin/bash
# Sequential-mode bench (one batch at a time) with the process restricted to N CUs (HSA_CU_MASK): how the MFMA side and the decode loop
# scale with the CU count -- the input for a static CU partition between the two kinds of work.
mkdir -p gpurun_out
for n in 256 224 192 128 96 64 48 32; do
  hi=$((n-1))
  echo "== CUs $n"
  HSA_CU_MASK="0:0-$hi" timeout 300 python bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
st=d['stages']
dec=sum(v['ms'] for k,v in st.items() if k.startswith('skinny') or k in ('decode_attention','rmsnorm','rope_kv'))
tot=sum(v['ms'] for v in st.values())
print('utt/s',d['value'],'ms/call',d['config']['ms_per_call'],'stage_sum',round(tot,2),'decode_kernels',round(dec,2),'mfma_side',round(tot-dec,2))
for k in ('skinny_mt1_nt4_norm_swiglu_bf16','skinny_mt1_nt1_bf16','skinny_mt1_nt4_norm_bf16','gemm_dma_256x256_bf16','gemm_8p_256x256_bf16','attention_bf16'):
    if k in st: print('   ',k,st[k]['ms'],st[k].get('gbs'),st[k].get('tflops'))
"
done
