#!/usr/bin/env python3
"""The seven GEMM shapes that run on the persistent 256 x 256 kernel (gemm_8pp) in one bench step (bench config: AV-HuBERT-L + Whisper-medium +
Llama-3.2-1B, B = 8 x 16 s), each launched once per layer it occurs in, with bias / GELU / residual epilogues as in the model.
For `rocprofv3 --pmc` passes: the per-launch average over this mix is the per-launch HBM traffic of the bench's dominant kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
td = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
# (M, N, K, K2, launches per step, gelu, residual).  Round 4: the first three (more tiles than CUs) run on the persistent ping-pong kernel.
MIX = [(12000, 4096, 1024, 0, 24, 1, 0), (2504, 16384, 2048, 0, 16, 0, 0), (12000, 3072, 1024, 0, 24, 0, 0), (12000, 1024, 4096, 0, 24, 0, 1),
       (3200, 4096, 1024, 0, 24, 1, 0), (12000, 1024, 1024, 0, 24, 0, 1), (3200, 3072, 1088, 0, 24, 0, 0)]      # (AV-HuBERT QKV: the LoRA term rides in ONE K range of 1024 + 64, model.hip: enc_cat)
for M, N, K, K2, n, gelu, res in MIX:
    a = torch.randn(M, K, device=DEV, dtype=td); w = torch.randn(N, K, device=DEV, dtype=td) * 0.02
    bias = torch.randn(N, device=DEV)
    out = torch.empty(M, N, device=DEV, dtype=td)
    kw = dict(bias=bias)
    if gelu: kw["act"] = 1
    if res: kw["residual"] = torch.randn(M, N, device=DEV, dtype=td)
    if K2: kw.update(A2=torch.randn(M, K2, device=DEV, dtype=td), W2=torch.randn(N, K2, device=DEV, dtype=td) * 0.02)
    for _ in range(n * reps):
        gemm("bf16", a, w, C_out=out, **kw)
torch.cuda.synchronize()
