#!/usr/bin/env python3
"""The decode-step GEMMs of the bench configuration (Llama-3.2-1B, 8 rows) through the weight-streaming kernels, a few launches each over
rotating weight copies (every byte read once, as in the decode loop).  Small on purpose: for `rocprofv3 --pmc` passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
td = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHARED = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # 1: the forms the decode scheduler's steps run (GemmArgs::shared_chip)
BIG = len(sys.argv) > 3 and sys.argv[3] == "8b"            # Llama-3.1-8B widths instead of Llama-3.2-1B
# name, N, K, K2, norm, swiglu, residual, f32 out
CASES = [("lora_dn", 128, 2048, 0, True, False, False, False), ("qkv+lora", 3072, 2048, 128, True, False, False, False),
         ("o_proj", 2048, 2048, 0, False, False, True, False), ("gate_up", 16384, 2048, 0, True, True, False, False),
         ("down", 2048, 8192, 0, False, False, True, False), ("lm_head", 128256, 2048, 0, True, False, False, True)]
if BIG:
    CASES = [("lora_dn", 512, 4096, 0, True, False, False, False), ("qkv+lora", 6144, 4096, 512, True, False, False, False),
             ("o_proj", 4096, 4096, 0, False, False, True, False), ("gate_up", 28672, 4096, 0, True, True, False, False),
             ("down", 4096, 14336, 0, False, False, True, False), ("lm_head", 128261, 4096, 0, True, False, False, True)]
for name, N, K, K2, norm, swi, res, f32 in CASES:
    nc = 4
    Ws = [torch.randn(N, K, device=DEV, dtype=td) * 0.02 for _ in range(nc)]
    W2 = torch.randn(N, K2, device=DEV, dtype=td) * 0.02 if K2 else None
    x = torch.randn(M, K, device=DEV, dtype=td)
    a2 = torch.randn(M, K2, device=DEV, dtype=td) if K2 else None
    gain = torch.ones(K, device=DEV)
    out = torch.empty(M, N // (2 if swi else 1), device=DEV, dtype=torch.float32 if f32 else td)
    r = torch.randn(M, N, device=DEV, dtype=td) if res else None
    for i in range(8):
        kw = {}
        if K2: kw.update(A2=a2, W2=W2)
        if norm: kw.update(norm_gain=gain, norm_eps=1e-5)
        if swi: kw.update(swiglu_pairs=1)
        gemm("bf16", x, Ws[i % nc], skinny=True, residual=r, C_out=out, out_f32=f32, shared_chip=SHARED, **kw)
torch.cuda.synchronize()
