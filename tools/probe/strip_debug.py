#!/usr/bin/env python3
"""debug: where do the strip kernels differ from torch (by ksplit / rows)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
torch.manual_seed(0)
for (M, N, K, S) in [(1, 64, 2048, 4), (1, 64, 2048, 8), (8, 64, 2048, 8), (1, 64, 4096, 8), (1, 64, 1024, 8), (1, 128, 2048, 16), (1, 64, 512, 8), (16, 2048, 8192, 8), (20, 2048, 8192, 8)]:
    a = torch.randn(M, K, device=DEV).bfloat16(); w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).bfloat16()
    out = gemm("bf16", a, w, ksplit=S).float()
    ref = a.float() @ w.float().T
    err = (out - ref).abs()
    bad = (err > 2e-2).nonzero()
    print(f"M={M} N={N} K={K} S={S}: max err {err.max().item():.3e}, bad {bad.shape[0]} / {err.numel()}", bad[:6].tolist())
    if bad.shape[0]:
        # which slices are missing?  compare against partial sums
        per = -(-(K // 64) // S)
        parts = [a.float()[:, s * per * 64:(s + 1) * per * 64] @ w.float()[:, s * per * 64:(s + 1) * per * 64].T for s in range(S)]
        m, n = bad[0].tolist()
        print("   out", out[m, n].item(), "ref", ref[m, n].item(), "slice partials", [round(p[m, n].item(), 4) for p in parts])
