#!/usr/bin/env python3
"""LLM prefill projections with few output tiles (N = 2048): one pass vs split-K 2 / 4 (LDS-DMA 256x128 tiles + reduce pass)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
td = torch.bfloat16
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (2504, 1850, 600):
    for N, K in ((2048, 8192), (2048, 2048), (3072, 2048)):
        a = torch.randn(M, K, device=DEV, dtype=td); w = torch.randn(N, K, device=DEV, dtype=td) * 0.02
        r = torch.randn(M, N, device=DEV, dtype=td); out = torch.empty(M, N, device=DEV, dtype=td)
        part = torch.empty(8 * M * N, device=DEV, dtype=torch.float32)
        line = f"M={M:5d} N={N:5d} K={K:5d}:"
        for ks in (1, 2, 4, 8):
            kw = dict(ksplit=ks, partial=part) if ks > 1 else {}
            us = timeit(lambda: gemm("bf16", a, w, residual=r, C_out=out, **kw))
            line += f"  ks={ks}: {us:7.1f} us ({2.0*M*N*K/us/1e6:6.0f} TF)"
        print(line)
