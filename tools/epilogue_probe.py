#!/usr/bin/env python3
"""Cost of the fused epilogues on the bench's Whisper GEMM shapes: plain vs bias vs bias+GELU vs bias+residual."""
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
for M, N, K in ((12000, 4096, 1024), (12000, 3072, 1024), (12000, 1024, 4096), (12000, 1024, 1024)):
    a = torch.randn(M, K, device=DEV, dtype=td); w = torch.randn(N, K, device=DEV, dtype=td) * 0.02
    r = torch.randn(M, N, device=DEV, dtype=td); out = torch.empty(M, N, device=DEV, dtype=td); bias = torch.randn(N, device=DEV)
    line = f"M={M} N={N} K={K}:"
    cases = (("plain", {}), ("bias", dict(bias=bias)), ("bias+gelu", dict(bias=bias, act=1)), ("bias+res", dict(bias=bias, residual=r)))
    best = {n: 1e9 for n, _ in cases}
    for rep in range(4):                      # interleaved repeats: clocks drift by ~10 % over a run
        for name, kw in (cases if rep % 2 == 0 else cases[::-1]):
            best[name] = min(best[name], timeit(lambda: gemm("bf16", a, w, C_out=out, **kw)))
    print(line + "".join(f"  {n}: {v:6.1f} us" for n, v in best.items()))
