#!/usr/bin/env python3
"""Static split-K factors for the under-filled GEMMs of one bench step: time (GEMM + reduce pass) per factor, alone on the chip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
td = torch.bfloat16


def timeit(f, iters=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


SHAPES = [("llm down", 2504, 2048, 8192), ("llm o", 2504, 2048, 2048), ("llm qkv", 2504, 3072, 2048), ("whisper fc2", 12000, 1024, 4096), ("whisper o", 12000, 1024, 1024),
          ("avh fc2", 3200, 1024, 4096), ("avh o", 3200, 1024, 1024), ("avh qkv", 3200, 3072, 1024), ("avh fc1", 3200, 4096, 1024)]
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device=DEV, dtype=td); w = torch.randn(N, K, device=DEV, dtype=td) * 0.02
    out = torch.empty(M, N, device=DEV, dtype=td)
    res = torch.randn(M, N, device=DEV, dtype=td)
    bias = torch.randn(N, device=DEV)
    part = torch.empty(8 * M * N, device=DEV, dtype=torch.float32)
    row = []
    for ks in (1, 2, 3, 4, 6, 8):
        kw = dict(bias=bias, residual=res)
        if ks > 1:
            kw.update(ksplit=ks, partial=part)
        try:
            us = timeit(lambda: gemm("bf16", a, w, C_out=out, **kw))
            row.append(f"ks{ks} {us:6.1f}")
        except Exception as e:
            row.append(f"ks{ks} err")
    vend = timeit(lambda: torch.addmm(res, a, w.t(), out=out))
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(row) + f" | hipBLASLt {vend:6.1f} us", flush=True)
