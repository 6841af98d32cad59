#!/usr/bin/env python3
"""One GEMM shape list under the tile forced by OAVSR_GEMM_TILE (set by the caller): time per launch, alone on the chip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV


def timeit(f, iters=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


out = []
for name, M, N, K in [("avh qkv", 3200, 3072, 1024), ("avh fc1", 3200, 4096, 1024), ("llm o", 2504, 2048, 2048), ("llm qkv", 2504, 3072, 2048), ("whisper o", 12000, 1024, 1024)]:
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16); w = torch.randn(N, K, device=DEV, dtype=torch.bfloat16) * 0.02
    c = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    us = timeit(lambda: gemm("bf16", a, w, C_out=c, bias=bias))
    out.append(f"{name} {us:6.1f}")
print(f"TILE={os.environ.get('OAVSR_GEMM_TILE', 'auto'):>4s} SMALL={os.environ.get('OAVSR_GEMM_SMALL', '1')}: " + " | ".join(out))
