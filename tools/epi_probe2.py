#!/usr/bin/env python3
"""How much of a K = 1024 GEMM is its epilogue: the same launch with and without bias / GELU / residual, alone on the chip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV


def timeit(f, iters=40):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, M, N, K in [("whisper fc1", 12000, 4096, 1024), ("whisper qkv", 12000, 3072, 1024), ("whisper o", 12000, 1024, 1024), ("K=512 probe", 12000, 4096, 512), ("K=2048 probe", 12000, 4096, 2048)]:
    a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16); w = torch.randn(N, K, device=DEV, dtype=torch.bfloat16) * 0.02
    c = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    bias = torch.randn(N, device=DEV); res = torch.randn(M, N, device=DEV, dtype=torch.bfloat16)
    t0 = timeit(lambda: gemm("bf16", a, w, C_out=c))
    t1 = timeit(lambda: gemm("bf16", a, w, C_out=c, bias=bias))
    t2 = timeit(lambda: gemm("bf16", a, w, C_out=c, bias=bias, act=1))
    t3 = timeit(lambda: gemm("bf16", a, w, C_out=c, bias=bias, residual=res))
    print(f"{name:14s} plain {t0:6.1f} | +bias {t1:6.1f} | +bias+gelu {t2:6.1f} | +bias+residual {t3:6.1f} us   ({2.0*M*N*K/t0/1e6:6.1f} TFLOP/s plain)")
