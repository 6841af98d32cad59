#!/usr/bin/env python3
"""Time one (M, N) at several K: the slope is the per-k-tile time of the main loop, the intercept the per-round fixed cost
(launch, prologue latency, epilogue) of the GEMM kernel.  usage: gemm_kscan.py M N [epilogue: 0 none, 1 bias+gelu, 2 bias+residual]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
M, N = int(sys.argv[1]), int(sys.argv[2])
epi = int(sys.argv[3]) if len(sys.argv) > 3 else 0
td = torch.bfloat16
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = torch.empty(M, N, device=DEV, dtype=td)
kw = {}
if epi: kw["bias"] = torch.randn(N, device=DEV)
if epi == 1: kw["act"] = 1
if epi == 2: kw["residual"] = torch.randn(M, N, device=DEV, dtype=td)
for K in (256, 512, 1024, 2048, 4096):
    a = torch.randn(M, K, device=DEV, dtype=td); w = torch.randn(N, K, device=DEV, dtype=td) * 0.02
    us = timeit(lambda: gemm("bf16", a, w, C_out=out, **kw))
    print(f"M={M} N={N} K={K:5d} epi={epi}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")
