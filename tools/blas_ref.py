#!/usr/bin/env python3
"""What the vendor library (torch.matmul -> hipBLASLt) reaches on the bench's GEMM shapes: a yardstick for gemm2.hip, never used
by the product path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm
dev = "cuda"
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, K in [("whisper qkv", 12000, 3072, 1024), ("whisper fc1", 12000, 4096, 1024), ("whisper fc2", 12000, 1024, 4096), ("whisper o", 12000, 1024, 1024),
                      ("avh qkv", 3200, 3072, 1024), ("llm gu", 2504, 16384, 2048), ("llm down", 2504, 2048, 8192), ("square 4k", 4096, 4096, 4096), ("square 8k", 8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: torch.matmul(a, w.t(), out=out))
    ours = timeit(lambda: gemm("bf16", a, w, C_out=out))                  # this repo's dispatcher (gemm2.hip / gemm3.hip)
    try:
        pp = timeit(lambda: gemm("bf16", a, w, C_out=out, kernel=2))      # the 8-wave ping-pong kernel whatever the dispatcher's rule says
    except Exception:
        pp = float("nan")
    try:
        ps = timeit(lambda: gemm("bf16", a, w, C_out=out, kernel=3))      # the persistent ping-pong kernel (round 4; the dispatcher takes it when tiles > CUs)
    except Exception:
        ps = float("nan")
    fl = 2.0 * M * N * K / 1e6
    print(f"{name:12s} M={M:6d} N={N:6d} K={K:5d}: hipBLASLt {us:8.1f} us {fl / us:7.1f} TFLOP/s | this repo {ours:8.1f} us {fl / ours:7.1f} TFLOP/s | ping-pong {pp:8.1f} us {fl / pp:7.1f} TFLOP/s | persistent {ps:8.1f} us {fl / ps:7.1f} TFLOP/s")
