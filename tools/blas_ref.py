#!/usr/bin/env python3
"""What the vendor library (torch.matmul -> hipBLASLt) reaches on the bench's GEMM shapes: a yardstick for gemm2.hip, never used
by the product path."""
import torch
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
    print(f"{name:12s} M={M:6d} N={N:6d} K={K:5d}: {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s")
