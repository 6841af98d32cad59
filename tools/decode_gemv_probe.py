#!/usr/bin/env python3
"""The decode GEMVs (8 rows, Llama-3.2-1B shapes) replayed from a HIP graph, 48 launches of each over rotating weight copies: time per
launch as in the decode loop, next to what a pure contiguous stream of the same bytes takes (tools/probe/cu_ingest)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
td = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cases = [  # name, N, K, K2, norm, swiglu, residual, out_f32
    ("lora_dn", 256, 2048, 0, True, False, False, False),
    ("qkv+lora", 3072, 2048, 256, True, False, False, False),
    ("qkv plain", 3072, 2048, 0, False, False, False, False),
    ("o_proj", 2048, 2048, 0, False, False, True, False),
    ("gate_up+norm+swiglu", 16384, 2048, 0, True, True, False, False),
    ("gate_up plain", 16384, 2048, 0, False, False, False, False),
    ("down", 2048, 8192, 0, False, False, True, False),
    ("lm_head+norm", 128256, 2048, 0, True, False, False, True),
    ("lm_head plain", 128256, 2048, 0, False, False, False, True),
    ("lm_head plain bf16 out", 128256, 2048, 0, False, False, False, False),
]
side = torch.cuda.Stream()
for name, N, K, K2, norm, swi, res, of32 in cases:
    nbytes = N * (K + K2) * 2
    ncopy = max(3, min(48, int(600e6 // nbytes) + 1))
    Ws = [torch.randn(N, K, device=DEV, dtype=td) * 0.02 for _ in range(ncopy)]
    W2s = [torch.randn(N, K2, device=DEV, dtype=td) * 0.02 for _ in range(ncopy)] if K2 else None
    x = torch.randn(M, K, device=DEV, dtype=td); a2 = torch.randn(M, K2, device=DEV, dtype=td) if K2 else None
    gain = torch.ones(K, device=DEV)
    out = torch.empty(M, N // (2 if swi else 1), device=DEV, dtype=torch.float32 if of32 else td)
    r = torch.randn(M, N, device=DEV, dtype=td) if res else None

    def run(i):
        kw = {}
        if K2: kw.update(A2=a2, W2=W2s[i % ncopy])
        if norm: kw.update(norm_gain=gain, norm_eps=1e-5)
        if swi: kw.update(swiglu_pairs=1)
        gemm("bf16", x, Ws[i % ncopy], skinny=True, residual=r, C_out=out, out_f32=of32, **kw)
    nl = 48
    with torch.cuda.stream(side):
        for i in range(3): run(i)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(nl): run(i)
        for _ in range(2): g.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(5): g.replay()
        e1.record(side)
        side.synchronize()
    us = e0.elapsed_time(e1) / (5 * nl) * 1e3
    print(f"M={M} {name:24s} N={N:6d} K={K:5d}+{K2:3d}: {us:7.2f} us per launch  {nbytes / us / 1e3:7.0f} GB/s", flush=True)
    del Ws, W2s
