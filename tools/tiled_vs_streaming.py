#!/usr/bin/env python3
"""Decode GEMMs at the scheduler's row counts, 1B and 7B / 8B widths: the weight-streaming kernels (shared-chip forms) against the tiled
LDS-DMA GEMM (un-split and split 2 / 4), weight copies rotating beyond the Infinity Cache.  What made greedy decode move to the tiled regime
(DESIGN.md section 5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
td = torch.bfloat16
for M in (57, 64, 32):
    for name, N, K in [("gate_up", 28672, 4096), ("down", 4096, 14336), ("o_proj", 4096, 4096), ("qkv", 6144, 4096), ("lm_head", 128256, 4096), ("gate_up 1B", 16384, 2048), ("down 1B", 2048, 8192)]:
        nb = N * K * 2
        nc = max(2, int(700e6 // nb) + 1)
        Ws = [(torch.randn(N, K) * 0.02).to(DEV, td) for _ in range(nc)]
        x = torch.randn(M, K).to(DEV, td)
        out = torch.empty(M, N, device=DEV, dtype=td)
        line = f"M={M} {name:10s} ({nb/1e6:.0f} MB)"
        for label, kw in (("skinny", dict(skinny=True, shared_chip=1)), ("tiled", dict()), ("tiled ks2", dict(ksplit=2)), ("tiled ks4", dict(ksplit=4))):
            if "ksplit" in kw:
                kw = dict(kw, partial=torch.empty(kw["ksplit"] * M * N, device=DEV, dtype=torch.float32))
            try:
                for i in range(nc): gemm("bf16", x, Ws[i % nc], C_out=out, **kw)
                best = 1e9
                for rep in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(2 * nc): gemm("bf16", x, Ws[i % nc], C_out=out, **kw)
                    e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / (2 * nc))
                line += f" | {label}: {best:6.1f} us {nb/best/1e6:5.2f} TB/s"
            except Exception as e:
                line += f" | {label}: ERR"
        print(line, flush=True)
        del Ws
        torch.cuda.empty_cache()
