#!/usr/bin/env python3
"""Which split-K factor for the decode-shaped GEMMs?  Per call site (Llama-3.2-1B / 3.1-8B widths) and row count: the launch time with
ks = 1, 2, 4, 8, 16 k-slices (reduce inside the launch), weight copies rotating beyond the Infinity Cache.  The engine's rule is
model.hip: decode_split (a function of N and K only -- part of the decode regime's summation order)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from omni_avsr_amd import lib as L  # noqa: E402
from gpu_util import gemm, DEV  # noqa: E402

td = torch.bfloat16
SITES = [("1b lora_dn", 256, 2048), ("1b qkv", 3072, 2304), ("1b o_proj", 2048, 2048), ("1b down", 2048, 8192),
         ("8b lora_dn", 512, 4096), ("8b qkv", 6144, 4608), ("8b o_proj", 4096, 4096), ("8b down", 4096, 14336)]
sync = torch.zeros(L.GEMM_SYNC_INTS, dtype=torch.int32, device=DEV)
part = torch.empty(16 * 128 * 6144, device=DEV, dtype=torch.float32)
for name, N, K in SITES:
    ncopy = max(2, int(300e6 // (N * K * 2)) + 1)
    Ws = [torch.randn(N, K, device=DEV, dtype=td) * 0.02 for _ in range(min(ncopy, 24))]
    for R in (8, 97):
        x = torch.randn(R, K, device=DEV, dtype=td)
        out = torch.empty(R, N, device=DEV, dtype=td)
        line = f"{name:10s} N={N:5d} K={K:5d} rows={R:3d}:"
        for ks in (1, 2, 4, 8, 16):
            if K // 64 < ks:
                continue
            kw = dict(ksplit=ks, partial=part, sync=sync) if ks > 1 else dict(decode_tiled=1)
            for i in range(3):
                gemm("bf16", x, Ws[i % len(Ws)], C_out=out, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 30
            e0.record()
            for i in range(n):
                gemm("bf16", x, Ws[i % len(Ws)], C_out=out, **kw)
            e1.record()
            torch.cuda.synchronize()
            line += f"  ks={ks:2d} {e0.elapsed_time(e1) * 1e3 / n:6.1f} us"
        print(line, flush=True)
    del Ws
    torch.cuda.empty_cache()
