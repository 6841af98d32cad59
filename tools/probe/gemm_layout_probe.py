#!/usr/bin/env python3
"""TIMING probe (results of the probe variants are garbage by construction): what would the persistent 256 x 256 GEMM's k loop gain if an
operand were stored blocked -- [row block of 8][k-tile][1 KiB], so that every wave's LDS-DMA instruction reads 1 KiB CONTIGUOUS instead of
8 rows x 128 B?  Needs the probe build (build_ab/csrc_probe -> build_ab/libomniavsr_probe.so: gemm4.hip takes the layout bits in the upper
half of the grid argument): OAVSR_LIB=build_ab/libomniavsr_probe.so python tools/probe/gemm_layout_probe.py
variants: 0 = shipped layout, 1 = A blocked, 2 = W blocked, 3 = both."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import gemm, DEV  # noqa: E402

SHAPES = {"qkv": (12000, 3072, 1024, 0, 0), "fc1": (12000, 4096, 1024, 1, 0), "gateup": (2504, 16384, 2048, 0, 1), "fc2": (12000, 1024, 4096, 0, 0),
          "sq4k": (4096, 4096, 4096, 0, 0), "sq8k": (8192, 8192, 8192, 0, 0)}


def main():
    rounds, iters = 5, 6
    for name, (M, N, K, act, sw) in SHAPES.items():
        per_set = (M * K + N * K + M * (N // 2 if sw else N)) * 2
        nset = max(2, int(300e6 // per_set) + 1)
        g = torch.Generator(device="cpu").manual_seed(1)
        sets = []
        for i in range(nset):
            A = (torch.rand(M, K, generator=g) * 2 - 1).to(DEV, torch.bfloat16)
            W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(DEV, torch.bfloat16)
            Cc = torch.empty(M, N // 2 if sw else N, device=DEV, dtype=torch.bfloat16)
            sets.append((A, W, Cc))
        bias = torch.randn(N, device=DEV) if not sw else None
        kw = dict(act=act)
        if sw:
            kw["swiglu_pairs"] = 1
        times = {v: [] for v in range(4)}
        for r in range(rounds):
            for v in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(2):
                    A, W, Cc = sets[i % nset]
                    gemm("bf16", A, W, bias=bias, kernel=3, grid=v << 16, C_out=Cc, **kw)
                e0.record()
                for i in range(iters):
                    A, W, Cc = sets[i % nset]
                    gemm("bf16", A, W, bias=bias, kernel=3, grid=v << 16, C_out=Cc, **kw)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3 / iters)
        fl = 2.0 * M * N * K
        print(f"{name}: M={M} N={N} K={K}")
        for v in range(4):
            t = sorted(times[v])
            print(f"  layout {v} ({['shipped', 'A blocked', 'W blocked', 'both blocked'][v]:>12}): median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f} us  {fl / t[len(t) // 2] / 1e6:7.1f} TFLOP/s")
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
