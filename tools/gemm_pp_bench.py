#!/usr/bin/env python3
"""A/B timing of the 256 x 256 bf16 GEMM kernels on the benchmark's multi-round shapes, all variants interleaved in ONE process
(guide rule 24): kernel selector 5 = 16-wave LDS-DMA, 2 = ping-pong, 3 = persistent ping-pong,
7 = 256 x 128, 8 = persistent ping-pong with 192-row tiles, 0 = the dispatcher's choice.  Operand sets rotate so that no set survives in the 256 MiB Infinity Cache between its uses.

usage: gemm_pp_bench.py [--rounds R] [--shapes qkv,fc1,gateup,fc2,sq4k,sq8k] [--grids 0,256]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import gemm, DEV  # noqa: E402

SHAPES = {
    # name: (M, N, K, act, swiglu, bias)
    "qkv": (12000, 3072, 1024, 0, 0, True),
    "fc1": (12000, 4096, 1024, 1, 0, True),
    "gateup": (2504, 16384, 2048, 0, 1, False),
    "avfc1": (3200, 4096, 1024, 1, 0, True),
    "fc2": (12000, 1024, 4096, 0, 0, True),
    "oproj": (12000, 1024, 1024, 0, 0, True),
    "sq4k": (4096, 4096, 4096, 0, 0, False),
    "sq8k": (8192, 8192, 8192, 0, 0, False),
    "qkv16": (24000, 3072, 1024, 0, 0, True),
    # the LLM prefill's and AV-HuBERT's sub-round shapes (8 x 16 s clips: 2504 prompt rows, 3200 video frames)
    "llm_qkv": (2504, 3072, 2048, 0, 0, False),
    "llm_o": (2504, 2048, 2048, 0, 0, False),
    "llm_down": (2504, 2048, 8192, 0, 0, False),
    "avh_o": (3200, 1024, 1024, 0, 0, True),
    "avh_fc2": (3200, 1024, 4096, 0, 0, True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--shapes", default="qkv,fc1,gateup,sq4k")
    ap.add_argument("--grids", default="0")
    ap.add_argument("--sels", default="5,2,3")
    a = ap.parse_args()
    sels = [int(x) for x in a.sels.split(",")]
    grids = [int(x) for x in a.grids.split(",")]
    for name in a.shapes.split(","):
        M, N, K, act, sw, has_bias = SHAPES[name]
        per_set = (M * K + N * K + M * (N // 2 if sw else N)) * 2
        nset = max(2, int(300e6 // per_set) + 1)
        g = torch.Generator(device="cpu").manual_seed(1)
        sets = []
        for i in range(nset):
            A = (torch.rand(M, K, generator=g) * 2 - 1).to(DEV, torch.bfloat16)
            W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(DEV, torch.bfloat16)
            Cc = torch.empty(M, N // 2 if sw else N, device=DEV, dtype=torch.bfloat16)
            sets.append((A, W, Cc))
        bias = torch.randn(N, device=DEV) if has_bias else None
        variants = []
        for s in sels:
            if s == 3:
                for gr in grids:
                    variants.append((s, gr))
            else:
                variants.append((s, 0))
        kw = dict(act=act)
        if sw:
            kw["swiglu_pairs"] = 1
        # reference bits from the 16-wave kernel; every variant must reproduce them
        ref = gemm("bf16", sets[0][0], sets[0][1], bias=bias, kernel=5, **kw).clone()
        times = {v: [] for v in variants}
        for v in variants:
            out = gemm("bf16", sets[0][0], sets[0][1], bias=bias, kernel=v[0], grid=v[1], **kw)
            assert torch.equal(out, ref), f"{name}: variant {v} differs from the 16-wave kernel"
        torch.cuda.synchronize()
        for r in range(a.rounds):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(2):
                    A, W, Cc = sets[i % nset]
                    gemm("bf16", A, W, bias=bias, kernel=v[0], grid=v[1], C_out=Cc, **kw)
                e0.record()
                for i in range(a.iters):
                    A, W, Cc = sets[i % nset]
                    gemm("bf16", A, W, bias=bias, kernel=v[0], grid=v[1], C_out=Cc, **kw)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3 / a.iters)
        fl = 2.0 * M * N * K
        print(f"{name}: M={M} N={N} K={K} act={act} swiglu={sw} sets={nset}")
        for v in variants:
            t = sorted(times[v])
            med, mn = t[len(t) // 2], t[0]
            print(f"  sel {v[0]} grid {v[1]:3d}: median {med:8.1f} us  min {mn:8.1f} us   {fl / med / 1e6:7.1f} TFLOP/s (median)  frac {fl / med / 1e6 / 2500:.3f}")
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
