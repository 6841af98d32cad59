#!/usr/bin/env python3
"""TIMING probe (results of the probe variants are garbage by construction): where does the time between two k loops of the persistent
256 x 256 GEMM go?  Needs the probe build (build_ab/csrc_epi -> build_ab/libomniavsr_epi.so: gemm4.hip takes probe bits in the upper half of
the grid argument): OAVSR_LIB=build_ab/libomniavsr_epi.so python tools/probe/gemm_epilogue_probe.py
variants: 0 = shipped, 1 = everything but the global stores (plain wait counts), 2 = no epilogue at all (accumulators summed into a store
that never happens), 4 = shipped epilogue with the plain wait counts (the next tile waits for the stores)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import gemm, DEV  # noqa: E402

SHAPES = {"qkv": (12000, 3072, 1024, 0, 0, 0), "fc1": (12000, 4096, 1024, 1, 0, 0), "gateup": (2504, 16384, 2048, 0, 1, 0), "fc2+res": (12000, 1024, 4096, 0, 0, 1),
          "o+res": (12000, 1024, 1024, 0, 0, 1), "sq4k": (4096, 4096, 4096, 0, 0, 0), "sq8k": (8192, 8192, 8192, 0, 0, 0)}
VARIANTS = {0: "shipped", 1: "no global stores", 2: "no epilogue", 4: "stores waited for"}
if os.environ.get("PROBE_SET") == "2":       # build_ab/csrc_epi2: the register-only epilogue (weights as the A operand, permuted weight rows)
    VARIANTS = {0: "shipped", 1: "no global stores", 2: "unpermuted W rows", 3: "both", 4: "stores waited for"}


def main():
    rounds, iters = 5, 6
    for name, (M, N, K, act, sw, res) in SHAPES.items():
        per_set = (M * K + N * K + M * (N // 2 if sw else N) * (2 if res else 1)) * 2
        nset = max(2, int(300e6 // per_set) + 1)
        g = torch.Generator(device="cpu").manual_seed(1)
        sets = []
        for i in range(nset):
            A = (torch.rand(M, K, generator=g) * 2 - 1).to(DEV, torch.bfloat16)
            W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(DEV, torch.bfloat16)
            Cc = torch.empty(M, N // 2 if sw else N, device=DEV, dtype=torch.bfloat16)
            Rr = torch.randn(M, N, device=DEV).to(torch.bfloat16) if res else None
            sets.append((A, W, Cc, Rr))
        bias = torch.randn(N, device=DEV) if not sw else None
        kw = dict(act=act)
        if sw:
            kw["swiglu_pairs"] = 1
        times = {v: [] for v in VARIANTS}
        for r in range(rounds):
            for v in VARIANTS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(2):
                    A, W, Cc, Rr = sets[i % nset]
                    gemm("bf16", A, W, bias=bias, kernel=3, grid=v << 16, C_out=Cc, residual=Rr, **kw)
                e0.record()
                for i in range(iters):
                    A, W, Cc, Rr = sets[i % nset]
                    gemm("bf16", A, W, bias=bias, kernel=3, grid=v << 16, C_out=Cc, residual=Rr, **kw)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3 / iters)
        fl = 2.0 * M * N * K
        ntiles = ((M + 255) // 256) * ((N + 255) // 256)
        print(f"{name}: M={M} N={N} K={K}  ({ntiles} tiles)")
        for v in VARIANTS:
            t = sorted(times[v])
            print(f"  {v} ({VARIANTS[v]:>18}): median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f} us  {fl / t[len(t) // 2] / 1e6:7.1f} TFLOP/s")
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
