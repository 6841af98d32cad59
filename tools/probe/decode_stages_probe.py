#!/usr/bin/env python3
"""TIMING probe: how deep must the LDS pipeline of a decode-shaped GEMM be?  Needs the probe build (build_ab/csrc_st -> build_ab/libomniavsr_st.so:
kernel selectors 9 / 10 = the 128 x 64 tile with 4 / 6 LDS stages instead of 3, 11 = 128 x 128 with 4 stages):
OAVSR_LIB=build_ab/libomniavsr_st.so python tools/probe/decode_stages_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import gemm, DEV  # noqa: E402

SHAPES = {"1b gate_up": (16384, 2048, 1), "1b lm_head": (128261, 2048, 0), "8b gate_up": (28672, 4096, 1), "8b down (unsplit)": (4096, 14336, 0), "1b qkv (unsplit)": (3072, 2048, 0)}
SEL = {0: "shipped", 9: "128x64, 4 stages", 10: "128x64, 6 stages", 11: "128x128, 4 stages"}


def main():
    for name, (N, K, sw) in SHAPES.items():
        for R in (1, 8, 32, 97):
            ncopy = max(2, int(400e6 // (N * K * 2)) + 1)
            Ws = [torch.randn(N, K, device=DEV, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)]
            A = torch.randn(R, K, device=DEV, dtype=torch.bfloat16)
            out = torch.empty(R, N // 2 if sw else N, device=DEV, dtype=torch.bfloat16 if sw else torch.float32)
            kw = dict(swiglu_pairs=1) if sw else dict(out_f32=True)
            line = f"{name:18s} rows={R:3d}:"
            for sel, what in SEL.items():
                ts = []
                for rep in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    gemm("bf16", A, Ws[0], kernel=sel, C_out=out, decode_tiled=1, **kw)
                    e0.record()
                    for i in range(6):
                        gemm("bf16", A, Ws[i % ncopy], kernel=sel, C_out=out, decode_tiled=1, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 6)
                ts.sort()
                line += f"  {what}: {ts[2]:6.1f} us"
            print(line, flush=True)
            del Ws


if __name__ == "__main__":
    main()
