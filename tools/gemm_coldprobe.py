#!/usr/bin/env python3
"""Back-to-back launches of one GEMM shape with operands rotated over more copies than the 256 MiB Infinity Cache holds:
which operand being "cold" (A, W, or the C destination) costs how much?   usage: gemm_coldprobe.py M N K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
M, N, K = (int(x) for x in sys.argv[1:4])
td = torch.bfloat16
nA = max(1, int(400e6 // (M * K * 2)) + 1); nW = max(1, int(400e6 // (N * K * 2)) + 1); nC = max(1, int(400e6 // (M * N * 2)) + 1)
As = [torch.randn(M, K, device=DEV, dtype=td) for _ in range(nA)]
Ws = [torch.randn(N, K, device=DEV, dtype=td) * 0.02 for _ in range(nW)]
Cs = [torch.empty(M, N, device=DEV, dtype=td) for _ in range(nC)]
bias = torch.randn(N, device=DEV)


def run(tag, ra, rw, rc, iters=60, act=0):
    def go(i):
        gemm("bf16", As[i % nA if ra else 0], Ws[i % nW if rw else 0], C_out=Cs[i % nC if rc else 0], bias=bias, act=act)
    for i in range(10):
        go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        go(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{tag:34s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s")


run("warm-up", 0, 0, 0)
run("all resident (same buffers)", 0, 0, 0)
run("A rotated", 1, 0, 0)
run("W rotated", 0, 1, 0)
run("C rotated", 0, 0, 1)
run("A, W, C rotated", 1, 1, 1)
run("all resident (again)", 0, 0, 0)
run("all resident, GELU epilogue", 0, 0, 0, act=1)
run("all resident, no act (again)", 0, 0, 0)
