#!/usr/bin/env python3
"""How much of the 256x256 LDS-DMA GEMM's k-tile time is memory latency?  Times M = N = K = 8192 (bf16) with real operands and
with every row of A and W aliased to one row (stride 0: all DMA requests hit the same few L1/L2 lines)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
if len(sys.argv) > 3:
    M, N, K = (int(x) for x in sys.argv[1:4])
else:
    M = N = K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
w = torch.randn(N, K, device=DEV, dtype=torch.bfloat16) * 0.02
out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)


def run(tag, A, W):
    for _ in range(3):
        gemm("bf16", A, W, C_out=out, M=M, N=N, K=K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        gemm("bf16", A, W, C_out=out, M=M, N=N, K=K)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    rounds = -(-(-(-M // 256) * -(-N // 256)) // 256)
    print(f"{tag:28s} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s   {us / rounds / (K // 64) * 1e3:7.1f} ns per k-tile and round")


run("real operands", a, w)
for pad in (64, 192, 320):   # row stride = K + pad elements: do power-of-two strides pile the tile's lines onto few L2 channels?
    ap = torch.randn(M, K + pad, device=DEV, dtype=torch.bfloat16)[:, :K]
    wp = (torch.randn(N, K + pad, device=DEV, dtype=torch.bfloat16) * 0.02)[:, :K]
    run(f"row stride K + {pad}", ap, wp)
    run(f"  only A padded (+{pad})", ap, w)
    run(f"  only W padded (+{pad})", a, wp)
run("real operands (again)", a, w)
run("A rows aliased (stride 0)", a[:1].expand(M, K), w)
run("W rows aliased", a, w[:1].expand(N, K))
run("A and W aliased", a[:1].expand(M, K), w[:1].expand(N, K))
