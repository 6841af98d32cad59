#!/usr/bin/env python3
"""Run one dense GEMM shape a few times (for rocprofv3 --pmc runs).  usage: gemm_one.py M N K [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
M, N, K = (int(x) for x in sys.argv[1:4])
it = int(sys.argv[4]) if len(sys.argv) > 4 else 5
a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
w = torch.randn(N, K, device=DEV, dtype=torch.bfloat16) * 0.02
out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
for _ in range(it):
    gemm("bf16", a, w, C_out=out)
torch.cuda.synchronize()
