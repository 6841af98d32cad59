import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
td = torch.bfloat16
def timeit(f, iters=30):
    for i in range(3): f(i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): f(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 120
for name, N, K in [("gate_up", 16384, 2048), ("lm_head", 128256, 2048)]:
    nc = max(2, int(600e6 // (N * K * 2)) + 1)
    Ws = [torch.randn(N, K, device=DEV, dtype=td) * 0.02 for _ in range(nc)]
    x = torch.randn(M, K, device=DEV, dtype=td)
    out = torch.empty(M, N, device=DEV, dtype=td)
    us = timeit(lambda i: gemm("bf16", x, Ws[i % nc], C_out=out))
    print(f"M=120 {name}: {us:7.1f} us  {N*K*2/us/1e3:6.0f} GB/s")
