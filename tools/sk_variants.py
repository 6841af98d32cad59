import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from gpu_util import gemm, DEV
def timeit(f, iters=60):
    for i in range(5): f(i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): f(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
td = torch.bfloat16
N, K, M = 16384, 2048, 8
nc = 8
Ws = [torch.randn(N, K, device=DEV, dtype=td) * 0.02 for _ in range(nc)]
x = torch.randn(M, K, device=DEV, dtype=td)
gain = torch.ones(K, device=DEV)
for name, kw, ncol in [("plain", {}, N), ("norm", dict(norm_gain=gain, norm_eps=1e-5), N), ("norm+swiglu", dict(norm_gain=gain, norm_eps=1e-5, swiglu_pairs=1), N // 2)]:
    out = torch.empty(M, ncol, device=DEV, dtype=td)
    us = timeit(lambda i: gemm("bf16", x, Ws[i % nc], skinny=True, C_out=out, **kw))
    print(f"gate_up {name:12s}: {us:6.1f} us  {N*K*2/us/1e3:6.0f} GB/s")
# empty-ish kernel launch floor: tiny skinny gemm
w0 = torch.randn(64, 64, device=DEV, dtype=td); x0 = torch.randn(8, 64, device=DEV, dtype=td); o0 = torch.empty(8, 64, device=DEV, dtype=td)
print("tiny launch floor: %.1f us" % timeit(lambda i: gemm("bf16", x0, w0, skinny=True, C_out=o0)))
