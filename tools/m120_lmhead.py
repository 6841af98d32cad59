import os, sys, math
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, DEV
td = torch.bfloat16
M, N, K = 120, 128256, 2048
a = torch.randn(M, K, device=DEV, dtype=td)
ws = [torch.randn(N, K, device=DEV, dtype=td) * 0.02 for _ in range(3)]
out = torch.empty(M, N, device=DEV, dtype=torch.float32)
def run(i): gemm("bf16", a, ws[i % 3], C_out=out, out_f32=True)
for i in range(3): run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(12): run(i)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 12 * 1e3
print(f"lm_head M=120: {us:.1f} us  {N*K*2/us/1e3:.0f} GB/s")
ref = a.float() @ ws[2].float().T
print("max err", (out - ref).abs().max().item())
