import torch, time
x = torch.randn(256*1024*1024//2, device="cuda", dtype=torch.bfloat16)  # 256 MB... use 1 GB
x = torch.randn(1024*1024*1024//2, device="cuda", dtype=torch.bfloat16)
def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
ms = t(lambda: x.sum())
print("sum 1GB bf16: %.1f us  %.0f GB/s" % (ms*1e3, x.numel()*2/ms/1e6))
xf = x.view(torch.float32)
ms = t(lambda: xf.sum())
print("sum 1GB f32: %.1f us  %.0f GB/s" % (ms*1e3, x.numel()*2/ms/1e6))
y = torch.empty_like(x)
ms = t(lambda: y.copy_(x))
print("copy 1GB: %.1f us  %.0f GB/s (r+w)" % (ms*1e3, 2*x.numel()*2/ms/1e6))
ms = t(lambda: xf.max())
print("max 1GB f32: %.1f us  %.0f GB/s" % (ms*1e3, x.numel()*2/ms/1e6))
