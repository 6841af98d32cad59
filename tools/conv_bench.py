#!/usr/bin/env python3
"""The trunk's implicit-GEMM convolutions (resnet.py:61-74: layer2..4 of the ResNet-18 trunk on 3200 frames = 8 x 16 s) one by one through the C ABI,
HIP-event timed, inputs rotated beyond the Infinity Cache.   usage: conv_bench.py [frames=3200]   (OAVSR_CONV_TILE256=0: the 256 x 128 tile everywhere)"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import gemm, DEV, P, stream  # noqa: E402
from omni_avsr_amd import lib as L  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
td = torch.bfloat16
shapes = [("layer2 c1 s2", 64, 128, 22, 2, 3), ("layer2 down", 64, 128, 22, 2, 1), ("layer2 3x3", 128, 128, 11, 1, 3), ("layer3 c1 s2", 128, 256, 11, 2, 3),
          ("layer3 down", 128, 256, 11, 2, 1), ("layer3 3x3", 256, 256, 6, 1, 3), ("layer4 c1 s2", 256, 512, 6, 2, 3), ("layer4 down", 256, 512, 6, 2, 1), ("layer4 3x3", 512, 512, 3, 1, 3)]
tot = 0.0
for name, Cin, Cout, H, stride, ks in shapes:
    pad = ks // 2
    OH = (H + 2 * pad - ks) // stride + 1
    xs = [torch.randn(F, H, H, Cin, device=DEV, dtype=td) for _ in range(3)]
    w = (torch.randn(Cout, ks * ks * Cin, device=DEV) / math.sqrt(ks * ks * Cin)).to(td)
    scale, shift = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    out = torch.empty(F * OH * OH, Cout, device=DEV, dtype=td)
    run = lambda i: gemm("bf16", xs[i % 3], w, M=F * OH * OH, amode=1, H=H, W_=H, Cin=Cin, OH=OH, OW=OH, stride=stride, pad=pad, ksize=ks, lda=0, scale=scale, bias=shift, C_out=out)  # noqa: E731
    if Cin == 128 and Cout == 128 and ks == 3 and stride == 1 and os.environ.get("OAVSR_CONV_C128", "1") != "0":      # what the engine runs: the image-resident kernel
        name += "*"
        run = lambda i: L.check(L.load().oavsr_op_conv3x3_c128(P(xs[i % 3]), F, H, H, P(w), P(scale), P(shift), None, None, P(out), stream()))  # noqa: E731
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(12):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 12
    fl = 2.0 * F * OH * OH * Cout * ks * ks * Cin
    n = {"layer2 3x3": 3, "layer2 3x3*": 3, "layer3 3x3": 3, "layer4 3x3": 3}.get(name, 1)
    tot += us * n
    print(f"{name:14s} M={F * OH * OH:7d} N={Cout:4d} K={ks * ks * Cin:5d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  (x{n} per call)")
print(f"trunk layer2..4 per call: {tot / 1e3:.3f} ms")
