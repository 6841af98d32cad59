#!/usr/bin/env python3
"""Micro-benchmarks of single kernels through the C ABI (HIP events on torch's current stream).
Weights are rotated over enough copies to exceed the 256 MiB Infinity Cache, as in the real decode loop where every
weight byte is read once per step.   python tools/bench_kernels.py [decode|gemm|attn|all]"""
import math
import os
import sys
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from omni_avsr_amd import lib as L  # noqa: E402
from gpu_util import gemm, DEV, P, stream, dcode  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def bench_decode():
    """the decode-shaped launches of a step, per call site (tools/decode_mix.py --time)"""
    import subprocess
    for rows, model in ((8, "1b"), (97, "1b"), (97, "8b")):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "decode_mix.py"), str(rows), model, "1", "--time"], check=False)


def bench_gemm(dtype="bf16"):
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    print(f"--- dense GEMMs, {dtype} ---")
    for name, M, N, K in [("whisper qkv", 12000, 3072, 1024), ("whisper fc1", 12000, 4096, 1024), ("whisper fc2", 12000, 1024, 4096),
                          ("whisper o", 12000, 1024, 1024), ("avh qkv", 3200, 3072, 1024), ("llm qkv", 2504, 3072, 2048), ("llm gu", 2504, 16384, 2048),
                          ("llm down", 2504, 2048, 8192), ("square 4k", 4096, 4096, 4096), ("square 8k", 8192, 8192, 8192)]:
        a = torch.randn(M, K, device=DEV, dtype=td)
        w = torch.randn(N, K, device=DEV, dtype=td) * 0.02
        out = torch.empty(M, N, device=DEV, dtype=td)
        us = timeit(lambda i: gemm(dtype, a, w, C_out=out), iters=10)
        print(f"{name:12s} M={M:6d} N={N:6d} K={K:5d}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s")


def bench_attn(dtype="bf16"):
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    print(f"--- attention, {dtype} ---")
    # (the first entry is measured twice: the first kernel of a process runs ~15 % slow -- clocks / caches still cold -- which round 4 first
    #  mistook for a property of T = 1500)
    for name, B, T, H, KV, hd, causal in [("(warm-up)", 8, 1500, 16, 16, 64, 0), ("whisper", 8, 1500, 16, 16, 64, 0), ("whisper b16", 16, 1500, 16, 16, 64, 0), ("whisper 1536", 8, 1536, 16, 16, 64, 0), ("avhubert", 8, 400, 16, 16, 64, 0), ("llm prefill", 8, 313, 32, 8, 64, 1), ("long causal", 4, 4096, 32, 8, 64, 1)]:
        nq, nkv = H * hd, KV * hd
        qkv = torch.randn(B * T, nq + 2 * nkv, device=DEV, dtype=td)
        out = torch.empty(B * T, nq, device=DEV, dtype=td)
        cu = torch.arange(0, (B + 1) * T, T, dtype=torch.int32, device=DEV)
        d = L.AttnDesc()
        d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.n_seqs, d.max_q, d.causal = dcode(dtype), hd, H, KV, B, T, causal
        es = qkv.element_size()
        d.Q, d.K, d.V, d.O = qkv.data_ptr(), qkv.data_ptr() + nq * es, qkv.data_ptr() + (nq + nkv) * es, out.data_ptr()
        d.ldq = d.ldk = d.ldv = nq + 2 * nkv
        d.ldo = nq
        d.cu_q, d.cu_k, d.scale = cu.data_ptr(), cu.data_ptr(), 1 / math.sqrt(hd)
        lib = L.load()
        fl = 4.0 * B * T * T * nq / (2 if causal else 1)
        line = f"{name:12s} B={B} T={T} H={H}/{KV} hd={hd}:"
        for form in ((0, 32) if (hd == 64 and dtype == "bf16") else (0,)):
            d.form = form
            us = timeit(lambda i: L.check(lib.oavsr_op_attention(C.byref(d), stream())), iters=10)
            line += f"  form {form:2d}: {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s"
        print(line)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("decode", "all"):
        bench_decode()
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("attn", "all"):
        bench_attn()
