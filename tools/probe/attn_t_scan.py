#!/usr/bin/env python3
"""Whisper-shaped flash attention (8 x T frames, 16 heads x 64) over T around 1500: which part of T = 1500 costs more than T = 1536?"""
import os, sys, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from omni_avsr_amd import lib as L
from gpu_util import DEV, stream, dcode
td = torch.bfloat16
lib = L.load()
def run(B, T, H=16, hd=64, iters=20, form=32):
    nq = H * hd
    qkv = torch.randn(B * T, 3 * nq, device=DEV, dtype=td)
    out = torch.empty(B * T, nq, device=DEV, dtype=td)
    cu = torch.arange(0, (B + 1) * T, T, dtype=torch.int32, device=DEV)
    d = L.AttnDesc()
    d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.n_seqs, d.max_q, d.causal = dcode("bf16"), hd, H, H, B, T, 0
    d.Q, d.K, d.V, d.O = qkv.data_ptr(), qkv.data_ptr() + nq * 2, qkv.data_ptr() + 2 * nq * 2, out.data_ptr()
    d.ldq = d.ldk = d.ldv = 3 * nq; d.ldo = nq
    d.cu_q, d.cu_k, d.scale, d.form = cu.data_ptr(), cu.data_ptr(), 1 / math.sqrt(hd), form
    for _ in range(3): L.check(lib.oavsr_op_attention(C.byref(d), stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.check(lib.oavsr_op_attention(C.byref(d), stream()))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"B={B} T={T:5d}: {us:8.1f} us  {4.0 * B * T * T * nq / us / 1e6:7.1f} TFLOP/s", flush=True)
run(8, 1536)
for T in (1280, 1408, 1472, 1480, 1500, 1504, 1535, 1536, 1600, 1792):
    run(8, T)
