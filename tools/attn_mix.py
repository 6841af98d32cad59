#!/usr/bin/env python3
"""Whisper-shaped encoder attention (B=8, T=1500, 16 heads, hd 64, bf16) a few times: for `rocprofv3 --pmc` passes over attn_dma_kernel.
usage: attn_mix.py [reps]"""
import ctypes as C
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from omni_avsr_amd import lib as L
from gpu_util import DEV, stream, dcode
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B, T, H, KV, hd = 8, 1500, 16, 16, 64
nq, nkv = H * hd, KV * hd
qkv = torch.randn(B * T, nq + 2 * nkv, device=DEV, dtype=torch.bfloat16)
out = torch.empty(B * T, nq, device=DEV, dtype=torch.bfloat16)
cu = torch.arange(0, (B + 1) * T, T, dtype=torch.int32, device=DEV)
d = L.AttnDesc()
d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.n_seqs, d.max_q, d.causal = dcode("bf16"), hd, H, KV, B, T, 0
d.Q, d.K, d.V, d.O = qkv.data_ptr(), qkv.data_ptr() + nq * 2, qkv.data_ptr() + (nq + nkv) * 2, out.data_ptr()
d.ldq = d.ldk = d.ldv = nq + 2 * nkv
d.ldo = nq
d.cu_q, d.cu_k, d.scale = cu.data_ptr(), cu.data_ptr(), 1 / math.sqrt(hd)
lib = L.load()
for _ in range(reps):
    L.check(lib.oavsr_op_attention(C.byref(d), stream()))
torch.cuda.synchronize()
