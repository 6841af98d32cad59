#!/usr/bin/env python3
"""Launch the decode-attention kernel in the bench geometry (8 rows, 32/8 heads x 64, ~320 keys) with / without the fused RoPE
append, 300 times each; run under `rocprofv3 --kernel-trace --stats` to read the GPU-side duration.   VARIANT=rope|norope"""
import math, os, sys
import ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from omni_avsr_amd import lib as L
from omni_avsr_amd.weights import rope_inv_freq
from gpu_util import DEV, dcode, stream

variant = os.environ.get("VARIANT", "rope")
dtype, hd, H, KV = "bf16", 64, 32, 8
utts, maxp, maxt = 8, 330, 32
td = torch.bfloat16
ld = (H + 2 * KV) * hd
qkv = torch.randn(utts, ld, device=DEV, dtype=td)
kp = torch.randn(utts, KV, maxp, hd, device=DEV, dtype=td); vp = torch.randn_like(kp)
kt = torch.randn(utts, KV, maxt, hd, device=DEV, dtype=td); vt = torch.randn_like(kt)
inv = rope_inv_freq(hd, 500000.0, None).to(DEV)
step = 10
pl = torch.full((utts,), 313, dtype=torch.int32, device=DEV)
pos = (pl + step - 1).to(torch.int32)
cpos = torch.full((utts,), step - 1, dtype=torch.int32, device=DEV)
st = torch.tensor([step], dtype=torch.int32, device=DEV)
out = torch.empty(utts, H * hd, dtype=td, device=DEV)
d = L.DecodeAttnDesc()
d.dtype, d.head_dim, d.n_heads, d.n_kv_heads, d.rows, d.beams, d.max_prefix, d.max_tail = dcode(dtype), hd, H, KV, utts, 1, maxp, maxt
d.Q, d.Kp, d.Vp, d.Kt, d.Vt, d.O = qkv.data_ptr(), kp.data_ptr(), vp.data_ptr(), kt.data_ptr(), vt.data_ptr(), out.data_ptr()
d.ldq, d.ldo = ld, H * hd
d.prefix_len, d.step, d.anc, d.scale = pl.data_ptr(), st.data_ptr(), None, 1.0 / math.sqrt(hd)
if variant == "rope":
    d.inv_freq, d.pos, d.cpos = inv.data_ptr(), pos.data_ptr(), cpos.data_ptr()
lib = L.load()
for _ in range(300):
    L.check(lib.oavsr_op_decode_attention(C.byref(d), stream()))
torch.cuda.synchronize()
print("done", variant)
