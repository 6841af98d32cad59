#!/usr/bin/env python3
"""The decode-shaped launches of ONE decode step (llm_decode_layers + lm_head, model.hip) at R rows for a model's widths, through the C ABI:
per layer LoRA-down, QKV (+ LoRA segment, row_scale), o_proj (+ residual, next-norm outputs), gate / up (+ SwiGLU, row_scale), down_proj
(+ residual, next-norm outputs) -- split-K exactly as decode_gemm sizes it (<= 16 rows: the weight-streaming strips; reduce inside the launch up to 32 rows, as a second launch above) --, the
decode attention on the same rows, then lm_head (fp32 logits).
Weight copies rotate beyond the 256 MiB Infinity Cache.  For `rocprofv3 --pmc` / `--kernel-trace` passes over the decode kernels that
ship (tools/pmc_round.py tags "decode..."), and with `--time` a per-site timing table.

usage: decode_mix.py [rows=97] [model=1b|8b|7b] [steps=2] [--time]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C  # noqa: E402
import math  # noqa: E402
import torch  # noqa: E402
from omni_avsr_amd import lib as L  # noqa: E402
from omni_avsr_amd.weights import rope_inv_freq  # noqa: E402
from gpu_util import gemm, DEV, stream  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
R = int(args[0]) if args else 97
model = args[1] if len(args) > 1 else "1b"
steps = int(args[2]) if len(args) > 2 else 2
timing = "--time" in sys.argv
# d, nqkv, nq, 2I, lora k2, vocab, layers
d, nqkv, nq, I2, k2, V, layers = {"1b": (2048, 3072, 2048, 16384, 256, 128261, 16), "8b": (4096, 6144, 4096, 28672, 512, 128261, 32),
                                  "7b": (3584, 4608, 3584, 37888, 448, 152068, 28)}[model]
td = torch.bfloat16


def split(N, K, min_ks=1):      # model.hip: decode_split
    ktiles = K // 64
    ks = 256 // max(1, -(-N // 64))
    ks = min(ks, max(1, ktiles // 8))
    ks = max(1, min(ks, 16))
    return max(ks, min(min_ks, max(1, ktiles)))


sites = [("lora_dn", k2, d, 0, 1, False), ("qkv", nqkv, d, k2, 1, False), ("o_proj", d, nq, 0, 2, True), ("gate_up", I2, d, 0, 1, False), ("down", d, I2 // 2, 0, 2, True)]
ncopy = 3 if model == "1b" else 2
W = {n: [torch.randn(N, K, device=DEV, dtype=td) * 0.02 for _ in range(ncopy)] for n, N, K, K2, mk, ng in sites}
W2 = [torch.randn(nqkv, k2, device=DEV, dtype=td) * 0.02 for _ in range(ncopy)]
Wl = torch.randn(V, d, device=DEV, dtype=td) * 0.02
x = torch.randn(R, d, device=DEV, dtype=td)
xg = torch.randn(R, d, device=DEV, dtype=td)
t = torch.randn(R, k2, device=DEV, dtype=td)
attn = torch.randn(R, nq, device=DEV, dtype=td)
act = torch.randn(R, I2 // 2, device=DEV, dtype=td)
qkv = torch.empty(R, nqkv, device=DEV, dtype=td)
logits = torch.empty(R, V, device=DEV, dtype=torch.float32)
gain = torch.ones(d, device=DEV)
rowsq = torch.rand((d // 64 + 2) * 128, device=DEV) * 64
STRIP = R <= 16 and os.environ.get("OAVSR_DECODE_STRIPS", "1") != "0"      # model.hip: decode_gemm takes the strips (decode_strip.hip) for <= 16 rows
RS = dict(row_sumsq=rowsq, row_sumsq_tiles=d // 64, row_dim=d, row_eps=1e-5, row_sumsq_form=int(STRIP))
NG = dict(rowsq_strips=1) if STRIP else {}
part = torch.empty(16 * 128 * max(nqkv, d, k2), device=DEV, dtype=torch.float32)
sync = torch.zeros(L.GEMM_SYNC_INTS, dtype=torch.int32, device=DEV) if R <= 32 else None      # model.hip: OAVSR_FUSED_REDUCE_ROWS


# decode attention (RoPE + KV append fused) on the same rows: 313 prompt keys (a 16 s AVSR clip at rates 4 / 4), step 10, one KV copy per layer slot
hd = 64 if model == "1b" else 128
Hh, KVh = nq // hd, (nqkv - nq) // (2 * hd)
nkv = 4 if model == "1b" else 2
kp = [torch.randn(R, KVh, 320, hd, device=DEV, dtype=td) for _ in range(nkv)]
vp = [torch.randn_like(kp[0]) for _ in range(nkv)]
kt = torch.randn(R, KVh, 32, hd, device=DEV, dtype=td)
vt = torch.randn_like(kt)
inv = rope_inv_freq(hd, 500000.0, None).to(DEV)
pl = torch.full((R,), 313, dtype=torch.int32, device=DEV)
pos = (pl + 9).to(torch.int32)
cpos = torch.full((R,), 9, dtype=torch.int32, device=DEV)
stp = torch.tensor([10], dtype=torch.int32, device=DEV)
ad = L.DecodeAttnDesc()
ad.dtype, ad.head_dim, ad.n_heads, ad.n_kv_heads, ad.rows, ad.beams, ad.max_prefix, ad.max_tail = L.DT_BF16, hd, Hh, KVh, R, 1, 320, 32
ad.Q, ad.Kt, ad.Vt, ad.O = qkv.data_ptr(), kt.data_ptr(), vt.data_ptr(), attn.data_ptr()
ad.ldq, ad.ldo = nqkv, nq
ad.prefix_len, ad.step, ad.anc, ad.scale = pl.data_ptr(), stp.data_ptr(), None, 1.0 / math.sqrt(hd)
ad.inv_freq, ad.pos, ad.cpos = inv.data_ptr(), pos.data_ptr(), cpos.data_ptr()
lib = L.load()


def launch(name, i):
    if name == "attention":
        ad.Kp, ad.Vp = kp[i % nkv].data_ptr(), vp[i % nkv].data_ptr()
        L.check(lib.oavsr_op_decode_attention(C.byref(ad), stream()))
        return
    w = W[name][i % ncopy] if name != "lm_head" else Wl
    if name == "lora_dn":
        gemm("bf16", xg, w, C_out=t, ksplit=split(k2, d), partial=part, sync=sync)
    elif name == "qkv":
        ks = split(nqkv, d + k2)
        kw = dict(ksplit=ks, partial=part, sync=sync) if ks > 1 else dict(decode_tiled=1)
        gemm("bf16", xg, w, A2=t, W2=W2[i % ncopy], C_out=qkv, **RS, **kw)
    elif name == "o_proj":
        gemm("bf16", attn, w, residual=x, C_out=x, ksplit=split(d, nq, 2), partial=part, sync=sync, ng_out=xg, ng_ld=d, ng_gain=gain, ng_rowsq=rowsq, **NG)
    elif name == "gate_up":
        gemm("bf16", xg, w, swiglu_pairs=1, C_out=act, decode_tiled=1, **RS)
    elif name == "down":
        gemm("bf16", act, w, residual=x, C_out=x, ksplit=split(d, I2 // 2, 2), partial=part, sync=sync, ng_out=xg, ng_ld=d, ng_gain=gain, ng_rowsq=rowsq, **NG)
    else:
        gemm("bf16", xg, w, out_f32=True, C_out=logits, decode_tiled=1, **RS)


if timing:
    for name in [s[0] for s in sites] + ["attention", "lm_head"]:
        for i in range(3):
            launch(name, i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 24
        e0.record()
        for i in range(n):
            launch(name, i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        if name == "attention":
            kvb = R * 322 * KVh * hd * 2 * 2
            print(f"{model} rows={R:3d} attention {Hh}/{KVh} x {hd}, 313 + 9 keys: {us:7.1f} us  {kvb / us / 1e3:7.1f} GB/s of K / V")
            continue
        N, K = {"lora_dn": (k2, d), "qkv": (nqkv, d + k2), "o_proj": (d, nq), "gate_up": (I2, d), "down": (d, I2 // 2), "lm_head": (V, d)}[name]
        print(f"{model} rows={R:3d} {name:8s} N={N:6d} K={K:5d} split={split(N, K, 2 if name in ('o_proj', 'down') else 1) if name not in ('gate_up', 'lm_head') else 1:2d}: {us:7.1f} us  {N * K * 2 / us / 1e3:7.1f} GB/s of weights")
else:
    for s in range(steps):
        for li in range(layers):
            for name in ["lora_dn", "qkv", "attention", "o_proj", "gate_up", "down"]:
                launch(name, li)
        launch("lm_head", 0)
    torch.cuda.synchronize()
