"""Checkpoint ingestion: reference state-dict (`model_avg_N.pth` layout) -> one 256-byte
aligned device blob + manifest of packed tensors that libomniavsr_hip.so consumes (include/omni_avsr.h).

Re-layouts done here, once, on the host (pure data movement + the algebra the reference does at run time):
  * conv weights -> [C_out][k_y][k_x][C_in] (implicit-GEMM K order); stem -> [64][5*7*7 padded to 256]
  * BatchNorm(eval) -> per-channel (scale, shift) applied in the GEMM epilogue (resnet.py:45,55,139)
  * weight-norm of the positional conv folded: w = g * v / ||v||  (wav2vec2.py:837)
  * q/k/v (and gate/up) concatenated row-wise so one GEMM produces them
  * LoRA kept UNMERGED: A matrices stacked into one down-projection, scale*B placed in a block
    matrix that the GEMM consumes as a second K segment -> (W x + s B A x) in one accumulator
  * norm gains, biases, BN params stay fp32; GEMM operands are stored in the compute dtype (bf16 or f32)
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

from .config import OmniSpec

ALIGN = 256
TASKS = ("audio", "video", "audiovisual")


def rope_inv_freq(head_dim: int, theta: float, scaling) -> torch.Tensor:
    """HF rotary init ('default' / 'llama3' rope_type)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    if not scaling:
        return inv
    factor, lo, hi = scaling["factor"], scaling["low_freq_factor"], scaling["high_freq_factor"]
    old = scaling["original_max_position_embeddings"]
    wavelen = 2 * math.pi / inv
    inv_l = torch.where(wavelen > old / lo, inv / factor, inv)
    smooth = (old / wavelen - lo) / (hi - lo)
    smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
    medium = ~(wavelen < old / hi) * ~(wavelen > old / lo)
    return torch.where(medium, smoothed, inv_l)


def _pad_cols(w: torch.Tensor, k: int) -> torch.Tensor:
    if w.shape[1] == k:
        return w
    out = torch.zeros(w.shape[0], k, dtype=w.dtype, device=w.device)
    out[:, : w.shape[1]] = w
    return out


def _bn_fold(sd, key: str, eps: float = 1e-5) -> Tuple[torch.Tensor, torch.Tensor]:
    scale = sd[key + ".weight"].float() / torch.sqrt(sd[key + ".running_var"].float() + eps)
    shift = sd[key + ".bias"].float() - sd[key + ".running_mean"].float() * scale
    return scale, shift


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_state_dict(sd: Dict[str, torch.Tensor], spec: OmniSpec, dtype: str = "bf16") -> Dict[str, torch.Tensor]:
    """Returns {packed_name: tensor}; GEMM operands in `dtype`, everything else fp32."""
    T = torch.bfloat16 if dtype == "bf16" else torch.float32
    out: Dict[str, torch.Tensor] = {}
    dev = next(iter(sd.values())).device

    def W(name, t):           # GEMM operand
        out[name] = t.to(T).contiguous()

    def F(name, t):           # fp32 side parameter
        out[name] = t.float().contiguous()


    def enc_layer(dst: str, src: str, attn: str, dim: int, lora_k2: int, lora_scale: float, k_bias: bool):
        q, k, v = (sd[f"{src}{attn}{n}_proj.weight"].float() for n in "qkv")
        W(dst + "qkv.w", torch.cat([q, k, v], 0))
        kb = sd[f"{src}{attn}k_proj.bias"].float() if k_bias else torch.zeros(dim, device=dev)
        F(dst + "qkv.b", torch.cat([sd[f"{src}{attn}q_proj.bias"].float(), kb, sd[f"{src}{attn}v_proj.bias"].float()]))
        if lora_k2:
            aq, av = sd[f"{src}{attn}lora_down_Q.weight"].float(), sd[f"{src}{attn}lora_down_V.weight"].float()
            bq, bv = sd[f"{src}{attn}lora_up_Q.weight"].float(), sd[f"{src}{attn}lora_up_V.weight"].float()
            r = aq.shape[0]
            dn = torch.zeros(lora_k2, dim, device=dev)
            dn[:r], dn[r:2 * r] = aq, av
            up = torch.zeros(3 * dim, lora_k2, device=dev)
            up[:dim, :r] = bq * lora_scale
            up[2 * dim:, r:2 * r] = bv * lora_scale
            W(dst + "lora_dn.w", dn)
            W(dst + "lora_up.w", up)
        W(dst + "o.w", sd[f"{src}{attn}out_proj.weight"])
        F(dst + "o.b", sd[f"{src}{attn}out_proj.bias"])
        F(dst + "ln1.w", sd[src + "self_attn_layer_norm.weight"]); F(dst + "ln1.b", sd[src + "self_attn_layer_norm.bias"])
        F(dst + "ln2.w", sd[src + "final_layer_norm.weight"]); F(dst + "ln2.b", sd[src + "final_layer_norm.bias"])
        W(dst + "fc1.w", sd[src + "fc1.weight"]); F(dst + "fc1.b", sd[src + "fc1.bias"])
        W(dst + "fc2.w", sd[src + "fc2.weight"]); F(dst + "fc2.b", sd[src + "fc2.bias"])

    # ---------------- video encoder (fairseq AVHubertModel keys) ----------------
    if spec.avhubert is not None and "video_encoder.layer_norm.weight" in sd:
        a = spec.avhubert
        p = "video_encoder."
        r = p + "feature_extractor_video.resnet."
        W("v.stem.w", _pad_cols(sd[r + "frontend3D.0.weight"].float().reshape(64, 245), 256))
        s, b = _bn_fold(sd, r + "frontend3D.1")
        F("v.stem.scale", s); F("v.stem.shift", b); F("v.stem.slope", sd[r + "frontend3D.2.weight"])
        for li in range(1, 5):
            for bi in range(2):
                bp = f"{r}trunk.layer{li}.{bi}."
                dp = f"v.l{li}.{bi}."
                for cn, bn, relu in (("conv1", "bn1", "relu1"), ("conv2", "bn2", "relu2")):
                    w = sd[bp + cn + ".weight"].float()
                    W(dp + cn + ".w", w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))
                    s, b = _bn_fold(sd, bp + bn)
                    F(dp + cn + ".scale", s); F(dp + cn + ".shift", b); F(dp + cn + ".slope", sd[bp + relu + ".weight"])
                if bp + "downsample.0.weight" in sd:
                    w = sd[bp + "downsample.0.weight"].float()
                    W(dp + "down.w", w.reshape(w.shape[0], w.shape[1]))
                    s, b = _bn_fold(sd, bp + "downsample.1")
                    F(dp + "down.scale", s); F(dp + "down.shift", b)
        W("v.proj.w", sd[p + "feature_extractor_video.proj.weight"]); F("v.proj.b", sd[p + "feature_extractor_video.proj.bias"])
        F("v.ln.w", sd[p + "layer_norm.weight"]); F("v.ln.b", sd[p + "layer_norm.bias"])
        W("v.post.w", sd[p + "post_extract_proj.weight"]); F("v.post.b", sd[p + "post_extract_proj.bias"])
        g, v = sd[p + "encoder.pos_conv.0.weight_g"].float(), sd[p + "encoder.pos_conv.0.weight_v"].float()
        w = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()             # [D][Cg][taps]
        W("v.pos.w", w.permute(0, 2, 1).reshape(w.shape[0], -1))               # k = tap*Cg + c
        F("v.pos.b", sd[p + "encoder.pos_conv.0.bias"])
        k2 = round_up(2 * int(round(a.dim / 16)), 64) if a.use_lora else 0
        for i in range(a.layers):
            enc_layer(f"v.L{i}.", f"{p}encoder.layers.{i}.", "self_attn.", a.dim, k2, 2.0, True)
        F("v.lnf.w", sd[p + "encoder.layer_norm.weight"]); F("v.lnf.b", sd[p + "encoder.layer_norm.bias"])
    # ---------------- audio encoder (HF WhisperEncoder keys) ----------------
    if spec.whisper is not None and "audio_encoder.conv1.weight" in sd:
        w_ = spec.whisper
        p = "audio_encoder."
        c1 = sd[p + "conv1.weight"].float()                                     # [D][80][3]
        W("a.conv1.w", _pad_cols(c1.permute(0, 2, 1).reshape(c1.shape[0], 240), 256))
        F("a.conv1.b", sd[p + "conv1.bias"])
        c2 = sd[p + "conv2.weight"].float()
        W("a.conv2.w", c2.permute(0, 2, 1).reshape(c2.shape[0], -1))
        F("a.conv2.b", sd[p + "conv2.bias"])
        W("a.pos", sd[p + "embed_positions.weight"])
        for i in range(w_.layers):
            enc_layer(f"a.L{i}.", f"{p}layers.{i}.", "self_attn.", w_.dim, 0, 0.0, False)
        F("a.lnf.w", sd[p + "layer_norm.weight"]); F("a.lnf.b", sd[p + "layer_norm.bias"])
    # ---------------- projectors ----------------
    for which, tag in (("audio", "pa"), ("video", "pv")):
        n = len(spec.rates(which)) if spec.proj_per_rate() else 1
        for idx in range(n):
            pp = f"{which}_proj.{idx}." if spec.proj_per_rate() else f"{which}_proj."
            if pp + "0.weight" not in sd:
                continue
            W(f"{tag}.{idx}.fc1.w", sd[pp + "0.weight"]); F(f"{tag}.{idx}.fc1.b", sd[pp + "0.bias"])
            W(f"{tag}.{idx}.fc2.w", sd[pp + "2.weight"]); F(f"{tag}.{idx}.fc2.b", sd[pp + "2.bias"])
            if pp + "3.weight" in sd:
                F(f"{tag}.{idx}.ln.w", sd[pp + "3.weight"]); F(f"{tag}.{idx}.ln.b", sd[pp + "3.bias"])
    # ---------------- LLM ----------------
    l = spec.llm
    d, nq, kvd = l.hidden, l.heads * l.head_dim, spec.kv_dim()
    W("l.embed", sd["llm.model.embed_tokens.weight"])
    head = sd.get("llm.lm_head.weight", sd["llm.model.embed_tokens.weight"])
    if head.data_ptr() == sd["llm.model.embed_tokens.weight"].data_ptr():
        out["l.lm_head"] = out["l.embed"]                                       # tied: one copy in the blob
    else:
        W("l.lm_head", head)
    F("l.norm.w", sd["llm.model.norm.weight"])
    F("l.inv_freq", rope_inv_freq(l.head_dim, l.rope_theta, l.rope_scaling).to(dev))
    inner, scale = spec.lora_inner(), spec.lora_scale()
    n_ad = (2 if spec.shared_lora else 1)
    k2 = round_up(2 * n_ad * inner, 64) if inner else 0
    for i in range(l.layers):
        sp, dp = f"llm.model.layers.{i}.", f"l.L{i}."
        ap = sp + "self_attn."
        W(dp + "qkv.w", torch.cat([sd[ap + f"{n}_proj.weight"].float() for n in "qkv"], 0))
        if l.qkv_bias:
            F(dp + "qkv.b", torch.cat([sd[ap + f"{n}_proj.bias"].float() for n in "qkv"]))
        W(dp + "o.w", sd[ap + "o_proj.weight"])
        gw, uw = sd[sp + "mlp.gate_proj.weight"].float(), sd[sp + "mlp.up_proj.weight"].float()
        # gate / up rows interleaved in blocks of 16 (g0 u0 g1 u1 ...): one MFMA lane then holds gate_j and up_j of the
        # same column, so SwiGLU is a register-local epilogue
        W(dp + "gu.w", torch.stack([gw.reshape(-1, 16, d), uw.reshape(-1, 16, d)], 1).reshape(2 * l.mlp, d))
        W(dp + "down.w", sd[sp + "mlp.down_proj.weight"])
        F(dp + "ln1.w", sd[sp + "input_layernorm.weight"]); F(dp + "ln2.w", sd[sp + "post_attention_layernorm.weight"])
        if inner:
            variants = [(str(t), f".{name}") for t, name in enumerate(TASKS)] if spec.is_task_specific else [("x", "")]
            for tag, suffix in variants:
                dn = torch.zeros(k2, d, device=dev)
                up = torch.zeros(nq + 2 * kvd, k2, device=dev)
                col = 0
                for which, row0, rows in (("Q", 0, nq), ("V", nq + kvd, kvd)):
                    srcs = [suffix] + (["_shared"] if spec.shared_lora else [])
                    for s_ in srcs:
                        a_ = sd[f"{ap}lora_down_{which}{s_}.weight"].float()
                        b_ = sd[f"{ap}lora_up_{which}{s_}.weight"].float()
                        dn[col:col + inner] = a_
                        up[row0:row0 + rows, col:col + inner] = b_ * scale
                        col += inner
                W(f"{dp}lora_dn.{tag}.w", dn)
                W(f"{dp}lora_up.{tag}.w", up)
    for t, name in enumerate(TASKS):
        if f"prompt_{name}" in sd:
            W(f"l.prompt.{t}", sd[f"prompt_{name}"][0])
    return out


def build_blob(packed: Dict[str, torch.Tensor], device="cpu") -> Tuple[torch.Tensor, List[Tuple[str, int, int]]]:
    """Lays the packed tensors out in one uint8 buffer (256-byte aligned).  Tensors that share storage
    (tied lm_head) are stored once.  Returns (blob, [(name, offset, nbytes)])."""
    offsets: Dict[int, int] = {}
    manifest: List[Tuple[str, int, int]] = []
    total = 0
    for name, t in packed.items():
        key = t.data_ptr()
        nbytes = t.numel() * t.element_size()
        if key not in offsets:
            total = (total + ALIGN - 1) // ALIGN * ALIGN
            offsets[key] = total
            total += nbytes
        manifest.append((name, offsets[key], nbytes))
    total = (total + ALIGN - 1) // ALIGN * ALIGN
    blob = torch.zeros(total, dtype=torch.uint8, device=device)
    done = set()
    for name, t in packed.items():
        key = t.data_ptr()
        if key in done:
            continue
        done.add(key)
        off = offsets[key]
        nbytes = t.numel() * t.element_size()
        blob[off:off + nbytes].copy_(t.reshape(-1).view(torch.uint8), non_blocking=False)
    return blob, manifest


def lora_k2(spec: OmniSpec) -> int:
    inner = spec.lora_inner()
    return round_up(2 * (2 if spec.shared_lora else 1) * inner, 64) if inner else 0


def av_lora_k2(spec: OmniSpec) -> int:
    a = spec.avhubert
    return round_up(2 * int(round(a.dim / 16)), 64) if (a is not None and a.use_lora) else 0
