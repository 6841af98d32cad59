"""Seeded synthetic checkpoints in the reference's state-dict layout (`AVSR_LLMs.state_dict()`).

There are no pretrained weights offline, so parity tests, smoke() and bench.py all run on weights drawn here:
tensor i (in creation order) comes from torch's CPU Philox/MT generator seeded with `seed + i`, N(0, std).
LoRA factors are BOTH non-zero (the reference zero-inits lora_down, which would hide LoRA bugs), BatchNorm
running stats are perturbed, PReLU slopes 0.25 +- 0.05, norm gains 1 +- 0.1.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import torch

from .config import OmniSpec


class _Gen:
    def __init__(self, seed: int, device="cpu", dtype=torch.float32):
        self.seed, self.i, self.device, self.dtype = seed, 0, torch.device(device), dtype

    def _g(self):
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed + self.i)
        self.i += 1
        return g

    def normal(self, *shape, std=0.02, mean=0.0):
        x = torch.randn(*shape, generator=self._g(), dtype=torch.float32, device=self.device)
        return x.mul_(std).add_(mean).to(self.dtype)

    def uniform(self, *shape, lo=0.0, hi=1.0):
        x = torch.rand(*shape, generator=self._g(), dtype=torch.float32, device=self.device)
        return x.mul_(hi - lo).add_(lo).to(self.dtype)


def _linear(sd, g, key, out_f, in_f, bias=True, std=None):
    std = std if std is not None else (1.0 / in_f) ** 0.5
    sd[key + ".weight"] = g.normal(out_f, in_f, std=std)
    if bias:
        sd[key + ".bias"] = g.normal(out_f, std=0.02)


def _ln(sd, g, key, dim):
    sd[key + ".weight"] = g.normal(dim, std=0.1, mean=1.0)
    sd[key + ".bias"] = g.normal(dim, std=0.05)


def _bn(sd, g, key, c):
    sd[key + ".weight"] = g.normal(c, std=0.1, mean=1.0)
    sd[key + ".bias"] = g.normal(c, std=0.1)
    sd[key + ".running_mean"] = g.normal(c, std=0.1)
    sd[key + ".running_var"] = g.uniform(c, lo=0.8, hi=1.2)
    sd[key + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long, device=g.device)


def synth_state_dict(spec: OmniSpec, seed: int = 1234, modalities=("audio", "video"), device="cpu",
                     dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """device='cuda' draws the tensors on the GPU (bench.py at full size; values then differ from the CPU stream).
    dtype=torch.bfloat16 halves the footprint when only the bf16 engine consumes the checkpoint."""
    g = _Gen(seed, device, dtype)
    sd: Dict[str, torch.Tensor] = OrderedDict()
    d = spec.llm.hidden
    # ---- audio_encoder.* : HF WhisperEncoder -------------------------------------------------------
    if spec.whisper is not None and "audio" in modalities:
        w = spec.whisper
        p = "audio_encoder."
        sd[p + "conv1.weight"] = g.normal(w.dim, 80, 3, std=(1.0 / 240) ** 0.5)
        sd[p + "conv1.bias"] = g.normal(w.dim, std=0.02)
        sd[p + "conv2.weight"] = g.normal(w.dim, w.dim, 3, std=(1.0 / (3 * w.dim)) ** 0.5)
        sd[p + "conv2.bias"] = g.normal(w.dim, std=0.02)
        sd[p + "embed_positions.weight"] = g.normal(1500, w.dim, std=0.1)
        for i in range(w.layers):
            lp = f"{p}layers.{i}."
            _linear(sd, g, lp + "self_attn.k_proj", w.dim, w.dim, bias=False)
            _linear(sd, g, lp + "self_attn.v_proj", w.dim, w.dim)
            _linear(sd, g, lp + "self_attn.q_proj", w.dim, w.dim)
            _linear(sd, g, lp + "self_attn.out_proj", w.dim, w.dim)
            _ln(sd, g, lp + "self_attn_layer_norm", w.dim)
            _linear(sd, g, lp + "fc1", w.ffn, w.dim)
            _linear(sd, g, lp + "fc2", w.dim, w.ffn)
            _ln(sd, g, lp + "final_layer_norm", w.dim)
        _ln(sd, g, p + "layer_norm", w.dim)
    # ---- video_encoder.* : fairseq AVHubertModel ----------------------------------------------------
    if spec.avhubert is not None and "video" in modalities:
        a = spec.avhubert
        p = "video_encoder."
        sd[p + "mask_emb"] = g.uniform(104)
        _linear(sd, g, p + "feature_extractor_audio.proj", a.dim, 104)
        r = p + "feature_extractor_video.resnet."
        sd[r + "frontend3D.0.weight"] = g.normal(64, 1, 5, 7, 7, std=(2.0 / 245) ** 0.5)
        _bn(sd, g, r + "frontend3D.1", 64)
        sd[r + "frontend3D.2.weight"] = g.normal(64, std=0.05, mean=0.25)
        chans = [64, 64, 128, 256, 512]
        for li in range(1, 5):
            for b in range(2):
                bp = f"{r}trunk.layer{li}.{b}."
                cin = chans[li - 1] if b == 0 else chans[li]
                cout = chans[li]
                sd[bp + "conv1.weight"] = g.normal(cout, cin, 3, 3, std=(2.0 / (9 * cin)) ** 0.5)
                _bn(sd, g, bp + "bn1", cout)
                sd[bp + "relu1.weight"] = g.normal(cout, std=0.05, mean=0.25)
                sd[bp + "relu2.weight"] = g.normal(cout, std=0.05, mean=0.25)
                sd[bp + "conv2.weight"] = g.normal(cout, cout, 3, 3, std=(1.0 / (9 * cout)) ** 0.5)
                _bn(sd, g, bp + "bn2", cout)
                if b == 0 and li > 1:
                    sd[bp + "downsample.0.weight"] = g.normal(cout, cin, 1, 1, std=(1.0 / cin) ** 0.5)
                    _bn(sd, g, bp + "downsample.1", cout)
        _linear(sd, g, p + "feature_extractor_video.proj", a.dim, 512)
        _ln(sd, g, p + "layer_norm", 2 * a.dim)
        _linear(sd, g, p + "post_extract_proj", a.dim, 2 * a.dim)
        cg = a.dim // a.conv_pos_groups
        sd[p + "encoder.pos_conv.0.bias"] = g.normal(a.dim, std=0.02)
        sd[p + "encoder.pos_conv.0.weight_g"] = g.uniform(1, 1, a.conv_pos, lo=0.5, hi=1.5)
        sd[p + "encoder.pos_conv.0.weight_v"] = g.normal(a.dim, cg, a.conv_pos, std=(4.0 / (a.conv_pos * a.dim)) ** 0.5)
        inner = int(round(a.dim / 16))
        for i in range(a.layers):
            lp = f"{p}encoder.layers.{i}."
            for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
                _linear(sd, g, lp + "self_attn." + nm, a.dim, a.dim)
            if a.use_lora:
                for nm in ("Q", "V"):
                    sd[f"{lp}self_attn.lora_down_{nm}.weight"] = g.normal(inner, a.dim, std=(1.0 / a.dim) ** 0.5)
                    sd[f"{lp}self_attn.lora_up_{nm}.weight"] = g.normal(a.dim, inner, std=0.5 * (1.0 / inner) ** 0.5)
            _ln(sd, g, lp + "self_attn_layer_norm", a.dim)
            _linear(sd, g, lp + "fc1", a.ffn, a.dim)
            _linear(sd, g, lp + "fc2", a.dim, a.ffn)
            _ln(sd, g, lp + "final_layer_norm", a.dim)
        _ln(sd, g, p + "encoder.layer_norm", a.dim)
        _linear(sd, g, p + "final_proj", 256, a.dim)
    # ---- projectors ----------------------------------------------------------------------------------
    for which, enc in (("audio", spec.whisper), ("video", spec.avhubert)):
        if enc is None or which not in modalities:
            continue
        rates = spec.rates(which)
        per_rate = spec.proj_per_rate()
        has_ln = spec.proj_has_layernorm(which)
        for idx, rate in enumerate(rates if per_rate else rates[:1]):
            pp = f"{which}_proj.{idx}." if per_rate else f"{which}_proj."
            din = enc.dim * (rate if spec.compression_mode == "stack" else 1)
            _linear(sd, g, pp + "0", spec.intermediate_size, din)
            _linear(sd, g, pp + "2", d, spec.intermediate_size)
            if has_ln:
                _ln(sd, g, pp + "3", d)
    # ---- llm.* -------------------------------------------------------------------------------------------
    l = spec.llm
    sd["llm.model.embed_tokens.weight"] = g.normal(l.vocab, d, std=0.02 * (2048 / d) ** 0.5 if d < 2048 else 0.02)
    inner, kvd = spec.lora_inner(), spec.kv_dim()
    tasks = ("audio", "video", "audiovisual")
    for i in range(l.layers):
        lp = f"llm.model.layers.{i}."
        _linear(sd, g, lp + "self_attn.q_proj", l.heads * l.head_dim, d, bias=l.qkv_bias)
        _linear(sd, g, lp + "self_attn.k_proj", kvd, d, bias=l.qkv_bias)
        _linear(sd, g, lp + "self_attn.v_proj", kvd, d, bias=l.qkv_bias)
        _linear(sd, g, lp + "self_attn.o_proj", d, l.heads * l.head_dim, bias=False)
        if spec.lora_rank:
            for nm, outd in (("Q", l.heads * l.head_dim), ("V", kvd)):
                names = [f".{t}" for t in tasks] if spec.is_task_specific else [""]
                for t in names:
                    sd[f"{lp}self_attn.lora_down_{nm}{t}.weight"] = g.normal(inner, d, std=(1.0 / d) ** 0.5)
                    sd[f"{lp}self_attn.lora_up_{nm}{t}.weight"] = g.normal(outd, inner, std=(1.0 / inner) ** 0.5)
                if spec.shared_lora:
                    sd[f"{lp}self_attn.lora_down_{nm}_shared.weight"] = g.normal(inner, d, std=(1.0 / d) ** 0.5)
                    sd[f"{lp}self_attn.lora_up_{nm}_shared.weight"] = g.normal(outd, inner, std=(1.0 / inner) ** 0.5)
        _linear(sd, g, lp + "mlp.gate_proj", l.mlp, d, bias=False)
        _linear(sd, g, lp + "mlp.up_proj", l.mlp, d, bias=False)
        _linear(sd, g, lp + "mlp.down_proj", d, l.mlp, bias=False)
        sd[lp + "input_layernorm.weight"] = g.normal(d, std=0.1, mean=1.0)
        sd[lp + "post_attention_layernorm.weight"] = g.normal(d, std=0.1, mean=1.0)
    sd["llm.model.norm.weight"] = g.normal(d, std=0.1, mean=1.0)
    if l.tie_embeddings:
        sd["llm.lm_head.weight"] = sd["llm.model.embed_tokens.weight"]
    else:
        sd["llm.lm_head.weight"] = g.normal(l.vocab, d, std=0.02)
    # ---- pre-embedded prompts (modeling_OmniAVSR.py:219-221) ---------------------------------------------
    for t in tasks:
        sd[f"prompt_{t}"] = g.normal(1, spec.prompt_len[t], d, std=0.02)
    return sd
