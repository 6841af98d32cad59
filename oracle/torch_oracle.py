"""TEST INFRASTRUCTURE ONLY — CPU fp32 restatement of the Omni-AVSR inference hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product path (`omni_avsr_amd`) never does: it fails loudly if the HIP library is
missing.

Every function cites the reference file:line it restates (paths relative to /root/reference).
Pinning:
  * video path (a-3..a-6): checked against the *real* vendored classes (AVHubertModel,
    ResEncoder, TransformerEncoder, MultiheadAttention.forward_lora) by
    `oracle/gen_golden.py`, whose outputs are committed under tests/golden/.
  * Whisper / Llama / Qwen2 / generate (third-party `transformers==4.43.1`, not vendored in the
    reference, requirements.txt:7): restated from the published algorithm and cross-checked
    against the installed transformers (5.15) classes by `oracle/gen_golden.py`.
    The reference's own tests hold no vector for this path => for these parts
    parity is pinned to transformers-5.15 outputs, not to reference golden files.

All weights come in a flat dict `sd` keyed exactly like `AVSR_LLMs.state_dict()`
(SURVEY.md §5.4), all math is fp32 on CPU.

bf16 emulation (`with bf16_emulation(): ...`): the same restatement with every tensor the bf16 engine keeps in
bf16 rounded to bf16 at the point where the engine stores it (GEMM operands / weights, each kernel's output after its
fused epilogue, soft-max probabilities before P.V, rotated q / k, the residual stream), fp32 accumulation everywhere else --
the arithmetic of the reference's `bf16-true` evaluation (eval_OmniAVSR.py:19) with this engine's fusion boundaries.  It is
the yardstick the bf16 engine is held to at a tight bound (tests/test_bf16_emulation_gpu.py); outside the context manager
`_q` is the identity and nothing below changes by a bit.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------


class _Precision:
    emu = False


_P = _Precision()


class bf16_emulation:
    """Context manager: round to bf16 wherever the bf16 engine stores bf16 (see the module docstring)."""

    def __enter__(self):
        self.prev, _P.emu = _P.emu, True
        return self

    def __exit__(self, *a):
        _P.emu = self.prev
        if not _P.emu:
            _QW.clear()
        return False


def _q(x: Tensor) -> Tensor:
    """Storage rounding of the bf16 engine; identity in fp32 mode."""
    return x.bfloat16().float() if _P.emu else x


_QW: Dict[int, tuple] = {}


def _qw(w: Tensor) -> Tensor:
    """_q(w.float()) for a WEIGHT of the state dict: in emulation mode the rounded copy is kept for the lifetime of the outermost
    `bf16_emulation()` block (keyed by the tensor object, which the entry keeps alive), so a decode loop does not re-round 1.2 B
    parameters on every step.  Same values as _q(w.float())."""
    if not _P.emu:
        return w.float()
    ent = _QW.get(id(w))
    if ent is None or ent[0] is not w:
        ent = (w, w.float().bfloat16().float())
        _QW[id(w)] = ent
    return ent[1]


def _ln(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """torch.nn.LayerNorm in fp32 (fairseq/modules/layer_norm.py:29-34)."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def _gelu(x: Tensor) -> Tensor:
    """erf-GELU in fp32 (fairseq/modules/gelu.py:25-26)."""
    return F.gelu(x.float())


def _linear(x: Tensor, sd: Dict[str, Tensor], key: str) -> Tensor:
    """GEMM operand W in the compute dtype, bias fp32; the caller rounds the result where the engine's fused epilogue ends."""
    w = _qw(sd[key + ".weight"])
    b = sd.get(key + ".bias")
    return F.linear(x, w, None if b is None else b.float())


def _softmax_pv(w: Tensor, v: Tensor) -> Tensor:
    """softmax(w) @ v.  Emulation: the flash kernels round the un-normalised exponentials to bf16 for the P.V MFMA and keep
    the row sum in fp32 (attn.hip)."""
    if not _P.emu:
        return F.softmax(w.float(), dim=-1) @ v
    e = torch.exp(w - w.amax(dim=-1, keepdim=True))
    return (_q(e) @ v) / e.sum(dim=-1, keepdim=True)


# ----------------------------------------------------------------------------------------------
# a-4  ResNet-18 visual front-end  (av_hubert/avhubert/resnet.py)
# ----------------------------------------------------------------------------------------------


def _bn_eval(x: Tensor, sd, key: str, eps: float = 1e-5) -> Tensor:
    """BatchNorm in eval mode == per-channel affine (resnet.py:139, :45, :55)."""
    w, b = sd[key + ".weight"].float(), sd[key + ".bias"].float()
    rm, rv = sd[key + ".running_mean"].float(), sd[key + ".running_var"].float()
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - rm.view(shape)) / torch.sqrt(rv.view(shape) + eps) * w.view(shape) + b.view(shape)


def _prelu(x: Tensor, a: Tensor) -> Tensor:
    shape = [1, -1] + [1] * (x.dim() - 2)
    a = a.float().view(shape)
    return torch.where(x >= 0, x, a * x)


def resnet_frontend3d(sd, p: str, video: Tensor) -> Tensor:
    """resnet.py:137-141,158: Conv3d(1,64,(5,7,7),(1,2,2),(2,3,3)) -> BN3d -> PReLU -> MaxPool3d.
    video: [B,1,T,88,88] -> [B,64,T,22,22]"""
    x = F.conv3d(_q(video.float()), _qw(sd[p + "frontend3D.0.weight"]), None, stride=(1, 2, 2), padding=(2, 3, 3))
    x = _bn_eval(x, sd, p + "frontend3D.1")
    x = _prelu(x, sd[p + "frontend3D.2.weight"])
    x = F.max_pool3d(x, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
    return _q(x)


def resnet_basic_block(sd, p: str, x: Tensor, stride: int, has_down: bool) -> Tensor:
    """resnet.py:61-74 BasicBlock.forward with PReLU."""
    residual = x
    out = F.conv2d(x, _qw(sd[p + "conv1.weight"]), None, stride=stride, padding=1)
    out = _bn_eval(out, sd, p + "bn1")
    out = _q(_prelu(out, sd[p + "relu1.weight"]))
    out = F.conv2d(out, _qw(sd[p + "conv2.weight"]), None, stride=1, padding=1)
    out = _bn_eval(out, sd, p + "bn2")
    if has_down:  # resnet.py:20-24: 1x1 conv stride s + BN
        residual = F.conv2d(x, _qw(sd[p + "downsample.0.weight"]), None, stride=stride)
        residual = _q(_bn_eval(residual, sd, p + "downsample.1"))
    out = out + residual
    return _q(_prelu(out, sd[p + "relu2.weight"]))


def resnet_trunk(sd, p: str, x: Tensor, return_stages: bool = False):
    """resnet.py:121-129: layer1..4 ([2,2,2,2] BasicBlocks) -> AdaptiveAvgPool2d(1) -> flatten."""
    stages = []
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = resnet_basic_block(sd, f"{p}layer{li}.0.", x, stride, has_down=(li != 1))
        x = resnet_basic_block(sd, f"{p}layer{li}.1.", x, 1, has_down=False)
        stages.append(x)
    x = _q(x.mean(dim=(2, 3)))
    return (x, stages) if return_stages else x


def res_encoder(sd, p: str, video: Tensor) -> Tensor:
    """ResEncoder.forward resnet.py:156-164. video [B,1,T,88,88] -> [B,512,T]"""
    B, _, T = video.shape[:3]
    x = resnet_frontend3d(sd, p, video)                      # [B,64,T,22,22]
    x = x.transpose(1, 2).reshape(B * T, 64, x.shape[3], x.shape[4])   # resnet.py:166-169
    x = resnet_trunk(sd, p + "trunk.", x)                    # [B*T,512]
    return x.view(B, T, -1).transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------------------------
# a-5 / a-6  wav2vec2 TransformerEncoder + MultiheadAttention (forward / forward_lora)
# ----------------------------------------------------------------------------------------------


def weight_norm_fold(g: Tensor, v: Tensor) -> Tensor:
    """nn.utils.weight_norm(conv, name='weight', dim=2) (wav2vec2.py:837):
    w = g * v / ||v|| with the norm taken over every dim except 2; g: [1,1,K], v: [C,C/g,K]."""
    norm = v.float().pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return g.float() * v.float() / norm


def fairseq_mha(sd, p: str, x: Tensor, heads: int, lora: bool, scaling_lora: float = 2.0) -> Tensor:
    """multihead_attention.py:389-672 (forward_lora) or :160-192 (stock forward; same mathematics
    with the LoRA terms absent).  x: [T,B,C]; no masks (the caller passes none,
    modeling_OmniAVSR.py:463)."""
    T, B, C = x.shape
    hd = C // heads
    q = _linear(x, sd, p + "q_proj")
    k = _linear(x, sd, p + "k_proj")
    v = _linear(x, sd, p + "v_proj")
    if lora and not _P.emu:  # :485-494
        q = q + F.linear(F.linear(x, sd[p + "lora_down_Q.weight"].float()), sd[p + "lora_up_Q.weight"].float()) * scaling_lora
        v = v + F.linear(F.linear(x, sd[p + "lora_down_V.weight"].float()), sd[p + "lora_up_V.weight"].float()) * scaling_lora
    elif lora:               # engine: t = x A^T stored in bf16, the scale folded into the stored up-projection, one fp32 accumulator
        q = q + F.linear(_q(F.linear(x, _qw(sd[p + "lora_down_Q.weight"]))), _q(sd[p + "lora_up_Q.weight"].float() * scaling_lora))
        v = v + F.linear(_q(F.linear(x, _qw(sd[p + "lora_down_V.weight"]))), _q(sd[p + "lora_up_V.weight"].float() * scaling_lora))
    q, k, v = _q(q), _q(k), _q(v)
    if not _P.emu:
        q = q * (hd ** -0.5)                                 # :511 (after LoRA)
    q = q.contiguous().view(T, B * heads, hd).transpose(0, 1)
    k = k.contiguous().view(T, B * heads, hd).transpose(0, 1)
    v = v.contiguous().view(T, B * heads, hd).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))                      # :619
    if _P.emu:
        w = w * (hd ** -0.5)                                 # engine: the scale multiplies the fp32 scores
    a = _q(_softmax_pv(w, v))                                # :647-654 fp32 softmax, bmm
    a = a.transpose(0, 1).contiguous().view(T, B, C)
    return _linear(a, sd, p + "out_proj")


def w2v_transformer_encoder(sd, p: str, x: Tensor, layers: int, heads: int, conv_pos: int, groups: int,
                            lora: bool, return_layers: bool = False):
    """TransformerEncoder.forward / extract_features wav2vec2.py:862-905 with layer_norm_first=True
    (large_vox_iter5.yaml:77). x: [B,T,C]."""
    w = _q(weight_norm_fold(sd[p + "pos_conv.0.weight_g"], sd[p + "pos_conv.0.weight_v"]))
    xc = F.conv1d(x.transpose(1, 2), w, sd[p + "pos_conv.0.bias"].float(), padding=conv_pos // 2, groups=groups)
    if conv_pos % 2 == 0:                                     # SamePad same_pad.py:16-21
        xc = xc[:, :, :-1]
    x = _q(x + _gelu(xc).transpose(1, 2))                     # :875-877
    x = x.transpose(0, 1)                                     # T,B,C
    outs = []
    for i in range(layers):                                   # layer: wav2vec2.py:977-1006
        lp = f"{p}layers.{i}."
        res = x
        h = _q(_ln(x, sd[lp + "self_attn_layer_norm.weight"], sd[lp + "self_attn_layer_norm.bias"]))
        h = fairseq_mha(sd, lp + "self_attn.", h, heads, lora)
        x = _q(res + h)
        res = x
        h = _q(_ln(x, sd[lp + "final_layer_norm.weight"], sd[lp + "final_layer_norm.bias"]))
        h = _q(_gelu(_linear(h, sd, lp + "fc1")))
        h = _linear(h, sd, lp + "fc2")
        x = _q(res + h)
        outs.append(x.transpose(0, 1))
    x = x.transpose(0, 1)
    x = _q(_ln(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"]))   # :865-866
    return (x, outs) if return_layers else x


def avhubert_extract_finetune(sd, p: str, video: Tensor, cfg: dict, return_stages: bool = False):
    """AVHubertModel.extract_finetune hubert.py:695-755, video-only branch (:706-709).
    video: [B,1,T,88,88] -> [B,T,D]."""
    D = cfg["embed"]
    feat = res_encoder(sd, p + "feature_extractor_video.resnet.", video)            # [B,512,T]
    fv = _q(_linear(feat.transpose(1, 2), sd, p + "feature_extractor_video.proj"))  # SubModel hubert.py:327
    fa = torch.zeros_like(fv)                                                         # :709
    feats = torch.cat([fa, fv], dim=-1)                                               # :715 (audio first)
    feats = _q(_ln(feats, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"]))  # :721
    x0 = _q(_linear(feats, sd, p + "post_extract_proj"))                              # :728
    x = w2v_transformer_encoder(sd, p + "encoder.", x0, cfg["layers"], cfg["heads"], cfg["conv_pos"],
                                cfg["conv_pos_groups"], cfg.get("use_lora", False), return_layers=return_stages)
    if return_stages:
        return x[0], {"resnet": feat, "front": x0, "layers": x[1]}
    return x


# ----------------------------------------------------------------------------------------------
# a-2  Whisper log-mel + encoder  (transformers WhisperFeatureExtractor / WhisperEncoder;
#      call sites modeling_OmniAVSR.py:531-537)
# ----------------------------------------------------------------------------------------------


def slaney_mel_filters(n_freqs: int = 201, n_mels: int = 80, sr: int = 16000, fmin=0.0, fmax=8000.0) -> np.ndarray:
    """Slaney-scale, slaney-normalised triangular filterbank [n_freqs, n_mels] — the published
    librosa/`transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney')` construction."""
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        mel = 3.0 * f / 200.0
        log_region = f >= 1000.0
        mel = np.where(log_region, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4)), mel)
        return mel

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f = 200.0 * m / 3.0
        log_region = m >= 15.0
        f = np.where(log_region, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), f)
        return f

    fft_freqs = np.linspace(0, sr // 2, n_freqs)
    mel_pts = np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2)
    filt_freqs = mel_to_hz(mel_pts)
    fdiff = np.diff(filt_freqs)
    slopes = filt_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filt_freqs[2:n_mels + 2] - filt_freqs[:n_mels])
    fb = fb * enorm[None, :]
    return fb.astype(np.float32)


def whisper_log_mel(wave: Tensor, n_samples: int = 480000) -> Tensor:
    """WhisperFeatureExtractor.__call__ (torch path, used by transformers==4.43.1 whenever torch is
    installed): zero-pad/truncate to 30 s, STFT(n_fft 400, hop 160, periodic Hann, center, reflect),
    |.|^2, drop last frame, slaney mels, log10(clamp 1e-10), clamp(max-8) per utterance, (x+4)/4.
    wave: [B,T_s] fp32 -> [B,80,3000]."""
    B, Ts = wave.shape
    x = torch.zeros(B, n_samples, dtype=torch.float32)
    n = min(Ts, n_samples)
    x[:, :n] = wave[:, :n].float()
    win = torch.hann_window(400, periodic=True, dtype=torch.float32)
    st = torch.stft(x, 400, 160, window=win, center=True, pad_mode="reflect", return_complex=True)
    mag = st[..., :-1].abs() ** 2                             # [B,201,3000]
    fb = torch.from_numpy(slaney_mel_filters())               # [201,80]
    mel = fb.T @ mag                                          # [B,80,3000]
    ls = torch.clamp(mel, min=1e-10).log10()
    mx = ls.amax(dim=(1, 2), keepdim=True)
    ls = torch.maximum(ls, mx - 8.0)
    return (ls + 4.0) / 4.0


def whisper_encoder(sd, p: str, mel: Tensor, layers: int, heads: int, return_layers: bool = False):
    """transformers WhisperEncoder.forward: conv1(k3,p1)+GELU, conv2(k3,s2,p1)+GELU, +pos-emb,
    pre-LN layers (k_proj without bias, q scaled by head_dim^-0.5), final LN.  mel: [B,80,3000] -> [B,1500,D]."""
    x = _q(_gelu(F.conv1d(_q(mel.float()), _qw(sd[p + "conv1.weight"]), sd[p + "conv1.bias"].float(), padding=1)))
    x = _gelu(F.conv1d(x, _qw(sd[p + "conv2.weight"]), sd[p + "conv2.bias"].float(), stride=2, padding=1))
    x = _q(x.permute(0, 2, 1) + _qw(sd[p + "embed_positions.weight"])[None])
    B, T, C = x.shape
    hd = C // heads
    outs = []
    for i in range(layers):
        lp = f"{p}layers.{i}."
        h = _q(_ln(x, sd[lp + "self_attn_layer_norm.weight"], sd[lp + "self_attn_layer_norm.bias"]))
        q = _linear(h, sd, lp + "self_attn.q_proj")
        if not _P.emu:
            q = q * (hd ** -0.5)
        k = F.linear(h, _qw(sd[lp + "self_attn.k_proj.weight"]))
        v = _linear(h, sd, lp + "self_attn.v_proj")
        q = _q(q).view(B, T, heads, hd).transpose(1, 2)
        k = _q(k).view(B, T, heads, hd).transpose(1, 2)
        v = _q(v).view(B, T, heads, hd).transpose(1, 2)
        w = q @ k.transpose(-1, -2)
        if _P.emu:
            w = w * (hd ** -0.5)                              # engine: the scale multiplies the fp32 scores
        a = _q(_softmax_pv(w, v)) if _P.emu else F.softmax(w, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, C)
        x = _q(x + _linear(a, sd, lp + "self_attn.out_proj"))
        h = _q(_ln(x, sd[lp + "final_layer_norm.weight"], sd[lp + "final_layer_norm.bias"]))
        x = _q(x + _linear(_q(_gelu(_linear(h, sd, lp + "fc1"))), sd, lp + "fc2"))
        outs.append(x)
    x = _q(_ln(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"]))
    return (x, outs) if return_layers else x


# ----------------------------------------------------------------------------------------------
# a-7  compressor + projector  (modeling_OmniAVSR.py:480-483,507-514,555-558,581-588; ctor :86-111,171-196)
# ----------------------------------------------------------------------------------------------


def compress(x: Tensor, rate: int, mode: str) -> Tensor:
    """avg-pooling: AvgPool1d(kernel=stride=rate) over time (floor; remainder dropped) :480-483.
    stack: concat `rate` consecutive frames on the channel axis, drop remainder :507-514."""
    B, T, C = x.shape
    n = T // rate
    x = x[:, : n * rate].reshape(B, n, rate, C)
    if mode == "avg-pooling":
        return _q(x.mean(dim=2))
    if mode == "stack":
        return x.reshape(B, n, rate * C)
    raise ValueError(mode)


def projector(sd, p: str, x: Tensor) -> Tensor:
    """Sequential(Linear, ReLU, Linear[, LayerNorm]) — index 3 exists only in the variants where the
    LayerNorm really is a Sequential member (modeling_OmniAVSR.py:102,111 vs the :104 quirk where it
    is swallowed as the Linear `bias` arg and never applied)."""
    h = _q(F.relu(_linear(x, sd, p + "0")))
    h = _q(_linear(h, sd, p + "2"))
    if p + "3.weight" in sd:
        h = _q(_ln(h, sd[p + "3.weight"], sd[p + "3.bias"]))
    return h


# ----------------------------------------------------------------------------------------------
# a-9  Llama / Qwen2 decoder with Omni-LoRA  (Omni_AVSR/Llama_LoRA.py, Qwen_LoRA.py)
# ----------------------------------------------------------------------------------------------


def rope_inv_freq(head_dim: int, theta: float, scaling: Optional[dict]) -> Tensor:
    """transformers rope init: default, or 'llama3' (factor / low_freq_factor / high_freq_factor /
    original_max_position_embeddings) as used by Llama-3.1/3.2 configs."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    if scaling is None:
        return inv
    factor, lo, hi, old = scaling["factor"], scaling["low_freq_factor"], scaling["high_freq_factor"], scaling["original_max_position_embeddings"]
    low_w, high_w = old / lo, old / hi
    wavelen = 2 * math.pi / inv
    inv_l = torch.where(wavelen > low_w, inv / factor, inv)
    smooth = (old / wavelen - lo) / (hi - lo)
    smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
    is_medium = ~(wavelen < high_w) * ~(wavelen > low_w)
    return torch.where(is_medium, smoothed, inv_l)


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _rms(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """LlamaRMSNorm / Qwen2RMSNorm: fp32 variance, weight * normalised."""
    xf = x.float()
    return w.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps))


class LLMOracle:
    """Functional Llama-3.x / Qwen2.5 decoder with hand-rolled LoRA on Q and V.
    Restates LlamaSdpaAttention_lora.forward (Llama_LoRA.py:244-316), LlamaDecoderLayer_lora
    (:587-655), LlamaModel_lora (:455-578), LlamaForCausalLM_lora.forward (:328-398) and the
    Qwen twins (Qwen_LoRA.py:529-619).  cfg keys: hidden, layers, heads, kv_heads, head_dim, mlp,
    rms_eps, rope_theta, rope_scaling, qkv_bias, lora_rank (reduction factor), lora_alpha,
    task_specific, shared."""

    def __init__(self, sd: Dict[str, Tensor], cfg: dict, prefix: str = "llm."):
        self.sd, self.cfg, self.p = sd, cfg, prefix
        self.inv_freq = rope_inv_freq(cfg["head_dim"], cfg["rope_theta"], cfg.get("rope_scaling"))
        self.scaling = cfg["lora_alpha"] / cfg["lora_rank"] if cfg.get("lora_rank") else 0.0   # Llama_LoRA.py:120
    def _lora(self, lp: str, which: str, x: Tensor, modality: Optional[str], rstd: Optional[Tensor] = None) -> Tensor:
        sd, cfg = self.sd, self.cfg
        if cfg.get("task_specific"):                           # Llama_LoRA.py:250-251
            dn, up = f"{lp}lora_down_{which}.{modality}.weight", f"{lp}lora_up_{which}.{modality}.weight"
        else:
            dn, up = f"{lp}lora_down_{which}.weight", f"{lp}lora_up_{which}.weight"
        pairs = [(dn, up)]
        if cfg.get("shared"):                                  # :254-259
            pairs.append((f"{lp}lora_down_{which}_shared.weight", f"{lp}lora_up_{which}_shared.weight"))
        if not _P.emu:
            out = F.linear(F.linear(x, sd[pairs[0][0]].float()), sd[pairs[0][1]].float())
            for d_, u_ in pairs[1:]:
                out = out + F.linear(F.linear(x, sd[d_].float()), sd[u_].float())
            return out * self.scaling
        # engine: t = x A^T stored in bf16, alpha / rank folded into the stored up-projection, all adapters summed in the QKV GEMM's fp32
        # accumulator.  Decode-shaped rows (rstd given): x is T(x * gain), t is rounded WITHOUT 1 / rms, and the accumulator -- this term
        # and the base projection alike -- is multiplied by 1 / rms afterwards (model.hip: llm_decode_layers)
        out = 0
        for d_, u_ in pairs:
            out = out + F.linear(_q(F.linear(x, _qw(sd[d_]))), _q(sd[u_].float() * self.scaling))
        return out * rstd if rstd is not None else out

    def embed(self, ids: Tensor) -> Tensor:
        return _q(self.sd[self.p + "model.embed_tokens.weight"].float()[ids])

    def forward(self, x: Tensor, pos0: int, cache: Optional[list], modality: Optional[str],
                last_only: bool = False, return_hidden: bool = False):
        """x: [B,S,d] embeddings, positions pos0..pos0+S-1, cache: list of (k,v) per layer
        ([B,kvh,ctx,hd]) appended in place.  Returns fp32 logits [B,S,V] (Llama_LoRA.py:372-373).
        bf16 emulation only: decode-shaped rows -- every decode step (S == 1, any number of rows) and, in a prefill with `last_only`, the
        last layer's MLP and lm_head on each utterance's final row -- follow the engine's deferred RMSNorm: the producer of the residual
        stream stores T(x * gain) and 1 / rms(x), the consumer GEMM multiplies its fp32 accumulator by the latter."""
        sd, cfg, p = self.sd, self.cfg, self.p
        B, S, d = x.shape
        H, KV, hd = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
        emu = _P.emu
        dec = S == 1                                         # a decode step: decode-shaped GEMMs throughout, whatever B is
        tail = last_only and S > 1                           # prefill: the engine prunes the LAST layer to each utterance's final row
        pos = torch.arange(pos0, pos0 + S, dtype=torch.float32)
        freqs = pos[:, None] * self.inv_freq[None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        cos, sin = emb.cos()[None, None], emb.sin()[None, None]
        x = _q(x.float())
        hiddens = []

        def fused_norm(xx, gain):        # deferred RMSNorm (gemm2.hip ng_* outputs / ops.hip rms_prep_kernel): (T(x * g), 1 / rms(x))
            return _q(xx * gain.float()), torch.rsqrt(xx.pow(2).mean(-1, keepdim=True) + cfg["rms_eps"])

        for i in range(cfg["layers"]):
            lp = f"{p}model.layers.{i}."
            ap = lp + "self_attn."
            last = i == cfg["layers"] - 1
            if emu and dec:
                h, rstd = fused_norm(x, sd[lp + "input_layernorm.weight"])
                bias = lambda n: (sd[ap + n + ".bias"].float() if ap + n + ".bias" in sd else 0.0)    # noqa: E731
                q = F.linear(h, _qw(sd[ap + "q_proj.weight"])) * rstd + bias("q_proj")
                k = F.linear(h, _qw(sd[ap + "k_proj.weight"])) * rstd + bias("k_proj")
                v = F.linear(h, _qw(sd[ap + "v_proj.weight"])) * rstd + bias("v_proj")
                if cfg.get("lora_rank"):
                    q = q + self._lora(ap, "Q", h, modality, rstd)
                    v = v + self._lora(ap, "V", h, modality, rstd)
            else:
                h = _q(_rms(x, sd[lp + "input_layernorm.weight"], cfg["rms_eps"]))
                q = _linear(h, sd, ap + "q_proj")
                k = _linear(h, sd, ap + "k_proj")
                v = _linear(h, sd, ap + "v_proj")
                if cfg.get("lora_rank"):
                    q = q + self._lora(ap, "Q", h, modality)
                    v = v + self._lora(ap, "V", h, modality)
            q = _q(q).view(B, S, H, hd).transpose(1, 2)
            k = _q(k).view(B, S, KV, hd).transpose(1, 2)
            v = _q(v).view(B, S, KV, hd).transpose(1, 2)
            q = _q(q * cos + _rot_half(q) * sin)               # apply_rotary_pos_emb (:277)
            k = _q(k * cos + _rot_half(k) * sin)
            if cache is not None:
                if cache[i] is not None:
                    k = torch.cat([cache[i][0], k], dim=2)
                    v = torch.cat([cache[i][1], v], dim=2)
                cache[i] = (k, v)
            ctx = k.shape[2]
            kk = k.repeat_interleave(H // KV, dim=1)           # repeat_kv :284-285
            vv = v.repeat_interleave(H // KV, dim=1)
            w = (q @ kk.transpose(-1, -2)) / math.sqrt(hd)
            if S > 1:                                          # causal (:298)
                qi = torch.arange(ctx - S, ctx)[:, None]
                ki = torch.arange(ctx)[None, :]
                w = w.masked_fill(ki > qi, float("-inf"))
            a = _q(_softmax_pv(w, vv)) if emu else F.softmax(w, dim=-1) @ vv
            a = a.transpose(1, 2).reshape(B, S, H * hd)
            x = _q(x + _linear(a, sd, ap + "o_proj"))
            # the engine prunes the LAST prefill layer to each utterance's final row and runs its MLP through the decode kernels
            if emu and (dec or (last and tail)):
                h, rstd = fused_norm(x, sd[lp + "post_attention_layernorm.weight"])
                g = F.silu(_q(F.linear(h, _qw(sd[lp + "mlp.gate_proj.weight"])) * rstd)) * _q(F.linear(h, _qw(sd[lp + "mlp.up_proj.weight"])) * rstd)
            else:
                h = _q(_rms(x, sd[lp + "post_attention_layernorm.weight"], cfg["rms_eps"]))
                g = F.silu(_q(_linear(h, sd, lp + "mlp.gate_proj"))) * _q(_linear(h, sd, lp + "mlp.up_proj"))
            x = _q(x + _linear(_q(g), sd, lp + "mlp.down_proj"))
            if return_hidden:
                hiddens.append(x)
        if last_only:
            x = x[:, -1:]
        W = _qw(sd[p + "lm_head.weight"])
        if emu and (dec or tail):                              # lm_head on decode-shaped rows: the final RMSNorm deferred the same way
            h, rstd = fused_norm(x, sd[p + "model.norm.weight"])
            logits = (F.linear(h, W) * rstd).float()
        else:
            x = _q(_rms(x, sd[p + "model.norm.weight"], cfg["rms_eps"]))
            logits = F.linear(x, W).float()
        return (logits, hiddens) if return_hidden else logits

    # -- generation ---------------------------------------------------------------------------

    def generate_greedy(self, embeds: Tensor, max_new: int, eos: int, pad: int, modality=None,
                        force_len: bool = False, return_logits: bool = False, timing: Optional[dict] = None):
        """HF greedy search (`generate(num_beams=1, do_sample=False)`), one utterance per row with its
        OWN prompt length handled by the caller (B=1 in the reference, data_module.py:238-242).
        embeds: [1,S,d].  Returns list of new token ids (EOS included, as HF returns it)."""
        assert embeds.shape[0] == 1
        import time as _time
        cache = [None] * self.cfg["layers"]
        t0 = _time.time()
        logits = self.forward(embeds, 0, cache, modality, last_only=True)[:, -1]
        if timing is not None:                                  # bench.py's CPU baseline: seconds of the prefill and of every step
            timing["prefill"], timing["steps"] = _time.time() - t0, []
        S = embeds.shape[1]
        out, all_logits = [], []
        for step in range(max_new):
            t0 = _time.time()
            if return_logits:
                all_logits.append(logits[0].clone())
            lg = logits[0].clone()
            if force_len:
                lg[eos] = float("-inf")
            tok = int(torch.argmax(lg))
            out.append(tok)
            if tok == eos and not force_len:
                break
            if step + 1 == max_new:
                break
            logits = self.forward(self.embed(torch.tensor([[tok]])), S + step, cache, modality, last_only=True)[:, -1]
            if timing is not None:
                timing["steps"].append(_time.time() - t0)
        return (out, all_logits) if return_logits else out

    def score_path(self, embeds: Tensor, tokens: List[int], modality=None) -> Tensor:
        """Teacher forcing: the fp32 logits [len(tokens), V] the model assigns at every step when it is fed `tokens` (the path
        another implementation decoded).  `logits[t, tokens[t]]` against `logits[t].max()` says how far that implementation's
        choice is from this oracle's arg-max -- a tolerance-aware id check for reduced-precision runs."""
        assert embeds.shape[0] == 1
        cache = [None] * self.cfg["layers"]
        S = embeds.shape[1]
        logits = self.forward(embeds, 0, cache, modality, last_only=True)[:, -1]
        out = [logits[0]]
        for step, tok in enumerate(tokens[:-1]):
            logits = self.forward(self.embed(torch.tensor([[int(tok)]])), S + step, cache, modality, last_only=True)[:, -1]
            out.append(logits[0])
        return torch.stack(out)

    def generate_beam(self, embeds: Tensor, max_new: int, num_beams: int, eos: int, pad: int, modality=None,
                      length_penalty: float = 1.0, reference_prefill: bool = False, timing: Optional[dict] = None) -> List[int]:
        """transformers==4.43.1 `GenerationMixin._beam_search` + `BeamSearchScorer` with
        do_sample=False, early_stopping=False, length_penalty=1.0 and `inputs_embeds` only
        (=> input_ids start empty, decoder_prompt_len = 0); call site modeling_OmniAVSR.py:313-322.
        One utterance.  Returns the best hypothesis' new tokens (EOS appended when there is room)."""
        assert embeds.shape[0] == 1
        K, V = num_beams, self.sd[self.p + "lm_head.weight"].shape[0]
        S = embeds.shape[1]
        import time as _time
        cache = [None] * self.cfg["layers"]
        t0 = _time.time()
        if reference_prefill:
            # exactly what the reference executes (bench.py's CPU baseline): HF expands the prompt K-fold
            # (`_expand_inputs_for_generation`) and Llama_LoRA.py:372-373 computes fp32 logits of EVERY position
            logits = self.forward(embeds.repeat(K, 1, 1), 0, cache, modality, last_only=False)[:, -1]
        else:
            # the K rows are identical, so run the prompt once and replicate the cache: same values
            logits = self.forward(embeds, 0, cache, modality, last_only=True)[:, -1]
            logits = logits.repeat(K, 1)
            for i in range(len(cache)):
                cache[i] = (cache[i][0].repeat(K, 1, 1, 1), cache[i][1].repeat(K, 1, 1, 1))
        if timing is not None:
            timing["prefill"], timing["steps"] = _time.time() - t0, []
        beam_scores = torch.full((K,), -1e9, dtype=torch.float32)
        beam_scores[0] = 0.0
        seqs: List[List[int]] = [[] for _ in range(K)]
        hyps: List[Tuple[float, List[int]]] = []              # (score, tokens)
        worst = 1e9
        done = False
        cur_len = 0
        while True:
            t0 = _time.time()
            logp = F.log_softmax(logits, dim=-1) + beam_scores[:, None]
            top_s, top_i = torch.topk(logp.view(-1), 2 * K, largest=True, sorted=True)
            nb_scores, nb_tokens, nb_idx = [], [], []
            for rank in range(2 * K):
                tok = int(top_i[rank]) % V
                bi = int(top_i[rank]) // V
                sc = float(top_s[rank])
                if tok == eos:
                    if rank >= K:
                        continue
                    gen_len = cur_len + 1                      # cur_len(process) = input_ids.shape[-1] + 1
                    score = sc / (gen_len ** length_penalty)
                    if len(hyps) < K or score > worst:
                        hyps.append((score, list(seqs[bi])))
                        if len(hyps) > K:
                            srt = sorted(range(len(hyps)), key=lambda j: hyps[j][0])
                            del hyps[srt[0]]
                            worst = min(h[0] for h in hyps)
                        else:
                            worst = min(score, worst)
                else:
                    nb_scores.append(sc)
                    nb_tokens.append(tok)
                    nb_idx.append(bi)
                if len(nb_scores) == K:
                    break
            # BeamHypotheses.is_done (early_stopping=False)
            if len(hyps) >= K:
                highest = float(top_s[0]) / ((cur_len + 1) ** length_penalty)
                done = done or (worst >= highest)
            seqs = [seqs[bi] + [t] for bi, t in zip(nb_idx, nb_tokens)]
            beam_scores = torch.tensor(nb_scores, dtype=torch.float32)
            cur_len += 1
            if done or cur_len >= max_new:
                break
            idx = torch.tensor(nb_idx)
            for i in range(len(cache)):
                cache[i] = (cache[i][0][idx], cache[i][1][idx])
            x = self.embed(torch.tensor(nb_tokens)[:, None])
            logits = self.forward(x, S + cur_len - 1, cache, modality, last_only=True)[:, -1]
            if timing is not None:
                timing["steps"].append(_time.time() - t0)
        if not done:                                           # finalize: add all running beams
            for b in range(K):
                score = float(beam_scores[b]) / (len(seqs[b]) ** length_penalty)
                if len(hyps) < K or score > worst:
                    hyps.append((score, list(seqs[b])))
                    if len(hyps) > K:
                        srt = sorted(range(len(hyps)), key=lambda j: hyps[j][0])
                        del hyps[srt[0]]
                        worst = min(h[0] for h in hyps)
                    else:
                        worst = min(score, worst)
        best = sorted(hyps, key=lambda h: h[0])[-1][1]
        if len(best) < max_new:
            best = best + [eos]
        return best


# ----------------------------------------------------------------------------------------------
# a-1 / a-8  full forward (inference branch)  modeling_OmniAVSR.py:308-323, 397-458
# ----------------------------------------------------------------------------------------------


class OmniOracle:
    """CPU fp32 twin of `AVSR_LLMs.forward(inputs, is_trainval=False, ...)`.

    cfg: dict with keys llm (dict for LLMOracle), family ('llama'|'qwen2'), whisper {layers,heads},
    avhubert {embed,layers,heads,conv_pos,conv_pos_groups,use_lora}, compression_mode,
    is_matryoshka, is_single_matry_projector, rates_audio, rates_video, ids {bos,eos,pad,audio_sos,
    audio_eos,video_sos,video_eos}, max_new, num_beams."""

    def __init__(self, sd: Dict[str, Tensor], cfg: dict):
        self.sd, self.cfg = sd, cfg
        self.llm = LLMOracle(sd, cfg["llm"])

    # modeling_OmniAVSR.py:528-558
    def encode_audio(self, audio: Tensor, max_len: int, rate: int, return_stages=False, mel_bf16: bool = False):
        """`mel_bf16`: the reference casts the log-mel features to bf16 whatever the precision plugin (`.to(torch.bfloat16)`, :534);
        the engine's f32 parity mode keeps them in fp32, its bf16 mode stores them in bf16 (= the emulation's `_q`)."""
        # emulation: bf16-true hands forward() the waveform in bf16 (Lightning's HalfPrecision.convert_input; `audio.to(torch.float32)`, :530)
        mel = whisper_log_mel(_q(audio.squeeze(-1).float()))
        if mel_bf16:
            mel = mel.bfloat16().float()
        enc = whisper_encoder(self.sd, "audio_encoder.", mel, self.cfg["whisper"]["layers"], self.cfg["whisper"]["heads"])
        n_keep = max(int(max_len / 16000 * 50), 25)            # :537
        enc_k = enc[:, :n_keep]
        out = compress(enc_k, rate, self.cfg["compression_mode"]) if rate != 1 else enc_k
        return (out, {"mel": mel, "enc": enc}) if return_stages else out

    # modeling_OmniAVSR.py:461-483
    def encode_video(self, video: Tensor, rate: int, return_stages=False):
        v = video.float().reshape(-1, video.shape[2], video.shape[1], video.shape[3], video.shape[4])  # :463
        enc = avhubert_extract_finetune(self.sd, "video_encoder.", v, self.cfg["avhubert"])
        out = compress(enc, rate, self.cfg["compression_mode"]) if rate != 1 else enc
        return (out, {"enc": enc}) if return_stages else out

    def _proj_prefix(self, which: str, rate: int) -> str:
        c = self.cfg
        if c.get("is_matryoshka") and not c.get("is_single_matry_projector"):
            rates = c[f"rates_{which}"]
            return f"{which}_proj.{rates.index(rate)}."       # matry_map_* :68-70,429,447 (KeyError if absent)
        return f"{which}_proj."

    # modeling_OmniAVSR.py:397-458
    def prepare_inputs(self, inputs: dict, modality: str, rate_a: int, rate_v: int, mel_bf16: bool = False, feats=None) -> Tensor:
        """`feats` = (a_feat, v_feat): compressed encoder outputs the caller already has from encode_audio / encode_video on the same inputs
        (tests that also compare the encoder stages: the 24-layer encoders then run once, not twice)."""
        c, ids = self.cfg, self.cfg["ids"]
        B = inputs["tokens"].shape[0]
        E = self.llm.embed
        if feats is not None:
            a_feat, v_feat = feats
        else:
            a_feat = self.encode_audio(inputs["audio"], int(max(inputs["lengths"])), rate_a, mel_bf16=mel_bf16) if modality in ("audio", "audiovisual") else None
            v_feat = self.encode_video(inputs["video"], rate_v) if modality in ("video", "audiovisual") else None
        prompt = _qw(self.sd[f"prompt_{modality}"]).expand(B, -1, -1)
        if c["family"] == "qwen2":
            text = prompt                                      # :417
        else:
            text = torch.cat([E(inputs["tokens"])[:, :1], prompt], dim=1)   # :419
        for feat, which, sos, eos in ((v_feat, "video", ids["video_sos"], ids["video_eos"]),
                                      (a_feat, "audio", ids["audio_sos"], ids["audio_eos"])):
            if feat is None:
                continue
            rate = rate_v if which == "video" else rate_a
            f = projector(self.sd, self._proj_prefix(which, rate), feat)
            blk = torch.cat([E(torch.full((B, 1), sos)), f, E(torch.full((B, 1), eos))], dim=1)
            if c["family"] == "qwen2":
                text = torch.cat([blk, text], dim=1)           # :434,452
            else:
                text = torch.cat([text[:, :1], blk, text[:, 1:]], dim=1)   # :436-437,454-455
        return text

    def forward(self, inputs: dict, modality: str, rate_a: Optional[int] = None, rate_v: Optional[int] = None,
                num_beams: Optional[int] = None, force_len: bool = False) -> List[List[int]]:
        """Per-utterance results identical to running the reference with batch 1 per utterance
        (its only inference mode, data_module.py:238-242)."""
        c = self.cfg
        K = c["num_beams"] if num_beams is None else num_beams
        outs = []
        B = inputs["tokens"].shape[0]
        for b in range(B):
            one = {k: (v[b:b + 1] if isinstance(v, torch.Tensor) else v) for k, v in inputs.items()}
            if "video_lens" in inputs:                         # var-len batches: trim the padding
                one["video"] = one["video"][:, : int(inputs["video_lens"][b])]
            if "lengths" in inputs:
                one["audio"] = one["audio"][:, : int(inputs["lengths"][b])]
                one["lengths"] = inputs["lengths"][b:b + 1]
            emb = self.prepare_inputs(one, modality, rate_a, rate_v)
            mod = modality if c["llm"].get("task_specific") else None
            if K == 1:
                ids = self.llm.generate_greedy(emb, c["max_new"], c["ids"]["eos"], c["ids"]["pad"], mod, force_len=force_len)
            else:
                ids = self.llm.generate_beam(emb, c["max_new"], K, c["ids"]["eos"], c["ids"]["pad"], mod)
            outs.append(ids)
        return outs
