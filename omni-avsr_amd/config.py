"""Model dimensions and the reference's config objects.

`LoRA_config` / `QwenLoRA_config` keep the field names of Omni_AVSR/Llama_LoRA.py:103-110 and
Omni_AVSR/Qwen_LoRA.py:92-103 so caller code written against the reference constructs them unchanged.
The HF / fairseq dimension tables restate the public model cards (the reference pulls them from the hub /
from the pickled hydra cfg; neither is reachable offline).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional


@dataclass
class LoRA_config:                      # Llama_LoRA.py:103-110
    RANK: int
    ALPHA: int = 1
    IS_LLAMA3: bool = False
    IS_LLAMA3_2_3B: bool = False
    IS_TASK_SPECIFIC: bool = False
    SHARED_LORA: bool = False


@dataclass
class QwenLoRA_config:                  # Qwen_LoRA.py:92-103
    RANK: int
    ALPHA: int = 1
    IS_QWEN25_0_5B: bool = False
    IS_QWEN25_1_5B: bool = False
    IS_QWEN25_3B: bool = False
    IS_QWEN25_7B: bool = False
    IS_QWEN25_14B: bool = False
    IS_QWEN25_32B: bool = False
    IS_TASK_SPECIFIC: bool = False
    SHARED_LORA: bool = False


@dataclass
class LLMSpec:
    family: str                 # "llama" | "qwen2"
    hidden: int
    layers: int
    heads: int
    kv_heads: int
    head_dim: int
    mlp: int
    vocab: int                  # rows of embed_tokens AFTER resize_token_embeddings (modeling_OmniAVSR.py:214)
    rms_eps: float
    rope_theta: float
    rope_scaling: Optional[dict] = None   # llama3: factor / low_freq_factor / high_freq_factor / original_max_position_embeddings
    qkv_bias: bool = False
    tie_embeddings: bool = False


_L3 = dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)
_L31 = dict(factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)

# vocab = base tokenizer length + added tokens (Llama: <pad> + 4 AV tags, lightning_OmniAVSR.py:83-85; Qwen: 4 AV tags :87)
LLM_SPECS: Dict[str, LLMSpec] = {
    "meta-llama/Llama-3.2-1B": LLMSpec("llama", 2048, 16, 32, 8, 64, 8192, 128261, 1e-5, 500000.0, _L3, False, True),
    "meta-llama/Llama-3.2-3B": LLMSpec("llama", 3072, 28, 24, 8, 128, 8192, 128261, 1e-5, 500000.0, _L3, False, True),
    "meta-llama/Meta-Llama-3.1-8B": LLMSpec("llama", 4096, 32, 32, 8, 128, 14336, 128261, 1e-5, 500000.0, _L31, False, False),
    "meta-llama/Meta-Llama-3-8B": LLMSpec("llama", 4096, 32, 32, 8, 128, 14336, 128261, 1e-5, 500000.0, None, False, False),
    "Qwen/Qwen2.5-0.5B": LLMSpec("qwen2", 896, 24, 14, 2, 64, 4864, 151669, 1e-6, 1000000.0, None, True, True),
    "Qwen/Qwen2.5-1.5B": LLMSpec("qwen2", 1536, 28, 12, 2, 128, 8960, 151669, 1e-6, 1000000.0, None, True, True),
    "Qwen/Qwen2.5-3B": LLMSpec("qwen2", 2048, 36, 16, 2, 128, 11008, 151669, 1e-6, 1000000.0, None, True, True),
    "Qwen/Qwen2.5-7B": LLMSpec("qwen2", 3584, 28, 28, 4, 128, 18944, 151669, 1e-6, 1000000.0, None, True, False),
    "Qwen/Qwen2.5-14B": LLMSpec("qwen2", 5120, 48, 40, 8, 128, 13824, 151669, 1e-6, 1000000.0, None, True, False),
    "Qwen/Qwen2.5-32B": LLMSpec("qwen2", 5120, 64, 40, 8, 128, 27648, 151669, 1e-6, 1000000.0, None, True, False),
}

# token ids (tokenizer facts; lightning_OmniAVSR.py:83-88, modeling_OmniAVSR.py:312-322)
LLAMA_IDS = dict(bos=128000, eos=128001, pad=128256, audio_sos=128257, audio_eos=128258, video_sos=128259, video_eos=128260)
QWEN_IDS = dict(bos=-1, eos=151643, pad=151643, audio_sos=151665, audio_eos=151666, video_sos=151667, video_eos=151668)


@dataclass
class EncoderSpec:
    dim: int
    layers: int
    heads: int
    ffn: int


WHISPER_SPECS: Dict[str, EncoderSpec] = {
    "openai/whisper-tiny": EncoderSpec(384, 4, 6, 1536), "openai/whisper-tiny.en": EncoderSpec(384, 4, 6, 1536),
    "openai/whisper-base": EncoderSpec(512, 6, 8, 2048), "openai/whisper-base.en": EncoderSpec(512, 6, 8, 2048),
    "openai/whisper-small": EncoderSpec(768, 12, 12, 3072), "openai/whisper-small.en": EncoderSpec(768, 12, 12, 3072),
    "openai/whisper-medium": EncoderSpec(1024, 24, 16, 4096), "openai/whisper-medium.en": EncoderSpec(1024, 24, 16, 4096),
}


@dataclass
class AVHubertSpec(EncoderSpec):
    conv_pos: int = 128
    conv_pos_groups: int = 16
    use_lora: bool = False      # --use-lora-avhubert: inner round(dim/16), scaling 2 (modeling_OmniAVSR.py:131-137)


AVHUBERT_LARGE = dict(dim=1024, layers=24, heads=16, ffn=4096)     # av_hubert/avhubert/conf/pretrain/large_vox_iter5.yaml:96-101
AVHUBERT_BASE = dict(dim=768, layers=12, heads=12, ffn=3072)


@dataclass
class OmniSpec:
    """Everything `AVSR_LLMs.__init__` (modeling_OmniAVSR.py:28-34) decides, in one plain object."""
    llm: LLMSpec
    ids: Dict[str, int]
    whisper: Optional[EncoderSpec]
    avhubert: Optional[AVHubertSpec]
    intermediate_size: int = 2048
    compression_mode: str = "avg-pooling"
    downsample_ratio_audio: object = 4          # int, or list when is_matryoshka
    downsample_ratio_video: object = 4
    is_matryoshka: bool = False
    is_single_matry_projector: bool = False
    remove_layernorm_from_projector: bool = False
    projector_layernorm_plain: bool = False     # modeling_LlamaAVSR.py variant: LayerNorm present iff not removed (no Omni quirks)
    lora_rank: int = 0                          # reduction factor RANK (0: no LoRA)
    lora_alpha: int = 1
    is_task_specific: bool = False
    shared_lora: bool = False
    prompt_len: Dict[str, int] = field(default_factory=lambda: {"audio": 8, "video": 8, "audiovisual": 8})
    max_dec_tokens: int = 32
    num_beams: int = 1

    # -- derived quantities, quirks of the reference kept on purpose ---------------------------------
    def rates(self, which: str) -> List[int]:
        r = self.downsample_ratio_audio if which == "audio" else self.downsample_ratio_video
        return list(r) if isinstance(r, (list, tuple)) else [int(r)]

    def proj_per_rate(self) -> bool:
        return bool(self.is_matryoshka and not self.is_single_matry_projector)

    def proj_has_layernorm(self, which: str) -> bool:
        """Whether Sequential index 3 (LayerNorm) really exists (modeling_OmniAVSR.py:74-111,152-196).
        Quirks: per-rate matryoshka avg-pool projectors pass the LayerNorm as nn.Linear's `bias` argument
        (:104,:162,:188) so it is never applied; the audio *stack* matryoshka branch has the polarity
        inverted (:74-77)."""
        rm = self.remove_layernorm_from_projector
        if self.projector_layernorm_plain:
            return not rm              # modeling_LlamaAVSR.py:60-108,150-196: Sequential(..., LayerNorm) unless removed
        if self.compression_mode == "stack":
            if self.is_matryoshka:
                return bool(rm) if which == "audio" else False
            if which == "audio":
                return not rm          # :82-85 (non-matryoshka audio stack is also inverted vs video)
            return not rm
        if self.is_matryoshka:
            if self.is_single_matry_projector:
                return not rm
            return False
        return not rm

    def lora_inner(self) -> int:
        return int(round(self.llm.hidden / self.lora_rank)) if self.lora_rank else 0     # Llama_LoRA.py:125

    def lora_scale(self) -> float:
        return self.lora_alpha / self.lora_rank if self.lora_rank else 0.0              # Llama_LoRA.py:120

    def kv_dim(self) -> int:
        return self.llm.kv_heads * self.llm.head_dim

    def oracle_cfg(self) -> dict:
        """Plain-dict view consumed by oracle/torch_oracle.py (the oracle never imports this package)."""
        l = self.llm
        return dict(
            family=l.family,
            llm=dict(hidden=l.hidden, layers=l.layers, heads=l.heads, kv_heads=l.kv_heads, head_dim=l.head_dim, mlp=l.mlp,
                     rms_eps=l.rms_eps, rope_theta=l.rope_theta, rope_scaling=l.rope_scaling, qkv_bias=l.qkv_bias,
                     lora_rank=self.lora_rank, lora_alpha=self.lora_alpha, task_specific=self.is_task_specific,
                     shared=self.shared_lora),
            whisper=None if self.whisper is None else dict(layers=self.whisper.layers, heads=self.whisper.heads),
            avhubert=None if self.avhubert is None else dict(embed=self.avhubert.dim, layers=self.avhubert.layers,
                                                             heads=self.avhubert.heads, conv_pos=self.avhubert.conv_pos,
                                                             conv_pos_groups=self.avhubert.conv_pos_groups,
                                                             use_lora=self.avhubert.use_lora),
            compression_mode=self.compression_mode, is_matryoshka=self.is_matryoshka,
            is_single_matry_projector=self.is_single_matry_projector,
            rates_audio=self.rates("audio"), rates_video=self.rates("video"), ids=dict(self.ids),
            max_new=self.max_dec_tokens, num_beams=self.num_beams)


def tiny_spec(family: str = "llama", *, head_dim: int = 64, task_specific: bool = False, shared: bool = False,
              use_lora_avhubert: bool = True, matryoshka: bool = False, compression_mode: str = "avg-pooling",
              vocab: int = 512, lora_rank: int = 4, single_projector: bool = False, remove_layernorm: bool = False) -> OmniSpec:
    """Small but structurally faithful configuration for parity tests (GQA 4:1, LoRA on, 2-layer encoders)."""
    if family == "llama":
        llm = LLMSpec("llama", 4 * head_dim, 2, 4, 1, head_dim, 512, vocab, 1e-5, 500000.0, _L3, False, True)
        ids = dict(bos=vocab - 8, eos=vocab - 7, pad=vocab - 5, audio_sos=vocab - 4, audio_eos=vocab - 3, video_sos=vocab - 2, video_eos=vocab - 1)
    else:
        llm = LLMSpec("qwen2", 4 * head_dim, 2, 4, 2, head_dim, 384, vocab, 1e-6, 1000000.0, None, True, False)
        ids = dict(bos=-1, eos=vocab - 7, pad=vocab - 7, audio_sos=vocab - 4, audio_eos=vocab - 3, video_sos=vocab - 2, video_eos=vocab - 1)
    return OmniSpec(
        llm=llm, ids=ids, whisper=EncoderSpec(128, 2, 2, 256),
        avhubert=AVHubertSpec(128, 2, 2, 256, conv_pos=16, conv_pos_groups=4, use_lora=use_lora_avhubert),
        intermediate_size=192, compression_mode=compression_mode,
        downsample_ratio_audio=[2, 4] if matryoshka else 4, downsample_ratio_video=[1, 2] if matryoshka else 2,
        is_matryoshka=matryoshka, is_single_matry_projector=single_projector, remove_layernorm_from_projector=remove_layernorm,
        lora_rank=lora_rank, lora_alpha=2, is_task_specific=task_specific, shared_lora=shared,
        prompt_len={"audio": 5, "video": 6, "audiovisual": 7}, max_dec_tokens=8, num_beams=1)


def spec_from_names(llm_model: str, audio_encoder_name: Optional[str], avhubert: Optional[str], **kw) -> OmniSpec:
    llm = LLM_SPECS[llm_model]
    ids = LLAMA_IDS if llm.family == "llama" else QWEN_IDS
    wh = WHISPER_SPECS[audio_encoder_name] if audio_encoder_name else None
    av = None
    if avhubert:
        base = AVHUBERT_LARGE if "large" in avhubert else AVHUBERT_BASE     # modeling_OmniAVSR.py:125-126
        av = AVHubertSpec(**base, use_lora=kw.pop("use_lora_avhubert", False))
    return OmniSpec(llm=llm, ids=dict(ids), whisper=wh, avhubert=av, **kw)
