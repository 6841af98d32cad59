"""Checkpoint ingestion: the real artefacts a reference user has on disk -> the state-dict layout
`AVSR_LLMs.load_state_dict` takes.

  * `load_omni_checkpoint`  : `model_avg_N.pth` written by utils/avg_checkpoints.py:14-44 (plain state dict of the WHOLE
                              AVSR_LLMs incl. frozen encoders + LLM; a stray `model.` prefix is stripped like :18 does) —
                              this single file is all inference needs (lightning_OmniAVSR.py:148-150 loads it strictly).
  * `load_fairseq_avhubert` : the AV-HuBERT `.pt` (fairseq pickle {"cfg","model","task_state",...}, checkpoint_utils.py:303-329)
                              WITHOUT fairseq / omegaconf installed: unknown pickled classes are replaced by inert stand-ins.
  * `load_hf_safetensors`   : an HF model directory (single or sharded *.safetensors) for the LLM / Whisper encoder.
  * `infer_spec`            : derives every architectural choice of `AVSR_LLMs.__init__` from tensor shapes / key names, so
                              a checkpoint + the LLM name is enough to build the engine.
"""
from __future__ import annotations

import json
import os
import pickle
import re
from typing import Dict, Optional

import torch

from .config import LLM_SPECS, LLAMA_IDS, QWEN_IDS, AVHubertSpec, EncoderSpec, OmniSpec


class _Inert:
    """Stand-in for pickled objects of packages that are not installed (omegaconf nodes, fairseq dataclasses ...)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    def __call__(self, *a, **k):
        return _Inert()


# Globals a checkpoint may legitimately reference.  Everything else (omegaconf nodes, fairseq dataclasses, and anything a hostile
# file names: os.system, builtins.eval ...) resolves to the inert stand-in, so loading a .pt / .pth never runs foreign code.
_SAFE_BUILTINS = {"set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice", "range", "object"}
_SAFE_GLOBALS = {
    "collections": {"OrderedDict", "defaultdict"},
    "argparse": {"Namespace"},
    "numpy": {"ndarray", "dtype"},
    "numpy.core.multiarray": {"_reconstruct", "scalar"},
    "numpy._core.multiarray": {"_reconstruct", "scalar"},
    "torch": {"Size", "device", "dtype", "Tensor"},
    "torch.nn.parameter": {"Parameter"},
    "torch._tensor": {"_rebuild_from_type_v2"},
    "torch.storage": {"UntypedStorage", "TypedStorage"},
}


def _safe_global(module: str, name: str) -> bool:
    if module == "builtins":
        return name in _SAFE_BUILTINS
    if module == "torch._utils":
        return name.startswith("_rebuild_")
    if module == "torch" and (name.endswith("Storage") or name in ("float32", "float16", "bfloat16", "float64", "int64", "int32", "int16", "int8", "uint8", "bool")):
        return True
    return name in _SAFE_GLOBALS.get(module, ())


class _LenientUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if not _safe_global(module, name):
            return _Inert
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return _Inert


class _lenient_pickle:
    Unpickler = _LenientUnpickler
    load = staticmethod(lambda f, **kw: _LenientUnpickler(f, **kw).load())
    __name__ = "pickle"


def _strip_prefix(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    if sd and all(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items()}
    return sd


def load_omni_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_lenient_pickle)
    if isinstance(ckpt, dict) and "state_dict" in ckpt and isinstance(ckpt["state_dict"], dict):   # raw Lightning checkpoint
        ckpt = ckpt["state_dict"]
    sd = {k: v for k, v in ckpt.items() if isinstance(v, torch.Tensor)}
    return _strip_prefix(sd, "model.")


def load_fairseq_avhubert(path: str, prefix: str = "video_encoder.") -> Dict[str, torch.Tensor]:
    state = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_lenient_pickle)
    model = state["model"] if isinstance(state, dict) and "model" in state else state
    return {prefix + k: v for k, v in model.items() if isinstance(v, torch.Tensor)}


def load_hf_safetensors(model_dir: str, prefix: str = "") -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    idx = os.path.join(model_dir, "model.safetensors.index.json")
    files = sorted(set(json.load(open(idx))["weight_map"].values())) if os.path.exists(idx) else \
        sorted(f for f in os.listdir(model_dir) if f.endswith(".safetensors"))
    sd: Dict[str, torch.Tensor] = {}
    for f in files:
        for k, v in load_file(os.path.join(model_dir, f)).items():
            sd[prefix + k] = v
    return sd


def resize_token_embeddings(sd: Dict[str, torch.Tensor], new_rows: int, added=None, prefix: str = "llm.") -> Dict[str, torch.Tensor]:
    """`self.llm.resize_token_embeddings(len(self.tokenizer))` (modeling_OmniAVSR.py:214) on a plain state dict: embed_tokens and lm_head
    grow to `new_rows` (+5 rows for Llama: <pad> and the four audio / video tags, +4 for Qwen; lightning_OmniAVSR.py:83-88).
    The added rows are `added` when given -- the trained values every `model_avg_N.pth` already carries -- else the mean of the existing
    rows (HF draws them at random; a mean keeps an untrained model's logits finite and the result deterministic).  `added`: one tensor
    [n_new, hidden] for tied embeddings; for UNTIED checkpoints (Llama-3.1-8B, Qwen2.5-7B: lm_head is a tensor of its own and its trained
    rows for <pad> / the tags differ from the embedding's) a pair / dict {"embed": ..., "lm_head": ...} -- a single tensor is refused
    there, because copying embedding rows into lm_head would give the added tokens wrong logits (beam search could then emit them)."""
    out = dict(sd)
    ek, hk = prefix + "model.embed_tokens.weight", prefix + "lm_head.weight"
    tied = hk not in sd or sd[hk].data_ptr() == sd[ek].data_ptr()
    if isinstance(added, dict):
        per_key = {ek: added.get("embed"), hk: added.get("lm_head")}
    elif isinstance(added, (tuple, list)):
        per_key = {ek: added[0], hk: added[1]}
    else:
        if added is not None and not tied and new_rows > sd[hk].shape[0]:
            raise ValueError("untied embeddings: `added` must carry separate rows for embed_tokens and lm_head ({'embed': ..., 'lm_head': ...})")
        per_key = {ek: added, hk: added}
    for k in ((ek,) if tied else (ek, hk)):
        w = sd[k]
        n_new = new_rows - w.shape[0]
        if n_new < 0:
            raise ValueError(f"{k} has {w.shape[0]} rows, more than the tokenizer's {new_rows}")
        if n_new:
            rows = per_key[k]
            extra = rows.to(w.dtype) if rows is not None else w.float().mean(0, keepdim=True).expand(n_new, -1).to(w.dtype)
            if extra.shape != (n_new, w.shape[1]):
                raise ValueError(f"added rows have shape {tuple(extra.shape)}, expected {(n_new, w.shape[1])}")
            out[k] = torch.cat([w, extra], 0)
    if tied:
        out[hk] = out[ek]
    return out


def assemble_state_dict(avhubert_pt: Optional[str] = None, whisper_dir: Optional[str] = None, llm_dir: Optional[str] = None,
                        adapter_pth: Optional[str] = None, vocab_rows: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Builds the `AVSR_LLMs.state_dict()` layout from the separate artefacts the reference's constructor pulls in
    (modeling_OmniAVSR.py:59-60,118-147,200-214): the AV-HuBERT fairseq `.pt`, the Whisper / LLM HF directories, and a checkpoint with
    whatever training produced (projectors, LoRA factors, prompt buffers, the added embedding rows under `llm.added_rows`, or
    `llm.added_rows.embed` + `llm.added_rows.lm_head` for models with an untied lm_head)."""
    sd: Dict[str, torch.Tensor] = {}
    if avhubert_pt:
        sd.update(load_fairseq_avhubert(avhubert_pt))
    if whisper_dir:
        w = load_hf_safetensors(whisper_dir)
        sd.update({"audio_encoder." + k[len("model.encoder."):] if k.startswith("model.encoder.") else "audio_encoder." + k[len("encoder."):]: v
                   for k, v in w.items() if k.startswith(("model.encoder.", "encoder."))})
    if llm_dir:
        sd.update(load_hf_safetensors(llm_dir, prefix="llm."))
        if "llm.lm_head.weight" not in sd:                                      # tie_word_embeddings checkpoints ship one tensor
            sd["llm.lm_head.weight"] = sd["llm.model.embed_tokens.weight"]
    added = None
    if adapter_pth:
        ad = load_omni_checkpoint(adapter_pth)
        added = ad.pop("llm.added_rows", None)
        if "llm.added_rows.embed" in ad or "llm.added_rows.lm_head" in ad:      # untied models: the two tensors' rows are stored separately
            added = {"embed": ad.pop("llm.added_rows.embed", None), "lm_head": ad.pop("llm.added_rows.lm_head", None)}
        sd.update(ad)
    if vocab_rows is not None and llm_dir:
        sd = resize_token_embeddings(sd, vocab_rows, added)
    return sd


def infer_spec(sd: Dict[str, torch.Tensor], llm_model: str, compression_mode: str = "avg-pooling", lora_alpha: int = 1,
               downsample_ratio_audio=None, downsample_ratio_video=None, max_dec_tokens: int = 32, num_beams: int = 15,
               ids: Optional[Dict[str, int]] = None) -> OmniSpec:
    """Reads the architecture off a reference checkpoint.  What shapes cannot tell (compression rates of parameter-free
    avg-pool layers, ALPHA, token ids of a resized tokenizer) comes from the arguments, like the reference's argparse flags."""
    llm = LLM_SPECS[llm_model]
    emb_rows = sd["llm.model.embed_tokens.weight"].shape[0]
    if emb_rows != llm.vocab:
        llm = type(llm)(**{**llm.__dict__, "vocab": emb_rows})
    if ids is None:
        ids = dict(LLAMA_IDS if llm.family == "llama" else QWEN_IDS)
    whisper = avh = None
    if "audio_encoder.conv1.weight" in sd:
        d = sd["audio_encoder.conv1.weight"].shape[0]
        n = 1 + max(int(m.group(1)) for k in sd if (m := re.match(r"audio_encoder\.layers\.(\d+)\.", k)))
        whisper = EncoderSpec(d, n, d // 64, sd["audio_encoder.layers.0.fc1.weight"].shape[0])
    if "video_encoder.post_extract_proj.weight" in sd:
        d = sd["video_encoder.post_extract_proj.weight"].shape[0]
        n = 1 + max(int(m.group(1)) for k in sd if (m := re.match(r"video_encoder\.encoder\.layers\.(\d+)\.", k)))
        v = sd["video_encoder.encoder.pos_conv.0.weight_v"]
        avh = AVHubertSpec(d, n, d // 64, sd["video_encoder.encoder.layers.0.fc1.weight"].shape[0], conv_pos=v.shape[2],
                           conv_pos_groups=d // v.shape[1], use_lora="video_encoder.encoder.layers.0.self_attn.lora_down_Q.weight" in sd)
    per_rate = any(re.match(r"(audio|video)_proj\.\d+\.0\.weight", k) for k in sd)
    pkey = next(k for k in sd if re.match(r"(audio|video)_proj\.(\d+\.)?0\.weight", k))
    inter = sd[pkey].shape[0]
    lp = "llm.model.layers.0.self_attn."
    task_specific = lp + "lora_down_Q.audio.weight" in sd
    shared = lp + "lora_down_Q_shared.weight" in sd
    dn = sd.get(lp + ("lora_down_Q.audio.weight" if task_specific else "lora_down_Q.weight"))
    rank = 0
    if dn is not None:
        inner = dn.shape[0]
        rank = next((r for r in range(1, 4097) if round(llm.hidden / r) == inner), 0)      # Llama_LoRA.py:125
    n_a = 1 + max([int(m.group(1)) for k in sd if (m := re.match(r"audio_proj\.(\d+)\.0\.weight", k))], default=0)
    n_v = 1 + max([int(m.group(1)) for k in sd if (m := re.match(r"video_proj\.(\d+)\.0\.weight", k))], default=0)
    ra = downsample_ratio_audio if downsample_ratio_audio is not None else ([4, 16][:n_a] if per_rate else 4)
    rv = downsample_ratio_video if downsample_ratio_video is not None else ([2, 5][:n_v] if per_rate else 2)
    is_matry = isinstance(ra, (list, tuple)) or isinstance(rv, (list, tuple))
    a_ln = any(re.match(r"audio_proj\.(\d+\.)?3\.weight", k) for k in sd)
    v_ln = any(re.match(r"video_proj\.(\d+\.)?3\.weight", k) for k in sd)
    spec = OmniSpec(llm=llm, ids=ids, whisper=whisper, avhubert=avh, intermediate_size=inter, compression_mode=compression_mode,
                    downsample_ratio_audio=ra, downsample_ratio_video=rv, is_matryoshka=is_matry,
                    is_single_matry_projector=bool(is_matry and not per_rate), remove_layernorm_from_projector=not (a_ln or v_ln),
                    lora_rank=rank, lora_alpha=lora_alpha, is_task_specific=task_specific, shared_lora=shared,
                    prompt_len={t: int(sd[f"prompt_{t}"].shape[1]) for t in ("audio", "video", "audiovisual") if f"prompt_{t}" in sd},
                    max_dec_tokens=max_dec_tokens, num_beams=num_beams)
    return spec
