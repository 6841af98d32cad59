#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/glue_*.npz by EXECUTING the reference's own glue code (build container only).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPYCACHEPREFIX=/tmp/pyc python oracle/gen_golden_glue.py

What runs from /root/reference, unmodified:
  * `AVSR_LLMs.prepare_inputs` (inference branch), `encode_audio`, `encode_video`  (Omni_AVSR/modeling_OmniAVSR.py:397-458,
    528-558, 461-483) on an instance assembled attribute by attribute (the constructor needs the hub: `from_pretrained`, :59-60,
    200-207), with the real vendored `AVHubertModel` as `video_encoder`, the installed transformers' `WhisperEncoder` /
    `WhisperFeatureExtractor` as `audio_encoder` / `audio_frontend`, and the projectors built by the very expressions of the
    constructor (:86-111,171-196) -- including the variants where the LayerNorm is swallowed as `nn.Linear`'s bias argument;
  * `LlamaDecoderLayer_lora.forward` -> `LlamaSdpaAttention_lora.forward` (Omni_AVSR/Llama_LoRA.py:587-655,214-316) and the
    Qwen twins (Omni_AVSR/Qwen_LoRA.py:388-450,529-619): a prefill over S positions and one cached decode step.
In-memory accommodations (no reference source is copied or edited): the two class names transformers 5.x dropped
(`LlamaSdpaAttention`, `Qwen2SdpaAttention`) are aliased to their successors before the import; `Tensor.cuda()` is a no-op while
the reference's `encode_audio` runs (it hard-codes `.cuda()`, :534); the Whisper encoder upcasts the bf16 features the
reference hands it (`.to(torch.bfloat16)`, :534) because its weights are fp32 here.
Weights are NOT stored: they come from `omni_avsr_amd.synth.synth_state_dict(tiny_spec(...), seed)`; the fixtures hold seeds,
the call arguments and the expected outputs.  The script asserts on the spot that oracle/torch_oracle.py reproduces every fixture.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402

from omni_avsr_amd.config import tiny_spec  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import ref_import as R  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 1234


def import_reference_glue():
    R._install_stubs()
    for p in (R.REF_ROOT, os.path.join(R.REF_ROOT, "av_hubert", "fairseq")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import transformers.models.llama.modeling_llama as ML
    import transformers.models.qwen2.modeling_qwen2 as MQ
    import transformers.utils.doc as D
    if not hasattr(ML, "LlamaSdpaAttention"):
        ML.LlamaSdpaAttention = ML.LlamaAttention
    if not hasattr(MQ, "Qwen2SdpaAttention"):
        MQ.Qwen2SdpaAttention = MQ.Qwen2Attention
    if not hasattr(D, "add_start_docstrings_to_model_forward"):
        D.add_start_docstrings_to_model_forward = lambda *a, **k: (lambda f: f)
    argv = sys.argv
    sys.argv = ["x"]
    try:
        import Omni_AVSR.modeling_OmniAVSR as M
        import Omni_AVSR.Llama_LoRA as LL
        import Omni_AVSR.Qwen_LoRA as QL
    finally:
        sys.argv = argv
    return M, LL, QL


class _UpcastEncoder(nn.Module):
    """The reference feeds bf16 features (:534); the HF encoder below holds fp32 weights."""

    def __init__(self, enc):
        super().__init__()
        self.enc = enc

    def forward(self, x):
        return self.enc(x.float())


def build_reference_model(M, spec, sd, modality, llm_name):
    """An `AVSR_LLMs` instance with exactly the attributes the inference branch reads, filled like the constructor fills them."""
    from transformers import WhisperConfig, WhisperFeatureExtractor
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    m = M.AVSR_LLMs.__new__(M.AVSR_LLMs)
    nn.Module.__init__(m)
    d, inter = spec.llm.hidden, spec.intermediate_size
    m.modality, m.llm_model, m.compression_mode = modality, llm_name, spec.compression_mode
    m.is_matryoshka, m.is_single_matry_projector = spec.is_matryoshka, spec.is_single_matry_projector
    m.remove_layernorm_from_projector = spec.remove_layernorm_from_projector
    m.downsample_ratio_audio, m.downsample_ratio_video = spec.downsample_ratio_audio, spec.downsample_ratio_video
    rm = spec.remove_layernorm_from_projector
    if modality in ("audio", "audiovisual"):
        w = spec.whisper
        cfg = WhisperConfig(d_model=w.dim, encoder_layers=w.layers, encoder_attention_heads=w.heads, encoder_ffn_dim=w.ffn, num_mel_bins=80,
                            max_source_positions=1500, decoder_layers=1, decoder_attention_heads=w.heads, decoder_ffn_dim=w.ffn, vocab_size=64)
        enc = WhisperEncoder(cfg)
        enc.load_state_dict({k[len("audio_encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.")}, strict=True)
        m.audio_encoder = _UpcastEncoder(enc.eval())
        m.audio_frontend = WhisperFeatureExtractor()
        audio_dim = w.dim
        ra = spec.downsample_ratio_audio
        # the constructor's expressions, modeling_OmniAVSR.py:65-111
        if spec.compression_mode == "stack":
            if spec.is_matryoshka:
                m.matry_map_audio = {el: i for i, el in enumerate(ra)}
                if rm:
                    m.audio_proj = nn.ModuleList([nn.Sequential(nn.Linear(audio_dim * r, inter), nn.ReLU(), nn.Linear(inter, d), nn.LayerNorm(d)) for r in ra])
                else:
                    m.audio_proj = nn.ModuleList([nn.Sequential(nn.Linear(audio_dim * r, inter), nn.ReLU(), nn.Linear(inter, d)) for r in ra])
            elif rm:
                m.audio_proj = nn.Sequential(nn.Linear(audio_dim * ra, inter), nn.ReLU(), nn.Linear(inter, d))
            else:
                m.audio_proj = nn.Sequential(nn.Linear(audio_dim * ra, inter), nn.ReLU(), nn.Linear(inter, d), nn.LayerNorm(d))
        else:
            if spec.is_matryoshka:
                m.matry_map_audio = {el: i for i, el in enumerate(ra)}
                m.avg_pool_audio = nn.ModuleList([nn.AvgPool1d(r) for r in ra])
                if rm:
                    m.audio_proj = (nn.Sequential(nn.Linear(audio_dim, inter), nn.ReLU(), nn.Linear(inter, d)) if spec.is_single_matry_projector else
                                    nn.ModuleList([nn.Sequential(nn.Linear(audio_dim, inter), nn.ReLU(), nn.Linear(inter, d)) for _ in ra]))
                elif spec.is_single_matry_projector:
                    m.audio_proj = nn.Sequential(nn.Linear(audio_dim, inter), nn.ReLU(), nn.Linear(inter, d), nn.LayerNorm(d))
                else:
                    m.audio_proj = nn.ModuleList([nn.Sequential(nn.Linear(audio_dim, inter), nn.ReLU(), nn.Linear(inter, d, nn.LayerNorm(d))) for _ in ra])
            else:
                m.avg_pool_audio = nn.AvgPool1d(ra)
                m.audio_proj = (nn.Sequential(nn.Linear(audio_dim, inter), nn.ReLU(), nn.Linear(inter, d)) if rm else
                                nn.Sequential(nn.Linear(audio_dim, inter), nn.ReLU(), nn.Linear(inter, d), nn.LayerNorm(d)))
        m.audio_proj.load_state_dict({k[len("audio_proj."):]: v for k, v in sd.items() if k.startswith("audio_proj.")}, strict=True)
    if modality in ("video", "audiovisual"):
        a = spec.avhubert
        cfg = R.avhubert_cfg(embed=a.dim, layers=a.layers, heads=a.heads, ffn=a.ffn, conv_pos=a.conv_pos, conv_pos_groups=a.conv_pos_groups)
        ve = R.build_reference_avhubert(cfg, a.use_lora)
        ref_sd = {k[len("video_encoder."):]: v for k, v in sd.items() if k.startswith("video_encoder.")}
        own = ve.state_dict()
        for k in own:
            ref_sd.setdefault(k, own[k])
        ve.load_state_dict(ref_sd, strict=True)
        m.video_encoder = ve
        video_dim = a.dim
        rv = spec.downsample_ratio_video
        if spec.compression_mode == "stack":       # :149-169
            if spec.is_matryoshka:
                m.matry_map_video = {el: i for i, el in enumerate(rv)}
                if rm:
                    m.video_proj = nn.ModuleList([nn.Sequential(nn.Linear(video_dim * r, inter), nn.ReLU(), nn.Linear(inter, d)) for r in rv])
                else:
                    m.video_proj = nn.ModuleList([nn.Sequential(nn.Linear(video_dim * r, inter), nn.ReLU(), nn.Linear(inter, d, nn.LayerNorm(d))) for r in rv])
            elif rm:
                m.video_proj = nn.Sequential(nn.Linear(video_dim * rv, inter), nn.ReLU(), nn.Linear(inter, d))
            else:
                m.video_proj = nn.Sequential(nn.Linear(video_dim * rv, inter), nn.ReLU(), nn.Linear(inter, d), nn.LayerNorm(d))
        else:                                       # :171-196
            if spec.is_matryoshka:
                m.matry_map_video = {el: i for i, el in enumerate(rv)}
                m.avg_pool_video = nn.ModuleList([nn.AvgPool1d(r) for r in rv])
                if rm:
                    m.video_proj = (nn.Sequential(nn.Linear(video_dim, inter), nn.ReLU(), nn.Linear(inter, d)) if spec.is_single_matry_projector else
                                    nn.ModuleList([nn.Sequential(nn.Linear(video_dim, inter), nn.ReLU(), nn.Linear(inter, d)) for _ in rv]))
                elif spec.is_single_matry_projector:
                    m.video_proj = nn.Sequential(nn.Linear(video_dim, inter), nn.ReLU(), nn.Linear(inter, d), nn.LayerNorm(d))
                else:
                    m.video_proj = nn.ModuleList([nn.Sequential(nn.Linear(video_dim, inter), nn.ReLU(), nn.Linear(inter, d, nn.LayerNorm(d))) for _ in rv])
            else:
                m.avg_pool_video = nn.AvgPool1d(rv)
                m.video_proj = (nn.Sequential(nn.Linear(video_dim, inter), nn.ReLU(), nn.Linear(inter, d)) if rm else
                                nn.Sequential(nn.Linear(video_dim, inter), nn.ReLU(), nn.Linear(inter, d), nn.LayerNorm(d)))
        m.video_proj.load_state_dict({k[len("video_proj."):]: v for k, v in sd.items() if k.startswith("video_proj.")}, strict=True)
    emb = nn.Embedding(spec.llm.vocab, d)
    emb.weight.data.copy_(sd["llm.model.embed_tokens.weight"])
    m.llm = types.SimpleNamespace(model=types.SimpleNamespace(embed_tokens=emb))
    ids = spec.ids
    m.tokenizer = types.SimpleNamespace(vocab={"<audio>": ids["audio_sos"], "</audio>": ids["audio_eos"], "<video>": ids["video_sos"], "</video>": ids["video_eos"]})
    for t in ("audio", "video", "audiovisual"):
        m.register_buffer(f"prompt_{t}", sd[f"prompt_{t}"].clone())
    return m.eval()


CASES = {
    # name: (tiny_spec kwargs, llm_model name, modality, rate_audio, rate_video)
    "llama_matry_perrate": (dict(family="llama", task_specific=True, shared=True, matryoshka=True), "meta-llama/Llama-3.2-1B", "audiovisual", 4, 2),
    "qwen_single_projector": (dict(family="qwen2", matryoshka=True, single_projector=True), "Qwen/Qwen2.5-7B", "audiovisual", 2, 1),
    # without --is-matryoshka the inference branch only works with --is-single-matry-projector: :429,447 index `matry_map_*`, which the
    # constructor creates in the matryoshka branches only (AttributeError otherwise)
    "llama_plain_audio": (dict(family="llama", single_projector=True), "meta-llama/Llama-3.2-1B", "audio", 4, 2),
    "llama_plain_video_noln": (dict(family="llama", remove_layernorm=True, single_projector=True), "meta-llama/Llama-3.2-1B", "video", 4, 2),
    "llama_stack_matry": (dict(family="llama", matryoshka=True, compression_mode="stack", remove_layernorm=True), "meta-llama/Llama-3.2-1B", "audiovisual", 2, 2),
    "qwen_stack_plain": (dict(family="qwen2", compression_mode="stack", single_projector=True), "Qwen/Qwen2.5-7B", "audiovisual", 4, 2),
}
# frames per case: the non-matryoshka video *stack* branch (:517-520) has no remainder handling (torch.stack raises on a ragged last
# group), so that case needs T_v % rate == 0
CASE_TV = {"qwen_stack_plain": 12}


def glue_inputs(spec, seed, Tv=11, Ts=7040):
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(1, Tv, 1, 88, 88, generator=g)
    audio = torch.randn(1, Ts, 1, generator=g)
    tokens = torch.full((1, 1), spec.ids["bos"]) if spec.llm.family == "llama" else torch.zeros(1, 0, dtype=torch.long)
    return dict(video=video, audio=audio, lengths=torch.tensor([Ts]), tokens=tokens)


def gen_prepare_inputs(M):
    out = {}
    for name, (kw, llm_name, modality, ra, rv) in CASES.items():
        spec = tiny_spec(**kw)
        sd = synth_state_dict(spec, seed=SEED)
        m = build_reference_model(M, spec, sd, modality, llm_name)
        inp = glue_inputs(spec, 21, Tv=CASE_TV.get(name, 11))
        cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self                     # modeling_OmniAVSR.py:534 hard-codes .cuda()
        try:
            with torch.no_grad():
                if spec.is_matryoshka:
                    emb = m.prepare_inputs(inp, False, test_ratio_matry_audio=ra, test_ratio_matry_video=rv)
                else:
                    emb = m.prepare_inputs(inp, False)
        finally:
            torch.Tensor.cuda = cuda
        orc = O.OmniOracle(sd, spec.oracle_cfg())
        with torch.no_grad():
            mine = orc.prepare_inputs(inp, modality, ra if spec.is_matryoshka else spec.rates("audio")[0], rv if spec.is_matryoshka else spec.rates("video")[0],
                                      mel_bf16=True)
        err = (mine - emb).abs().max().item()
        print(f"prepare_inputs[{name}]: S={emb.shape[1]} oracle vs executed reference max abs err {err:.2e}")
        assert mine.shape == emb.shape and err < 2e-4, name
        out[name] = emb[0].numpy()
    np.savez_compressed(os.path.join(GOLD, "glue_prepare_inputs.npz"), seed=SEED, input_seed=21, Tv=11, Ts=7040,
                        **{f"Tv_{k}": v for k, v in CASE_TV.items()}, **out)


def _layer_sd(sd, family):
    """one decoder layer of the synthetic checkpoint -> the reference module's own key names"""
    p = "llm.model.layers.0."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


def gen_decoder_layer(LL, QL):
    from transformers import LlamaConfig, Qwen2Config
    from transformers.cache_utils import DynamicCache
    res = {}
    for family in ("llama", "qwen2"):
        spec = tiny_spec(family=family, head_dim=64 if family == "llama" else 128, task_specific=True, shared=True)
        l = spec.llm
        sd = synth_state_dict(spec, seed=SEED, modalities=())
        if family == "llama":
            cfg = LlamaConfig(hidden_size=l.hidden, intermediate_size=l.mlp, num_hidden_layers=1, num_attention_heads=l.heads, num_key_value_heads=l.kv_heads,
                              vocab_size=l.vocab, rms_norm_eps=l.rms_eps, rope_theta=l.rope_theta, max_position_embeddings=16384, attention_bias=False,
                              head_dim=l.head_dim)
            layer = LL.LlamaDecoderLayer_lora(cfg, 0, LL.LoRA_config(RANK=spec.lora_rank, ALPHA=spec.lora_alpha, IS_LLAMA3=True, IS_TASK_SPECIFIC=True, SHARED_LORA=True))
        else:
            cfg = Qwen2Config(hidden_size=l.hidden, intermediate_size=l.mlp, num_hidden_layers=1, num_attention_heads=l.heads, num_key_value_heads=l.kv_heads,
                              vocab_size=l.vocab, rms_norm_eps=l.rms_eps, rope_theta=l.rope_theta, max_position_embeddings=16384)
            # hidden // gqa_factor must equal kv_heads * head_dim = hidden / 2 here; the class only knows the real models' factors, so the
            # value projections it creates are re-shaped below to the checkpoint's (the forward never looks at the factor again)
            layer = QL.Qwen2DecoderLayer_lora(cfg, 0, QL.QwenLoRA_config(RANK=spec.lora_rank, ALPHA=spec.lora_alpha, IS_QWEN25_7B=True, IS_TASK_SPECIFIC=True, SHARED_LORA=True))
            att = layer.self_attn
            inner = round(l.hidden / spec.lora_rank)
            for t in ("audio", "video", "audiovisual"):
                att.lora_up_V[t] = nn.Linear(inner, spec.kv_dim(), bias=False)
            att.lora_up_V_shared = nn.Linear(inner, spec.kv_dim(), bias=False)
        att = layer.self_attn
        for nme, val in (("num_heads", l.heads), ("num_key_value_heads", l.kv_heads), ("hidden_size", l.hidden)):
            if not hasattr(att, nme):
                setattr(att, nme, val)
        layer.load_state_dict(_layer_sd(sd, family), strict=True)
        layer.eval()
        inv = O.rope_inv_freq(l.head_dim, l.rope_theta, l.rope_scaling)

        def cos_sin(pos):
            fr = pos.float()[:, None] * inv[None, :]
            e = torch.cat((fr, fr), -1)
            return e.cos()[None], e.sin()[None]

        if family == "qwen2":     # Qwen_LoRA.py:576-581 asks the layer's own rotary module: (value_states, seq_len=...) -> cos / sin tables indexed by position_ids
            class Rot(nn.Module):
                def forward(self, x, seq_len=None):
                    c, s_ = cos_sin(torch.arange(seq_len))
                    return c[0], s_[0]
            att.rotary_emb = Rot()
            import transformers.models.qwen2.modeling_qwen2 as MQ
            real_apply = MQ.apply_rotary_pos_emb

            def apply5(q, k, cos, sin, position_ids=None, unsqueeze_dim=1):      # the 4.43 signature (cos / sin tables + position_ids)
                cos, sin = cos[position_ids], sin[position_ids]
                return real_apply(q, k, cos, sin, unsqueeze_dim=unsqueeze_dim)
            QL.apply_rotary_pos_emb = apply5
        g = torch.Generator().manual_seed(5)
        S = 7
        x = torch.randn(1, S, l.hidden, generator=g)
        x1 = torch.randn(1, 1, l.hidden, generator=g)
        cache = DynamicCache()
        if not hasattr(cache, "get_usable_length"):
            cache.get_usable_length = lambda new_len, layer_idx=0: cache.get_seq_length(layer_idx)
        with torch.no_grad():
            kw = dict(past_key_value=cache, use_cache=True, modality="audiovisual")
            if family == "llama":
                y = layer(x, position_ids=torch.arange(S)[None], position_embeddings=cos_sin(torch.arange(S)), **kw)[0]
                y1 = layer(x1, position_ids=torch.tensor([[S]]), position_embeddings=cos_sin(torch.tensor([S])), **kw)[0]
            else:
                y = layer(x, position_ids=torch.arange(S)[None], **kw)[0]
                y1 = layer(x1, position_ids=torch.tensor([[S]]), **kw)[0]
        # the oracle on the same layer
        cfg1 = dict(spec.oracle_cfg()["llm"], layers=1)
        sd1 = {k: v for k, v in sd.items() if not k.startswith("llm.model.layers.1.")}
        orc = O.LLMOracle(sd1, cfg1)
        oc = [None]
        with torch.no_grad():
            _, h = orc.forward(x, 0, oc, "audiovisual", return_hidden=True)
            _, h1 = orc.forward(x1, S, oc, "audiovisual", return_hidden=True)
        e0, e1 = (h[0] - y).abs().max().item(), (h1[0] - y1).abs().max().item()
        print(f"decoder layer [{family}]: oracle vs executed reference, prefill {e0:.2e}, cached decode step {e1:.2e}")
        assert e0 < 2e-5 and e1 < 2e-5, family
        res[f"{family}_prefill"], res[f"{family}_step"] = y[0].numpy(), y1[0].numpy()
    np.savez_compressed(os.path.join(GOLD, "glue_decoder_layer.npz"), seed=SEED, input_seed=5, S=7, **res)


if __name__ == "__main__":
    assert R.reference_available(), "needs /root/reference (build container)"
    torch.manual_seed(0)
    M, LL, QL = import_reference_glue()
    gen_prepare_inputs(M)
    gen_decoder_layer(LL, QL)
    print("glue fixtures written to", GOLD)
