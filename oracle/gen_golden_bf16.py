#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/ref_bf16.npz by EXECUTING the reference's own classes in torch.bfloat16 on the CPU of
the build container: the arithmetic of its evaluation, `Trainer(precision='bf16-true')` (eval_OmniAVSR.py:18).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPYCACHEPREFIX=/tmp/pyc python oracle/gen_golden_bf16.py

What `bf16-true` does in pytorch-lightning 2.3.3 (requirements.txt:5), restated here because Lightning is not installed:
  * `HalfPrecision.convert_module`: the whole module `.to(torch.bfloat16)`;
  * `HalfPrecision.convert_input` (called by the evaluation loop on every batch): every floating tensor of the batch -> bf16.  The video
    AND THE WAVEFORM therefore reach `AVSR_LLMs.forward` rounded to bf16 -- which is why `encode_audio` starts with
    `audio.to(torch.float32)` (modeling_OmniAVSR.py:530-531, and the comment above it);
  * `HalfPrecision.forward_context`: `torch.set_default_dtype(torch.bfloat16)` around the step.
What runs from /root/reference, unmodified (import accommodations: oracle/gen_golden_glue.py): `AVSR_LLMs.prepare_inputs` / `encode_audio`
/ `encode_video` (modeling_OmniAVSR.py:397-458,528-558,461-483) over the vendored `AVHubertModel` with the LoRA-patched
`MultiheadAttention.forward_lora` (multihead_attention.py:485-511) and the installed transformers' `WhisperEncoder`, then two
`LlamaDecoderLayer_lora` / `Qwen2DecoderLayer_lora` layers (Llama_LoRA.py:587-655, Qwen_LoRA.py:388-450) + the final RMSNorm + lm_head
with `logits.float()` (Llama_LoRA.py:372-373): a prefill over the prompt and one cached greedy step.
Stored: seeds, shapes and the bf16 outputs (as f32) -- compressed modality features, `inputs_embeds`, last-position logits of the
prefill and of the step.  Weights come from `synth_state_dict(tiny_spec(...), seed)`; nothing of the reference is stored.
tests/test_bf16_reference_gpu.py holds the bf16 engine to them: its distance to the fp32 oracle must not exceed 1.5 x the distance of
the reference's own bf16 run to the fp32 oracle, per stage.
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402

from omni_avsr_amd.config import tiny_spec  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import gen_golden_glue as G  # noqa: E402
from oracle import ref_import as R  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 1234
INPUT_SEED = 33
CASES = {
    # name: (tiny_spec kwargs, llm_model name, rate_audio, rate_video, frames)
    "llama": (dict(family="llama", task_specific=True, shared=True, matryoshka=True), "meta-llama/Llama-3.2-1B", 4, 2, 12),
    "qwen": (dict(family="qwen2", head_dim=128, matryoshka=True, single_projector=True), "Qwen/Qwen2.5-7B", 2, 1, 10),
}


def bf16_inputs(spec, Tv):
    """what `HalfPrecision.convert_input` hands to forward(): floating tensors in bf16"""
    g = torch.Generator().manual_seed(INPUT_SEED)
    video = torch.randn(1, Tv, 1, 88, 88, generator=g).bfloat16()
    audio = torch.randn(1, 640 * Tv, 1, generator=g).bfloat16()
    tokens = torch.full((1, 1), spec.ids["bos"]) if spec.llm.family == "llama" else torch.zeros(1, 0, dtype=torch.long)
    return dict(video=video, audio=audio, lengths=torch.tensor([640 * Tv]), tokens=tokens)


def build_llm_layers(LL, QL, spec, sd):
    """the reference's decoder-layer classes for every layer of the tiny LLM + HF's RMSNorm and the lm_head, bf16"""
    from transformers import LlamaConfig, Qwen2Config
    l = spec.llm
    layers = []
    for i in range(l.layers):
        if l.family == "llama":
            from transformers.models.llama.modeling_llama import LlamaRMSNorm as Norm
            cfg = LlamaConfig(hidden_size=l.hidden, intermediate_size=l.mlp, num_hidden_layers=l.layers, num_attention_heads=l.heads,
                              num_key_value_heads=l.kv_heads, vocab_size=l.vocab, rms_norm_eps=l.rms_eps, rope_theta=l.rope_theta,
                              max_position_embeddings=16384, attention_bias=False, head_dim=l.head_dim)
            layer = LL.LlamaDecoderLayer_lora(cfg, i, LL.LoRA_config(RANK=spec.lora_rank, ALPHA=spec.lora_alpha, IS_LLAMA3=True,
                                                                     IS_TASK_SPECIFIC=spec.is_task_specific, SHARED_LORA=spec.shared_lora))
        else:
            from transformers.models.qwen2.modeling_qwen2 import Qwen2RMSNorm as Norm
            cfg = Qwen2Config(hidden_size=l.hidden, intermediate_size=l.mlp, num_hidden_layers=l.layers, num_attention_heads=l.heads,
                              num_key_value_heads=l.kv_heads, vocab_size=l.vocab, rms_norm_eps=l.rms_eps, rope_theta=l.rope_theta,
                              max_position_embeddings=16384)
            layer = QL.Qwen2DecoderLayer_lora(cfg, i, QL.QwenLoRA_config(RANK=spec.lora_rank, ALPHA=spec.lora_alpha, IS_QWEN25_7B=True,
                                                                         IS_TASK_SPECIFIC=spec.is_task_specific, SHARED_LORA=spec.shared_lora))
            att = layer.self_attn           # the class only knows the real models' GQA factors (gen_golden_glue.gen_decoder_layer)
            inner = round(l.hidden / spec.lora_rank)
            if spec.is_task_specific:
                for t in ("audio", "video", "audiovisual"):
                    att.lora_up_V[t] = nn.Linear(inner, spec.kv_dim(), bias=False)
                if spec.shared_lora:
                    att.lora_up_V_shared = nn.Linear(inner, spec.kv_dim(), bias=False)
            else:
                att.lora_up_V = nn.Linear(inner, spec.kv_dim(), bias=False)
        att = layer.self_attn
        for nme, val in (("num_heads", l.heads), ("num_key_value_heads", l.kv_heads), ("hidden_size", l.hidden)):
            if not hasattr(att, nme):
                setattr(att, nme, val)
        p = f"llm.model.layers.{i}."
        layer.load_state_dict({k[len(p):]: v for k, v in sd.items() if k.startswith(p)}, strict=True)
        layers.append(layer.eval().to(torch.bfloat16))
    norm = Norm(l.hidden, eps=l.rms_eps)
    norm.weight.data.copy_(sd["llm.model.norm.weight"])
    head = nn.Linear(l.hidden, l.vocab, bias=False)
    head.weight.data.copy_(sd["llm.lm_head.weight"] if "llm.lm_head.weight" in sd else sd["llm.model.embed_tokens.weight"])
    return layers, norm.to(torch.bfloat16), head.to(torch.bfloat16)


def run_llm(QL, spec, layers, norm, head, embeds, modality, embed_row):
    """prefill over `embeds` [1, S, d] (bf16) + one cached greedy step: the reference's layer stack the way LlamaModel_lora / Qwen2Model_lora
    drive it (rotary tables in fp32, cast to the activations' dtype as transformers' rotary modules do)"""
    from transformers.cache_utils import DynamicCache
    l = spec.llm
    inv = O.rope_inv_freq(l.head_dim, l.rope_theta, l.rope_scaling)

    def cos_sin(pos, dtype):
        fr = pos.float()[:, None] * inv[None, :]
        e = torch.cat((fr, fr), -1)
        return e.cos()[None].to(dtype), e.sin()[None].to(dtype)

    if l.family == "qwen2":
        class Rot(nn.Module):
            def forward(self, x, seq_len=None):
                c, s_ = cos_sin(torch.arange(seq_len), x.dtype)
                return c[0], s_[0]
        import transformers.models.qwen2.modeling_qwen2 as MQ
        real_apply = MQ.apply_rotary_pos_emb

        def apply5(q, k, cos, sin, position_ids=None, unsqueeze_dim=1):      # the 4.43 signature (tables + position_ids)
            return real_apply(q, k, cos[position_ids], sin[position_ids], unsqueeze_dim=unsqueeze_dim)
        QL.apply_rotary_pos_emb = apply5
        for layer in layers:
            layer.self_attn.rotary_emb = Rot()
    cache = DynamicCache()
    if not hasattr(cache, "get_usable_length"):
        cache.get_usable_length = lambda new_len, layer_idx=0: cache.get_seq_length(layer_idx)
    mod = modality if spec.is_task_specific else None

    def stack(x, pos):
        kw = dict(past_key_value=cache, use_cache=True, modality=mod)
        for layer in layers:
            if l.family == "llama":
                x = layer(x, position_ids=pos[None], position_embeddings=cos_sin(pos, x.dtype), **kw)[0]
            else:
                x = layer(x, position_ids=pos[None], **kw)[0]
        return head(norm(x)).float()                                           # Llama_LoRA.py:372-373

    S = embeds.shape[1]
    lg0 = stack(embeds, torch.arange(S))[0, -1]
    tok = int(lg0.argmax())
    lg1 = stack(embed_row(tok), torch.tensor([S]))[0, -1]                      # one cached greedy step
    return lg0, tok, lg1, S


def main():
    assert R.reference_available(), "needs /root/reference (build container)"
    torch.manual_seed(0)
    M, LL, QL = G.import_reference_glue()
    out = dict(seed=SEED, input_seed=INPUT_SEED)
    for name, (kw, llm_name, ra, rv, Tv) in CASES.items():
        spec = tiny_spec(**kw)
        sd = synth_state_dict(spec, seed=SEED)
        m = G.build_reference_model(M, spec, sd, "audiovisual", llm_name)
        m.audio_encoder = m.audio_encoder.enc                       # bf16 weights take the bf16 features as they come (:534)
        m = m.to(torch.bfloat16)                                    # HalfPrecision.convert_module
        m.llm.model.embed_tokens.to(torch.bfloat16)                 # (the stand-in `llm` namespace is not a registered submodule)
        inp = bf16_inputs(spec, Tv)
        feats = {}
        ev, ea = m.encode_video, m.encode_audio
        m.encode_video = lambda *a, **k: feats.setdefault("video", ev(*a, **k))
        m.encode_audio = lambda *a, **k: feats.setdefault("audio", ea(*a, **k))
        cuda, dflt = torch.Tensor.cuda, torch.get_default_dtype()
        torch.Tensor.cuda = lambda self, *a, **k: self               # modeling_OmniAVSR.py:534 hard-codes .cuda()
        torch.set_default_dtype(torch.bfloat16)                      # HalfPrecision.forward_context
        try:
            with torch.no_grad():
                emb = m.prepare_inputs(inp, False, test_ratio_matry_audio=ra, test_ratio_matry_video=rv)
                assert emb.dtype == torch.bfloat16
                layers, norm, head = build_llm_layers(LL, QL, spec, sd)
                E16 = sd["llm.model.embed_tokens.weight"].bfloat16()
                lg0, tok, lg1, S = run_llm(QL, spec, layers, norm, head, emb, "audiovisual", lambda t: E16[t][None, None])
        finally:
            torch.Tensor.cuda = cuda
            torch.set_default_dtype(dflt)
        # the fp32 oracle on the SAME (bf16-rounded) inputs: how far the reference's bf16 run is from fp32 arithmetic
        orc = O.OmniOracle(sd, spec.oracle_cfg())
        inp32 = {k: (v.float() if v.is_floating_point() else v) for k, v in inp.items()}
        with torch.no_grad():
            a32, v32 = orc.encode_audio(inp32["audio"], 640 * Tv, ra, mel_bf16=True), orc.encode_video(inp32["video"], rv)
            e32 = orc.prepare_inputs(inp32, "audiovisual", ra, rv, mel_bf16=True)
            oc = [None] * spec.llm.layers
            mod32 = "audiovisual" if spec.is_task_specific else None
            lg32 = orc.llm.forward(e32, 0, oc, mod32)[0, -1]
            lg32_1 = orc.llm.forward(orc.llm.embed(torch.tensor([[tok]])), S, oc, mod32)[0, -1]

        def d(x, y):
            e = (x.float() - y.float()).abs()
            return f"max {e.max().item():.3e} mean {e.mean().item():.3e} (scale {y.abs().mean().item():.2e})"
        print(f"[{name}] reference bf16 vs fp32 oracle: audio feats {d(feats['audio'], a32)} | video feats {d(feats['video'], v32)} | "
              f"inputs_embeds {d(emb, e32)} | prefill logits {d(lg0, lg32)} | step logits {d(lg1, lg32_1)}; S={S}, greedy token {tok} (fp32: {int(lg32.argmax())})")
        out.update({f"{name}_Tv": Tv, f"{name}_ra": ra, f"{name}_rv": rv,
                    f"{name}_audio_feats": feats["audio"][0].float().numpy(), f"{name}_video_feats": feats["video"][0].float().numpy(),
                    f"{name}_embeds": emb[0].float().numpy(), f"{name}_logits": lg0.numpy(), f"{name}_token": tok, f"{name}_step_logits": lg1.numpy()})
    np.savez_compressed(os.path.join(GOLD, "ref_bf16.npz"), **out)
    print("written", os.path.join(GOLD, "ref_bf16.npz"))


if __name__ == "__main__":
    main()
