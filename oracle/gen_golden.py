#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/*.npz (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPYCACHEPREFIX=/tmp/pyc python oracle/gen_golden.py

What pins what:
  * golden/avhubert_*.npz : outputs of the REAL vendored classes imported from /root/reference
        (AVHubertModel.extract_finetune, hubert.py:695-755 -> ResEncoder resnet.py:156-164 ->
        TransformerEncoder wav2vec2.py:862-905 -> MultiheadAttention.forward / forward_lora
        multihead_attention.py:160-192 / 389-672), LoRA monkey-patched exactly as modeling_OmniAVSR.py:125-142.
  * golden/whisper_*.npz, golden/llm_*.npz : outputs of the installed `transformers` (5.15; the reference pins
        4.43.1 which is not vendored): WhisperFeatureExtractor, WhisperEncoder, LlamaForCausalLM /
        Qwen2ForCausalLM with the LoRA term merged into q_proj / v_proj (algebraically identical to
        Llama_LoRA.py:250-259), `generate(do_sample=False)` greedy and beam.
Weights are NOT stored: they come from `omni_avsr_amd.synth.synth_state_dict(tiny_spec(...), seed)` which is
deterministic; the fixtures hold the seeds, the inputs' seeds and the expected outputs only.
The script also asserts, on the spot, that oracle/torch_oracle.py reproduces every fixture.
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from omni_avsr_amd.config import tiny_spec  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import ref_import as R  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 1234


def _vid(T, seed):
    return torch.randn(1, 1, T, 88, 88, generator=torch.Generator().manual_seed(seed))


def gen_avhubert(use_lora: bool):
    spec = tiny_spec(use_lora_avhubert=use_lora)
    a = spec.avhubert
    sd = synth_state_dict(spec, seed=SEED, modalities=("video",))
    cfg = R.avhubert_cfg(embed=a.dim, layers=a.layers, heads=a.heads, ffn=a.ffn, conv_pos=a.conv_pos, conv_pos_groups=a.conv_pos_groups)
    model = R.build_reference_avhubert(cfg, use_lora)
    ref_sd = {k[len("video_encoder."):]: v for k, v in sd.items() if k.startswith("video_encoder.")}
    own = model.state_dict()
    missing = [k for k in own if k not in ref_sd]
    for k in missing:                      # buffers the synthetic checkpoint does not carry (e.g. untied final_proj variants)
        ref_sd[k] = own[k]
    model.load_state_dict(ref_sd, strict=True)
    T = 11
    video = _vid(T, 7)
    with torch.no_grad():
        x, _, layers = model.extract_finetune(source={"video": video, "audio": None})
        feat = model.feature_extractor_video.resnet(video)                     # [1,512,T]
    out = dict(seed=SEED, video_seed=7, T=T, use_lora=int(use_lora), x=x[0].numpy(), resnet=feat[0].numpy(),
               layers=np.stack([l[:, 0].numpy() for l in layers]), missing=np.array(len(missing)))
    # oracle must reproduce the real classes
    with torch.no_grad():
        ox, st = O.avhubert_extract_finetune(sd, "video_encoder.", video, spec.oracle_cfg()["avhubert"], return_stages=True)
    e = (ox - x).abs().max().item()
    e_res = (st["resnet"] - feat).abs().max().item()
    e_l = max((st["layers"][i][0] - layers[i][:, 0]).abs().max().item() for i in range(len(layers)))
    print(f"avhubert lora={use_lora}: oracle vs REAL reference classes: final {e:.2e} resnet {e_res:.2e} layers {e_l:.2e}")
    assert e < 2e-5 and e_res < 2e-5 and e_l < 2e-4
    np.savez_compressed(os.path.join(GOLD, f"avhubert_lora{int(use_lora)}.npz"), **out)


def gen_whisper():
    from transformers import WhisperConfig, WhisperFeatureExtractor
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    spec = tiny_spec()
    w = spec.whisper
    sd = synth_state_dict(spec, seed=SEED, modalities=("audio",))
    enc = WhisperEncoder(WhisperConfig(d_model=w.dim, encoder_layers=w.layers, encoder_attention_heads=w.heads, encoder_ffn_dim=w.ffn,
                                       num_mel_bins=80, max_source_positions=1500, activation_function="gelu")).eval()
    enc.load_state_dict({k[len("audio_encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.")}, strict=True)
    n = 9000
    wave = torch.randn(1, n, generator=torch.Generator().manual_seed(11))
    wave = torch.nn.functional.layer_norm(wave, (n,))
    fe = WhisperFeatureExtractor()                     # default = whisper-* preprocessor (80 mels, n_fft 400, hop 160, 30 s)
    feats = fe(wave.numpy()[0], sampling_rate=16000, return_tensors="pt").input_features   # [1,80,3000]
    with torch.no_grad():
        hid = enc(feats).last_hidden_state
    omel = O.whisper_log_mel(wave)
    e_mel = (omel - feats).abs().max().item()
    with torch.no_grad():
        ohid = O.whisper_encoder(sd, "audio_encoder.", feats, w.layers, w.heads)
    e_enc = (ohid - hid).abs().max().item()
    fb = O.slaney_mel_filters()
    e_fb = np.abs(fb - fe.mel_filters).max()
    print(f"whisper: oracle vs transformers: log-mel {e_mel:.2e} encoder {e_enc:.2e} mel filterbank {e_fb:.2e}")
    assert e_mel < 2e-4 and e_enc < 2e-4 and e_fb < 1e-6
    np.savez_compressed(os.path.join(GOLD, "whisper.npz"), seed=SEED, wave_seed=11, n=n, mel_head=feats[0, :, :80].numpy(),
                        mel_tail_value=feats[0, 0, -1].numpy(), hid_head=hid[0, :64].numpy(), hid_sum=hid[0].sum(0).numpy())


def _hf_llm(spec, sd, task):
    """HF model with LoRA merged into q/v weights (W + s*B A)."""
    from transformers import LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM
    l = spec.llm
    if l.family == "llama":
        cfg = LlamaConfig(vocab_size=l.vocab, hidden_size=l.hidden, intermediate_size=l.mlp, num_hidden_layers=l.layers,
                          num_attention_heads=l.heads, num_key_value_heads=l.kv_heads, head_dim=l.head_dim, rms_norm_eps=l.rms_eps,
                          rope_theta=l.rope_theta, rope_scaling=dict(rope_type="llama3", **l.rope_scaling), tie_word_embeddings=l.tie_embeddings,
                          attention_bias=False, max_position_embeddings=131072)
        m = LlamaForCausalLM(cfg)
    else:
        cfg = Qwen2Config(vocab_size=l.vocab, hidden_size=l.hidden, intermediate_size=l.mlp, num_hidden_layers=l.layers,
                          num_attention_heads=l.heads, num_key_value_heads=l.kv_heads, rms_norm_eps=l.rms_eps, rope_theta=l.rope_theta,
                          tie_word_embeddings=l.tie_embeddings, max_position_embeddings=32768, use_sliding_window=False)
        m = Qwen2ForCausalLM(cfg)
    hf = {}
    scale = spec.lora_scale()
    for k, v in sd.items():
        if not k.startswith("llm.") or "lora_" in k:
            continue
        hf[k[4:]] = v.clone()
    for i in range(l.layers):
        ap = f"llm.model.layers.{i}.self_attn."
        for which, proj in (("Q", "q_proj"), ("V", "v_proj")):
            sfx = f".{task}" if spec.is_task_specific else ""
            delta = sd[f"{ap}lora_up_{which}{sfx}.weight"] @ sd[f"{ap}lora_down_{which}{sfx}.weight"]
            if spec.shared_lora:
                delta = delta + sd[f"{ap}lora_up_{which}_shared.weight"] @ sd[f"{ap}lora_down_{which}_shared.weight"]
            hf[f"model.layers.{i}.self_attn.{proj}.weight"] = hf[f"model.layers.{i}.self_attn.{proj}.weight"] + delta * scale
    m.load_state_dict(hf, strict=True)
    return m.eval()


def gen_llm(family: str):
    spec = tiny_spec(family=family, task_specific=True, shared=True)
    sd = synth_state_dict(spec, seed=SEED, modalities=())
    task = "audiovisual"
    m = _hf_llm(spec, sd, task)
    S = 23
    emb = torch.randn(1, S, spec.llm.hidden, generator=torch.Generator().manual_seed(5)) * 0.05
    llm = O.LLMOracle(sd, spec.oracle_cfg()["llm"])
    with torch.no_grad():
        ref_logits = m(inputs_embeds=emb).logits[0]
        o_logits = llm.forward(emb, 0, [None] * spec.llm.layers, task)[0]
    e = (o_logits - ref_logits).abs().max().item()
    print(f"{family}: oracle (unmerged LoRA) vs transformers (merged LoRA): logits max err {e:.2e}")
    assert e < 5e-4
    eos, pad = spec.ids["eos"], spec.ids["pad"]
    out = dict(seed=SEED, emb_seed=5, S=S, logits_last=ref_logits[-1].numpy())
    for K in (1, 3, 15):
        with torch.no_grad():
            kw = dict(inputs_embeds=emb, max_new_tokens=12, num_beams=K, do_sample=False, eos_token_id=eos, pad_token_id=pad,
                      early_stopping=False, length_penalty=1.0)
            if family == "llama":
                kw["bos_token_id"] = spec.ids["bos"]
            hf_ids = m.generate(**kw)[0].tolist()
            if K == 1:
                mine = llm.generate_greedy(emb, 12, eos, pad, task)
            else:
                mine = llm.generate_beam(emb, 12, K, eos, pad, task)
        print(f"  K={K}: transformers {hf_ids}\n        oracle       {mine}")
        assert hf_ids == mine, f"generate mismatch K={K}"
        out[f"ids_k{K}"] = np.array(hf_ids)
    # exercise the EOS / finished-hypothesis logic: declare tokens the model likes to emit to be EOS
    greedy = out["ids_k1"].tolist()
    for j, alt_eos in enumerate(sorted(set(greedy[1:6]))):
        for K in (1, 3, 15):
            with torch.no_grad():
                kw = dict(inputs_embeds=emb, max_new_tokens=12, num_beams=K, do_sample=False, eos_token_id=alt_eos, pad_token_id=pad,
                          early_stopping=False, length_penalty=1.0)
                hf_ids = m.generate(**kw)[0].tolist()
                mine = llm.generate_greedy(emb, 12, alt_eos, pad, task) if K == 1 else llm.generate_beam(emb, 12, K, alt_eos, pad, task)
            hf_trim = hf_ids[: hf_ids.index(alt_eos) + 1] if alt_eos in hf_ids else hf_ids
            print(f"  eos={alt_eos} K={K}: transformers {hf_ids} oracle {mine}")
            assert hf_trim == mine, f"generate mismatch eos={alt_eos} K={K}"
            out[f"alt{j}_eos"] = np.array(alt_eos)
            out[f"alt{j}_ids_k{K}"] = np.array(mine)
    np.savez_compressed(os.path.join(GOLD, f"llm_{family}.npz"), **out)


def main():
    assert R.reference_available(), "needs /root/reference"
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    gen_avhubert(True)
    gen_avhubert(False)
    gen_whisper()
    gen_llm("llama")
    gen_llm("qwen2")
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
