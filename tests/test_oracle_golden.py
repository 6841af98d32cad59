"""CPU: the oracle (oracle/torch_oracle.py) against the committed golden vectors.
avhubert_*.npz were produced by the REAL vendored reference classes, whisper/llm_*.npz by the installed transformers
(oracle/gen_golden.py; generator script committed, weights regenerated from the deterministic synthetic checkpoint)."""
import os

import numpy as np
import pytest
import torch

from omni_avsr_amd.config import tiny_spec
from omni_avsr_amd.synth import synth_state_dict
from oracle import torch_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("use_lora", [True, False])
def test_avhubert_against_real_reference_classes(use_lora):
    g = np.load(os.path.join(GOLD, f"avhubert_lora{int(use_lora)}.npz"))
    spec = tiny_spec(use_lora_avhubert=use_lora)
    sd = synth_state_dict(spec, seed=int(g["seed"]), modalities=("video",))
    video = torch.randn(1, 1, int(g["T"]), 88, 88, generator=torch.Generator().manual_seed(int(g["video_seed"])))
    with torch.no_grad():
        x, st = O.avhubert_extract_finetune(sd, "video_encoder.", video, spec.oracle_cfg()["avhubert"], return_stages=True)
    assert np.abs(x[0].numpy() - g["x"]).max() < 1e-4
    assert np.abs(st["resnet"][0].numpy() - g["resnet"]).max() < 1e-4
    for i in range(spec.avhubert.layers):
        assert np.abs(st["layers"][i][0].numpy() - g["layers"][i]).max() < 1e-3


def test_whisper_against_transformers():
    g = np.load(os.path.join(GOLD, "whisper.npz"))
    spec = tiny_spec()
    sd = synth_state_dict(spec, seed=int(g["seed"]), modalities=("audio",))
    n = int(g["n"])
    wave = torch.nn.functional.layer_norm(torch.randn(1, n, generator=torch.Generator().manual_seed(int(g["wave_seed"]))), (n,))
    mel = O.whisper_log_mel(wave)
    assert np.abs(mel[0, :, :80].numpy() - g["mel_head"]).max() < 1e-4
    assert abs(float(mel[0, 0, -1]) - float(g["mel_tail_value"])) < 1e-5
    with torch.no_grad():
        hid = O.whisper_encoder(sd, "audio_encoder.", mel, spec.whisper.layers, spec.whisper.heads)
    assert np.abs(hid[0, :64].numpy() - g["hid_head"]).max() < 1e-3
    assert np.abs(hid[0].sum(0).numpy() - g["hid_sum"]).max() < 5e-2


@pytest.mark.parametrize("family", ["llama", "qwen2"])
def test_llm_logits_and_generate_against_transformers(family):
    g = np.load(os.path.join(GOLD, f"llm_{family}.npz"))
    spec = tiny_spec(family=family, task_specific=True, shared=True)
    sd = synth_state_dict(spec, seed=int(g["seed"]), modalities=())
    llm = O.LLMOracle(sd, spec.oracle_cfg()["llm"])
    emb = torch.randn(1, int(g["S"]), spec.llm.hidden, generator=torch.Generator().manual_seed(int(g["emb_seed"]))) * 0.05
    with torch.no_grad():
        logits = llm.forward(emb, 0, [None] * spec.llm.layers, "audiovisual", last_only=True)[0, -1]
        assert np.abs(logits.numpy() - g["logits_last"]).max() < 1e-3
        eos, pad = spec.ids["eos"], spec.ids["pad"]
        assert llm.generate_greedy(emb, 12, eos, pad, "audiovisual") == g["ids_k1"].tolist()
        for K in (3, 15):
            assert llm.generate_beam(emb, 12, K, eos, pad, "audiovisual") == g[f"ids_k{K}"].tolist()
        j = 0
        while f"alt{j}_eos" in g:          # EOS / finished-hypothesis paths
            e = int(g[f"alt{j}_eos"])
            assert llm.generate_greedy(emb, 12, e, pad, "audiovisual") == g[f"alt{j}_ids_k1"].tolist()
            assert llm.generate_beam(emb, 12, 3, e, pad, "audiovisual") == g[f"alt{j}_ids_k3"].tolist()
            assert llm.generate_beam(emb, 12, 15, e, pad, "audiovisual") == g[f"alt{j}_ids_k15"].tolist()
            j += 1
        assert j >= 2


def test_compressor_semantics():
    """modeling_OmniAVSR.py:480-483 (AvgPool1d floor) and :507-514 (stack, remainder dropped)."""
    x = torch.arange(2 * 7 * 3, dtype=torch.float32).reshape(2, 7, 3)
    ref = torch.nn.functional.avg_pool1d(x.transpose(1, 2), 3).transpose(1, 2)
    assert torch.equal(O.compress(x, 3, "avg-pooling"), ref)
    st = O.compress(x, 3, "stack")
    assert st.shape == (2, 2, 9) and torch.equal(st[0, 1], x[0, 3:6].reshape(-1))


def test_glue_prepare_inputs_against_executed_reference():
    """tests/golden/glue_prepare_inputs.npz: `inputs_embeds` returned by the reference's OWN `AVSR_LLMs.prepare_inputs` /
    `encode_audio` / `encode_video` (modeling_OmniAVSR.py:397-458,528-558,461-483), executed in the build container by
    oracle/gen_golden_glue.py -- six constructor variants (per-rate / single projector, swallowed LayerNorm, stack mode, Llama with
    BOS / Qwen without, audio-only, video-only)."""
    from oracle.gen_golden_glue import CASES, CASE_TV, glue_inputs
    g = np.load(os.path.join(GOLD, "glue_prepare_inputs.npz"))
    for name, (kw, _llm_name, modality, ra, rv) in CASES.items():
        spec = tiny_spec(**kw)
        sd = synth_state_dict(spec, seed=int(g["seed"]))
        orc = O.OmniOracle(sd, spec.oracle_cfg())
        inp = glue_inputs(spec, int(g["input_seed"]), Tv=CASE_TV.get(name, int(g["Tv"])), Ts=int(g["Ts"]))
        if not spec.is_matryoshka:
            ra, rv = spec.rates("audio")[0], spec.rates("video")[0]
        with torch.no_grad():
            emb = orc.prepare_inputs(inp, modality, ra, rv, mel_bf16=True)      # the reference casts the log-mel to bf16 (:534)
        assert emb.shape[1:] == g[name].shape, name
        assert np.abs(emb[0].numpy() - g[name]).max() < 2e-4, name


@pytest.mark.parametrize("family", ["llama", "qwen2"])
def test_decoder_layer_against_executed_reference(family):
    """tests/golden/glue_decoder_layer.npz: outputs of the reference's own `LlamaDecoderLayer_lora.forward` ->
    `LlamaSdpaAttention_lora.forward` (Llama_LoRA.py:587-655,214-316) and the Qwen twins (Qwen_LoRA.py:388-450,529-619) with
    task-specific + shared LoRA: a prefill over S positions and one decode step against the cache."""
    g = np.load(os.path.join(GOLD, "glue_decoder_layer.npz"))
    spec = tiny_spec(family=family, head_dim=64 if family == "llama" else 128, task_specific=True, shared=True)
    sd = synth_state_dict(spec, seed=int(g["seed"]), modalities=())
    sd1 = {k: v for k, v in sd.items() if not k.startswith("llm.model.layers.1.")}
    llm = O.LLMOracle(sd1, dict(spec.oracle_cfg()["llm"], layers=1))
    gen = torch.Generator().manual_seed(int(g["input_seed"]))
    S = int(g["S"])
    x = torch.randn(1, S, spec.llm.hidden, generator=gen)
    x1 = torch.randn(1, 1, spec.llm.hidden, generator=gen)
    cache = [None]
    with torch.no_grad():
        _, h = llm.forward(x, 0, cache, "audiovisual", return_hidden=True)
        _, h1 = llm.forward(x1, S, cache, "audiovisual", return_hidden=True)
    assert np.abs(h[0][0].numpy() - g[f"{family}_prefill"]).max() < 2e-5
    assert np.abs(h1[0][0].numpy() - g[f"{family}_step"]).max() < 2e-5


@pytest.mark.parametrize("name,kw", [("llama", dict(family="llama", task_specific=True, shared=True, matryoshka=True)),
                                     ("qwen", dict(family="qwen2", head_dim=128, matryoshka=True, single_projector=True))])
def test_oracle_and_its_bf16_emulation_against_the_references_bf16_run(name, kw):
    """tests/golden/ref_bf16.npz (oracle/gen_golden_bf16.py): the reference's classes executed in torch.bfloat16 as `bf16-true` runs them.
    The fp32 oracle must sit at bf16 distance from them (not at fp32 distance: that would mean the fixture is not a bf16 run; not far: that
    would mean different mathematics), and the bf16-EMULATING oracle -- the model of this engine's rounding points -- must be no further
    from the fp32 oracle than the reference's bf16 run is (the same bar tests/test_bf16_reference_gpu.py applies to the engine itself)."""
    z = np.load(os.path.join(GOLD, "ref_bf16.npz"))
    spec = tiny_spec(**kw)
    sd = synth_state_dict(spec, seed=int(z["seed"]))
    Tv, ra, rv = int(z[f"{name}_Tv"]), int(z[f"{name}_ra"]), int(z[f"{name}_rv"])
    g = torch.Generator().manual_seed(int(z["input_seed"]))
    video = torch.randn(1, Tv, 1, 88, 88, generator=g).bfloat16().float()
    audio = torch.randn(1, 640 * Tv, 1, generator=g).bfloat16().float()
    tokens = torch.full((1, 1), spec.ids["bos"]) if spec.llm.family == "llama" else torch.zeros(1, 0, dtype=torch.long)
    inp = dict(video=video, audio=audio, lengths=torch.tensor([640 * Tv]), tokens=tokens)
    orc = O.OmniOracle(sd, spec.oracle_cfg())
    mod = "audiovisual" if spec.is_task_specific else None

    def run():
        with torch.no_grad():
            e = orc.prepare_inputs(inp, "audiovisual", ra, rv, mel_bf16=True)
            return e[0], orc.llm.forward(e, 0, [None] * spec.llm.layers, mod)[0, -1]
    e32, lg32 = run()
    with O.bf16_emulation():
        e16, lg16 = run()
    for what, truth, emu, ref in (("inputs_embeds", e32, e16, z[f"{name}_embeds"]), ("prefill logits", lg32, lg16, z[f"{name}_logits"])):
        d_ref = np.abs(ref - truth.numpy())
        d_emu = (emu - truth).abs().numpy()
        scale = float(truth.abs().mean())
        assert 2e-4 * scale < d_ref.mean() < 0.05 * scale, f"{name} {what}: the reference's bf16 run is {d_ref.mean():.2e} from the fp32 oracle (scale {scale:.2e})"
        assert d_emu.mean() <= 1.5 * d_ref.mean(), f"{name} {what}: emulation {d_emu.mean():.2e} vs reference bf16 {d_ref.mean():.2e}"
    assert int(lg32.argmax()) == int(z[f"{name}_token"])
