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
