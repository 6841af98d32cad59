"""CPU: host-side logic of the product (no compute calls): config quirks, checkpoint packing algebra, C-ABI exports."""
import ctypes
import os
import re

import pytest
import torch

from omni_avsr_amd import lib as L
from omni_avsr_amd.config import LoRA_config, QwenLoRA_config, OmniSpec, spec_from_names, tiny_spec
from omni_avsr_amd.modeling import whisper_keep_tokens
from omni_avsr_amd.synth import synth_state_dict
from omni_avsr_amd.weights import build_blob, lora_k2, pack_state_dict, rope_inv_freq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "omni_avsr.h")).read()
    declared = set(re.findall(r"\b(oavsr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/omni_avsr.h but not exported"
    assert declared == set(L.SYMBOLS), f"ctypes table out of sync: {declared ^ set(L.SYMBOLS)}"
    assert lib.oavsr_abi_version() == L.ABI_VERSION == 4
    assert ctypes.sizeof(L.OavsrConfig) % 4 == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libomniavsr_hip.so")
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        L.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "omni-avsr_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{fn} imports the oracle"


def test_reference_config_objects_keep_field_names():
    c = LoRA_config(RANK=32, ALPHA=4, IS_LLAMA3=True, IS_TASK_SPECIFIC=True, SHARED_LORA=True)
    assert (c.RANK, c.ALPHA, c.IS_LLAMA3_2_3B) == (32, 4, False)
    q = QwenLoRA_config(RANK=32, IS_QWEN25_7B=True)
    assert q.IS_QWEN25_7B and not q.SHARED_LORA
    s = spec_from_names("meta-llama/Llama-3.2-1B", "openai/whisper-medium.en", "large", lora_rank=32, lora_alpha=4)
    assert s.lora_inner() == 64 and s.lora_scale() == 0.125            # README --rank 32 --alpha 4 (Llama_LoRA.py:119-125)
    s7 = spec_from_names("Qwen/Qwen2.5-7B", None, None, lora_rank=32)
    assert s7.lora_inner() == 112 and s7.kv_dim() == 3584 // 7         # Qwen_LoRA.py:464-475 gqa_factor table
    for name, gqa in (("Qwen/Qwen2.5-0.5B", 7), ("Qwen/Qwen2.5-1.5B", 6), ("Qwen/Qwen2.5-3B", 8), ("Qwen/Qwen2.5-14B", 5), ("Qwen/Qwen2.5-32B", 5)):
        sp = spec_from_names(name, None, None)
        assert sp.kv_dim() == sp.llm.hidden // gqa
    for name, div in (("meta-llama/Llama-3.2-1B", 4), ("meta-llama/Meta-Llama-3.1-8B", 4), ("meta-llama/Llama-3.2-3B", 3)):
        sp = spec_from_names(name, None, None)
        assert sp.kv_dim() == sp.llm.hidden // div                     # Llama_LoRA.py:143-156


@pytest.mark.parametrize("mode,matry,single,rm,exp_a,exp_v", [
    ("avg-pooling", False, False, False, True, True),     # :111,:196
    ("avg-pooling", False, False, True, False, False),    # :109,:194
    ("avg-pooling", True, True, False, True, True),       # :102,:186
    ("avg-pooling", True, False, False, False, False),    # :104,:188 LayerNorm swallowed as Linear's bias arg
    ("avg-pooling", True, False, True, False, False),
    ("stack", True, False, True, True, False),            # :74-75 polarity inverted for audio
    ("stack", True, False, False, False, False),          # :77 / :162
    ("stack", False, False, False, True, True),           # :85,:169
    ("stack", False, False, True, False, False),
])
def test_projector_layernorm_quirks(mode, matry, single, rm, exp_a, exp_v):
    s = tiny_spec(matryoshka=matry, compression_mode=mode)
    s.is_single_matry_projector, s.remove_layernorm_from_projector = single, rm
    assert s.proj_has_layernorm("audio") == exp_a and s.proj_has_layernorm("video") == exp_v


def test_whisper_keep_tokens_matches_reference_expression():
    for n in (0, 7999, 8000, 16000, 123457, 256000, 480000):
        assert whisper_keep_tokens(n) == max(int(n / 16000 * 50), 25)   # modeling_OmniAVSR.py:537


def test_packing_algebra_lora_and_qkv():
    """[x | x A_all^T] . [W_qkv | s B_block]^T  ==  reference's unmerged arithmetic (Llama_LoRA.py:246-259)."""
    spec = tiny_spec(task_specific=True, shared=True)
    sd = synth_state_dict(spec, seed=3, modalities=())
    pk = pack_state_dict(sd, spec, "f32")
    d, nq, kvd = spec.llm.hidden, spec.llm.heads * spec.llm.head_dim, spec.kv_dim()
    x = torch.randn(5, d, generator=torch.Generator().manual_seed(0))
    ap = "llm.model.layers.1.self_attn."
    for t, task in enumerate(("audio", "video", "audiovisual")):
        tt = x @ pk[f"l.L1.lora_dn.{t}.w"].T
        got = x @ pk["l.L1.qkv.w"].T + tt @ pk[f"l.L1.lora_up.{t}.w"].T
        s = spec.lora_scale()
        q = x @ sd[ap + "q_proj.weight"].T + ((x @ sd[ap + f"lora_down_Q.{task}.weight"].T) @ sd[ap + f"lora_up_Q.{task}.weight"].T
                                                + (x @ sd[ap + "lora_down_Q_shared.weight"].T) @ sd[ap + "lora_up_Q_shared.weight"].T) * s
        k = x @ sd[ap + "k_proj.weight"].T
        v = x @ sd[ap + "v_proj.weight"].T + ((x @ sd[ap + f"lora_down_V.{task}.weight"].T) @ sd[ap + f"lora_up_V.{task}.weight"].T
                                                + (x @ sd[ap + "lora_down_V_shared.weight"].T) @ sd[ap + "lora_up_V_shared.weight"].T) * s
        assert torch.allclose(got, torch.cat([q, k, v], 1), atol=1e-5)
    assert pk["l.L1.lora_dn.0.w"].shape[0] == lora_k2(spec) and lora_k2(spec) % 64 == 0
    assert pk["l.lm_head"].data_ptr() == pk["l.embed"].data_ptr()      # tied embeddings stored once


def test_packing_conv_layouts_and_blob():
    spec = tiny_spec()
    sd = synth_state_dict(spec, seed=5)
    pk = pack_state_dict(sd, spec, "bf16")
    w = sd["video_encoder.feature_extractor_video.resnet.trunk.layer2.0.conv1.weight"]
    p = pk["v.l2.0.conv1.w"].float().reshape(128, 3, 3, 64)
    assert torch.allclose(p[5, 1, 2, :], w[5, :, 1, 2].bfloat16().float())
    assert pk["v.stem.w"].shape == (64, 256) and float(pk["v.stem.w"][:, 245:].abs().max()) == 0.0
    assert pk["a.conv1.w"].shape[1] == 256 and pk["v.L0.qkv.b"].dtype == torch.float32
    assert float(pk["a.L0.qkv.b"][spec.whisper.dim:2 * spec.whisper.dim].abs().max()) == 0.0   # Whisper k_proj has no bias
    blob, man = build_blob(pk)
    names = [n for n, _, _ in man]
    assert len(set(names)) == len(names) and all(off % 256 == 0 for _, off, _ in man)
    off = dict((n, (o, b)) for n, o, b in man)
    assert off["l.lm_head"][0] == off["l.embed"][0]
    o, b = off["v.lnf.w"]
    assert torch.equal(blob[o:o + b].view(torch.float32), pk["v.lnf.w"])


def test_rope_inv_freq_llama3_matches_oracle():
    from oracle import torch_oracle as O
    sc = dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)
    assert torch.allclose(rope_inv_freq(64, 500000.0, sc), O.rope_inv_freq(64, 500000.0, sc))
    assert torch.allclose(rope_inv_freq(128, 1e6, None), O.rope_inv_freq(128, 1e6, None))


def test_synth_is_deterministic_and_complete():
    spec = tiny_spec(task_specific=True, shared=True)
    a, b = synth_state_dict(spec, seed=9), synth_state_dict(spec, seed=9)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert "llm.model.layers.0.self_attn.lora_up_V.audiovisual.weight" in a and "prompt_audiovisual" in a
    assert a["video_encoder.encoder.layers.0.self_attn.lora_down_Q.weight"].shape == (8, 128)


@pytest.mark.parametrize("mode", ["avg-pooling", "stack"])
@pytest.mark.parametrize("matry", [False, True])
@pytest.mark.parametrize("rm", [False, True])
def test_llamaavsr_projector_layernorm_is_plain(mode, matry, rm):
    """modeling_LlamaAVSR.py:60-108,150-196: Sequential(..., nn.LayerNorm) unless remove_layernorm_from_projector -- none of the
    Omni variant's quirks (LayerNorm swallowed as a Linear bias, inverted polarity)."""
    import dataclasses
    spec = dataclasses.replace(tiny_spec(matryoshka=matry, compression_mode=mode, remove_layernorm=rm), projector_layernorm_plain=True)
    assert spec.proj_has_layernorm("audio") == (not rm) and spec.proj_has_layernorm("video") == (not rm)


def test_llamaavsr_shim_prompt_ids_follow_reference_slicing():
    """prompt ids = tokenizer(prompt).input_ids[:, start:-1], start = 1 for Llama (drop BOS), 0 for Qwen (modeling_LlamaAVSR.py:283-285);
    test_ratio_matry for audiovisual is (video rate, audio rate) (:334,339)."""
    from omni_avsr_amd.modeling_llamaavsr import AVSR_LLMs

    class Tok:
        vocab = {"<audio>": 501, "</audio>": 502, "<video>": 503, "</video>": 504}

        def __call__(self, text, return_tensors=None):
            class R:
                input_ids = torch.tensor([[7, 11, 12, 13, 9]])
            return R()

    spec = tiny_spec("llama", matryoshka=True)
    for name, want in (("meta-llama/Llama-3.2-1B", [11, 12, 13]), ("Qwen/Qwen2.5-7B", [7, 11, 12, 13])):
        m = AVSR_LLMs("audiovisual", "large", False, name, spec.llm.hidden, 192, Tok(), "Transcribe.", 0, [2, 4], [2, 4], "w", "avg-pooling", [], 8, 1,
                      is_matryoshka=True, spec=spec, dtype="f32")
        assert m.prompt_ids == want
    seen = {}
    import omni_avsr_amd.modeling as M
    orig = M.AVSR_LLMs.forward
    try:
        M.AVSR_LLMs.forward = lambda self, inputs, **kw: seen.update(kw)
        m({}, is_trainval=False, test_ratio_matry=(4, 2))
    finally:
        M.AVSR_LLMs.forward = orig
    assert seen["test_ratio_matry_video"] == 4 and seen["test_ratio_matry_audio"] == 2 and seen["modality"] is None
