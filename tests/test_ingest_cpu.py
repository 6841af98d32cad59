"""CPU: checkpoint ingestion — round trip of a reference-layout checkpoint through disk, lenient fairseq unpickling,
safetensors directories, architecture inference from shapes."""
import os
import pickle
import sys
import types

import torch

from omni_avsr_amd import ingest
from omni_avsr_amd.config import tiny_spec
from omni_avsr_amd.synth import synth_state_dict


def test_model_avg_roundtrip_and_infer_spec(tmp_path):
    spec = tiny_spec(family="llama", task_specific=True, shared=True, matryoshka=True)
    sd = synth_state_dict(spec, seed=3)
    p = tmp_path / "model_avg_10.pth"
    torch.save({"model." + k: v for k, v in sd.items()}, p)            # un-stripped Lightning prefix must be tolerated
    got = ingest.load_omni_checkpoint(str(p))
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    import omni_avsr_amd.config as C
    C.LLM_SPECS["tiny/llama"] = spec.llm
    s2 = ingest.infer_spec(got, "tiny/llama", lora_alpha=spec.lora_alpha, downsample_ratio_audio=[2, 4], downsample_ratio_video=[1, 2],
                           ids=spec.ids, max_dec_tokens=8, num_beams=1)
    assert (s2.whisper.dim, s2.whisper.layers, s2.whisper.ffn) == (128, 2, 256)
    assert (s2.avhubert.dim, s2.avhubert.layers, s2.avhubert.conv_pos, s2.avhubert.conv_pos_groups, s2.avhubert.use_lora) == (128, 2, 16, 4, True)
    assert s2.lora_rank == spec.lora_rank and s2.is_task_specific and s2.shared_lora and s2.is_matryoshka and not s2.is_single_matry_projector
    assert s2.intermediate_size == spec.intermediate_size and s2.prompt_len == spec.prompt_len
    assert s2.proj_has_layernorm("audio") == spec.proj_has_layernorm("audio")


def test_fairseq_pickle_without_fairseq(tmp_path):
    """The AV-HuBERT .pt pickles omegaconf / fairseq objects in 'cfg'; loading must not need those packages."""
    mod = types.ModuleType("fake_omegaconf_pkg")
    exec("class DictConfig:\n    def __init__(self):\n        self.content = {'encoder_layers': 24}\n", mod.__dict__)
    mod.DictConfig.__module__ = "fake_omegaconf_pkg"
    mod.DictConfig.__qualname__ = "DictConfig"
    sys.modules["fake_omegaconf_pkg"] = mod
    state = {"cfg": {"model": mod.DictConfig()}, "model": {"encoder.layer_norm.weight": torch.ones(4), "mask_emb": torch.zeros(3)}, "task_state": {}}
    p = tmp_path / "avhubert.pt"
    torch.save(state, p)
    del sys.modules["fake_omegaconf_pkg"]                               # the class can no longer be imported
    sd = ingest.load_fairseq_avhubert(str(p))
    assert set(sd) == {"video_encoder.encoder.layer_norm.weight", "video_encoder.mask_emb"}
    assert torch.equal(sd["video_encoder.encoder.layer_norm.weight"], torch.ones(4))


def test_hf_safetensors_dir(tmp_path):
    from safetensors.torch import save_file
    d = tmp_path / "llm"
    os.makedirs(d)
    save_file({"model.embed_tokens.weight": torch.arange(6.0).reshape(3, 2)}, str(d / "model-00001-of-00002.safetensors"))
    save_file({"lm_head.weight": torch.ones(3, 2)}, str(d / "model-00002-of-00002.safetensors"))
    import json
    json.dump({"weight_map": {"model.embed_tokens.weight": "model-00001-of-00002.safetensors", "lm_head.weight": "model-00002-of-00002.safetensors"}},
              open(d / "model.safetensors.index.json", "w"))
    sd = ingest.load_hf_safetensors(str(d), prefix="llm.")
    assert set(sd) == {"llm.model.embed_tokens.weight", "llm.lm_head.weight"}


class _Boom:
    def __reduce__(self):
        import os as _os
        return (_os.system, ("touch " + _Boom.marker,))


def test_hostile_pickle_runs_nothing(tmp_path):
    """A checkpoint that names os.system (or any global outside the allow-list) must load as inert data, not execute."""
    _Boom.marker = str(tmp_path / "pwned")
    p = tmp_path / "evil.pth"
    torch.save({"model.w": torch.ones(2), "payload": _Boom()}, p)
    sd = ingest.load_omni_checkpoint(str(p))
    assert not os.path.exists(_Boom.marker)
    assert set(sd) == {"w"} and torch.equal(sd["w"], torch.ones(2))
    for mod, name in (("os", "system"), ("builtins", "eval"), ("builtins", "exec"), ("subprocess", "Popen"), ("posix", "system")):
        assert not ingest._safe_global(mod, name)
