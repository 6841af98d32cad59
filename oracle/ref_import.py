"""TEST INFRASTRUCTURE ONLY (oracle side). Never imported by the product path.

Imports the *real* reference classes from /root/reference (build container only; the
directory does not exist on the GPU box) so that `oracle/gen_golden.py` can produce
golden vectors and `oracle/torch_oracle.py` can be pinned against them.

The reference needs omegaconf / hydra / python_speech_features / cv2, none of which is
installed here.  We register tiny in-memory stand-in *modules* for those third-party
packages (they are only touched at import time for dataclass defaults); no reference
source is copied or modified.  See SURVEY.md §8(c).
"""
import os
import sys
import types
import argparse

REF_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "av_hubert", "fairseq", "fairseq"))


def _install_stubs():
    if "omegaconf" in sys.modules and getattr(sys.modules["omegaconf"], "_oavsr_stub", False):
        return
    om = types.ModuleType("omegaconf")
    om._oavsr_stub = True
    om.II = lambda s: None
    om.MISSING = "???"

    class DictConfig(dict):
        pass

    class ListConfig(list):
        pass

    class _OC:
        @staticmethod
        def create(x=None):
            return DictConfig(x or {})

        @staticmethod
        def set_struct(*a, **k):
            return None

        @staticmethod
        def is_config(x):
            return isinstance(x, (DictConfig, ListConfig))

        @staticmethod
        def to_container(x, **k):
            return dict(x)

        @staticmethod
        def merge(*a):
            out = DictConfig()
            for d in a:
                out.update(d)
            return out

    class _ctx:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    om.DictConfig = DictConfig
    om.ListConfig = ListConfig
    om.OmegaConf = _OC
    om.open_dict = _ctx
    om_utils = types.ModuleType("omegaconf._utils")
    om_utils.is_primitive_type = lambda t: t in (int, float, str, bool)
    om._utils = om_utils
    sys.modules["omegaconf"] = om
    sys.modules["omegaconf._utils"] = om_utils

    hy = types.ModuleType("hydra")
    hy._oavsr_stub = True
    core = types.ModuleType("hydra.core")
    gh = types.ModuleType("hydra.core.global_hydra")

    class GlobalHydra:
        @staticmethod
        def instance():
            return GlobalHydra()

        def is_initialized(self):
            return False

        def clear(self):
            pass

    gh.GlobalHydra = GlobalHydra
    cs = types.ModuleType("hydra.core.config_store")

    class ConfigStore:
        _inst = None

        @staticmethod
        def instance():
            if ConfigStore._inst is None:
                ConfigStore._inst = ConfigStore()
            return ConfigStore._inst

        def store(self, *a, **k):
            pass

    cs.ConfigStore = ConfigStore
    exp = types.ModuleType("hydra.experimental")
    exp.compose = lambda *a, **k: None
    exp.initialize = _ctx
    hy.core = core
    hy.experimental = exp
    hy.compose = exp.compose
    hy.initialize = _ctx
    core.global_hydra = gh
    core.config_store = cs
    for name, mod in [("hydra", hy), ("hydra.core", core), ("hydra.core.global_hydra", gh),
                      ("hydra.core.config_store", cs), ("hydra.experimental", exp)]:
        sys.modules[name] = mod

    psf = types.ModuleType("python_speech_features")
    psf.logfbank = lambda *a, **k: None
    sys.modules["python_speech_features"] = psf
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")


def import_reference():
    """Returns a namespace with the real reference classes on the video path."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected on the build container only)")
    sys.dont_write_bytecode = True
    _install_stubs()
    for p in (REF_ROOT, os.path.join(REF_ROOT, "av_hubert", "fairseq")):
        if p not in sys.path:
            sys.path.insert(0, p)
    argv = sys.argv
    sys.argv = ["x"]  # hubert.py:29 picks absolute imports when len(sys.argv)==1
    try:
        import fairseq  # noqa: F401
        from fairseq.models.wav2vec.wav2vec2 import TransformerEncoder
        from fairseq.modules.multihead_attention import MultiheadAttention
        from av_hubert.avhubert.hubert import AVHubertModel
        from av_hubert.avhubert.resnet import ResEncoder
    finally:
        sys.argv = argv
    return types.SimpleNamespace(
        TransformerEncoder=TransformerEncoder, MultiheadAttention=MultiheadAttention,
        AVHubertModel=AVHubertModel, ResEncoder=ResEncoder)


def avhubert_cfg(embed=1024, layers=24, heads=16, ffn=4096, conv_pos=128, conv_pos_groups=16):
    """argparse.Namespace mirroring av_hubert/avhubert/conf/pretrain/large_vox_iter5.yaml:70-101
    and the dataclass defaults of hubert.py:60-300 (only the fields the ctor reads)."""
    return argparse.Namespace(
        label_rate=25, sample_rate=25, sub_encoder_layers=0, resnet_relu_type="prelu", resnet_weights=None,
        audio_feat_dim=104, modality_dropout=0.0, audio_dropout=0.0, modality_fuse="concat",
        encoder_embed_dim=embed, encoder_layers=layers, encoder_ffn_embed_dim=ffn,
        encoder_attention_heads=heads, activation_fn="gelu", dropout=0.0, attention_dropout=0.0,
        activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0, dropout_features=0.0,
        final_dim=256, untie_final_proj=True, layer_norm_first=True, conv_pos=conv_pos,
        conv_pos_groups=conv_pos_groups, mask_prob_image=0.0, mask_prob_audio=0.0, mask_selection="static",
        mask_other=0, mask_length_image=5, mask_length_audio=10, no_mask_overlap=False, mask_min_space=1,
        mask_channel_prob=0.0, mask_channel_selection="static", mask_channel_other=0,
        mask_channel_length=10, no_mask_channel_overlap=False, mask_channel_min_space=1,
        feature_grad_mult=1.0, logit_temp=0.1, skip_masked=False, skip_nomask=False, sim_type="cosine",
        selection_type="same_seq", masking_type="input", target_glu=False, max_positions=100000,
    )


def build_reference_avhubert(cfg, use_lora: bool):
    """Instantiates the real AVHubertModel and (optionally) monkey-patches LoRA exactly as
    Omni_AVSR/modeling_OmniAVSR.py:125-142 does (rank=16 => inner round(D/16), scaling_lora=2)."""
    import torch.nn as nn
    ref = import_reference()
    task_cfg = argparse.Namespace(sample_rate=25)
    m = ref.AVHubertModel(cfg, task_cfg, [None])
    if use_lora:
        D = cfg.encoder_embed_dim
        for lyr in m.encoder.layers:
            lyr.apply_lora = True
            sa = lyr.self_attn
            sa.rank = 16
            sa.scaling_lora = 2
            r = round(D / sa.rank)
            sa.lora_down_Q = nn.Linear(D, r, bias=False)
            sa.lora_up_Q = nn.Linear(r, D, bias=False)
            sa.lora_down_V = nn.Linear(D, r, bias=False)
            sa.lora_up_V = nn.Linear(r, D, bias=False)
    m.eval()
    return m
