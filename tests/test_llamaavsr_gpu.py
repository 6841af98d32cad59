"""-m gpu: the `modeling_LlamaAVSR` drop-in against the CPU oracle: single prompt embedded from the LLM table,
non-task-specific LoRA, plain projector LayerNorm, test_ratio_matry = (video rate, audio rate)."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd.config import LoRA_config, tiny_spec  # noqa: E402
from omni_avsr_amd.modeling_llamaavsr import AVSR_LLMs  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402
from test_stages_gpu import inputs  # noqa: E402


@pytest.mark.parametrize("family", ["llama", "qwen2"])
@pytest.mark.parametrize("matry,modality", [(True, "audiovisual"), (True, "video"), (False, "audiovisual"), (False, "audio")])
def test_llamaavsr_shim_token_ids_f32(family, matry, modality):
    spec = dataclasses.replace(tiny_spec(family, matryoshka=matry, task_specific=False), projector_layernorm_plain=True)
    prompt_ids = [17, 5, 301, 44, 9]
    spec = dataclasses.replace(spec, prompt_len={t: len(prompt_ids) for t in ("audio", "video", "audiovisual")})
    sd = synth_state_dict(spec, seed=4321)
    # projector Sequential index 3 must be a real LayerNorm in this variant (the Omni per-rate quirk would have dropped it)
    assert any(k.endswith(".3.weight") for k in sd if "_proj." in k)
    ckpt = {k: v for k, v in sd.items() if not k.startswith("prompt_")}        # a LlamaAVSR checkpoint carries no prompt buffers
    ra_all, rv_all = spec.rates("audio"), spec.rates("video")
    model = AVSR_LLMs(modality, "large" if spec.avhubert and spec.avhubert.dim >= 1024 else "base", spec.avhubert.use_lora if spec.avhubert else False,
                      "tiny", spec.llm.hidden, spec.intermediate_size, None, "unused", spec.ids["pad"],
                      ra_all if matry else ra_all[0], rv_all if matry else rv_all[0], "tiny", spec.compression_mode, [], spec.max_dec_tokens, 1,
                      PETF_LLM_name="lora", peft_config_llm=LoRA_config(spec.lora_rank, spec.lora_alpha), is_matryoshka=matry,
                      prompt_ids=prompt_ids, spec=spec, dtype="f32")
    model.load_state_dict(ckpt)
    emb = sd["llm.model.embed_tokens.weight"][torch.tensor(prompt_ids)].unsqueeze(0)
    ref_sd = dict(ckpt, prompt_audio=emb, prompt_video=emb, prompt_audiovisual=emb)
    orc = O.OmniOracle(ref_sd, spec.oracle_cfg())
    Tv, Ts = [12, 9], [7680, 5000]
    video, audio = inputs(Tv=Tv, Ts=Ts)
    ra, rv = ra_all[-1], rv_all[-1]
    tokens = torch.full((2, 1), spec.ids["bos"]) if family == "llama" else torch.zeros(2, 0, dtype=torch.long)
    batch = dict(video=video, audio=audio, lengths=torch.tensor(Ts), video_lengths=torch.tensor(Tv), video_lens=torch.tensor(Tv), tokens=tokens)
    ratio = ((rv, ra) if modality == "audiovisual" else (ra if modality == "audio" else rv)) if matry else None
    got = model(batch, is_trainval=False, test_ratio_matry=ratio)
    ref = orc.forward(batch, modality, ra if matry else ra_all[0], rv if matry else rv_all[0], num_beams=1)
    for b in range(2):
        want = ref[b]
        assert got[b, : len(want)].tolist() == want, f"utt {b}: got {got[b].tolist()} want {want}"
