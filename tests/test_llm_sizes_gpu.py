"""-m gpu: EVERY LLM the reference's size tables name (lightning_OmniAVSR.py:28-37, Qwen_LoRA.py:464-475 `IS_QWEN25_*`, Llama_LoRA.py:119-125
`IS_LLAMA3_2_3B`) at FULL WIDTH -- two decoder layers, tiny encoders so the CPU oracle stays affordable (pattern:
tests/test_fullwidth_7b8b_gpu.py, which holds Qwen2.5-7B and Llama-3.1-8B; tests/test_fullsize_gpu.py holds Llama-3.2-1B whole):

  Qwen2.5-0.5B   hidden  896, 14 q / 2 kv heads x  64 (GQA group 7), MLP  4 864, LoRA inner 28  (K2 = 4 x 28 -> 128)
  Qwen2.5-1.5B   hidden 1536, 12 q / 2 kv heads x 128 (group 6),     MLP  8 960, LoRA inner 48  (K2 = 192)
  Qwen2.5-3B     hidden 2048, 16 q / 2 kv heads x 128 (group 8),     MLP 11 008, LoRA inner 64
  Qwen2.5-14B    hidden 5120, 40 q / 8 kv heads x 128 (group 5),     MLP 13 824, LoRA inner 160 (K2 = 640)
  Qwen2.5-32B    the 14B widths with MLP 27 648
  Llama-3.2-3B   hidden 3072, 24 q / 8 kv heads x 128 (group 3),     MLP  8 192, LoRA inner 96  (K2 = 384), tied lm_head, llama3 RoPE

What runs at these widths and nowhere else in the suite: prefill flash attention and decode attention with GQA groups 3 / 5 / 6 / 7, the
persistent / LDS-DMA GEMMs with K = 896 / 1536 / 3072 / 5120, LoRA K segments padded from 112 / 192 / 384 / 640, the decode-shaped GEMMs' split
factors at those (N, K).

  * f32: greedy AND HF-beam-search (K = 15) token ids bit-exact vs the CPU oracle;
  * bf16: prefill logits / decoded tokens within the bf16-emulation bound of tests/test_bf16_emulation_gpu.py."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd.config import LLAMA_IDS, LLM_SPECS, QWEN_IDS, tiny_spec  # noqa: E402
from omni_avsr_amd.modeling import OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

from test_fullwidth_7b8b_gpu import LOGIT_TOL, clips, tokens_for  # noqa: E402

MODELS = {"qwen25_0_5b": "Qwen/Qwen2.5-0.5B", "qwen25_1_5b": "Qwen/Qwen2.5-1.5B", "qwen25_3b": "Qwen/Qwen2.5-3B", "qwen25_14b": "Qwen/Qwen2.5-14B",
          "qwen25_32b": "Qwen/Qwen2.5-32B", "llama32_3b": "meta-llama/Llama-3.2-3B"}
INNER = {"qwen25_0_5b": 28, "qwen25_1_5b": 48, "qwen25_3b": 64, "qwen25_14b": 160, "qwen25_32b": 160, "llama32_3b": 96}


def wide_spec(name, max_new=6):
    llm = dataclasses.replace(LLM_SPECS[MODELS[name]], layers=2)
    spec = tiny_spec(family=llm.family, head_dim=llm.head_dim, task_specific=True, shared=True)
    spec.llm, spec.ids = llm, dict(QWEN_IDS if llm.family == "qwen2" else LLAMA_IDS)
    spec.lora_rank, spec.lora_alpha, spec.max_dec_tokens = 32, 4, max_new
    return spec


@pytest.mark.parametrize("name", sorted(MODELS))
def test_f32_ids_equal_the_oracle_and_bf16_within_the_emulation_bound(name):
    """(one test per model: the 1-2 B parameters of seeded weights are drawn once)"""
    spec = wide_spec(name)
    assert spec.lora_inner() == INNER[name]
    sd = synth_state_dict(spec, seed=1234)
    eng = OmniEngine(spec, dtype="f32").load_state_dict(sd)
    orc = O.OmniOracle(sd, spec.oracle_cfg())
    video, audio, Tv, Ts = clips(2, 3)
    inp = dict(video=video, audio=audio, lengths=torch.tensor(Ts), video_lens=torch.tensor(Tv), tokens=tokens_for(spec, 2))
    for beams in (1, 15):
        ids, lens = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 2, beams=beams, max_new=spec.max_dec_tokens)
        ids, lens = ids.cpu(), lens.cpu()
        with torch.no_grad():
            ref = orc.forward(inp, "audiovisual", 4, 2, num_beams=beams)
        got = [ids[b, : int(lens[b])].tolist() for b in range(2)]
        assert got == ref, f"{name}, num_beams={beams}: {got} vs oracle {ref}"
    eng.close()
    spec = wide_spec(name, max_new=8)
    eng = OmniEngine(spec, dtype="bf16").load_state_dict(sd)
    orc = O.OmniOracle(sd, spec.oracle_cfg())
    video, audio, Tv, Ts = clips(1, 7, lo=12)
    ids, lens = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 2, beams=1, max_new=spec.max_dec_tokens, force_len=True)
    got = ids[0, : int(lens[0])].cpu().tolist()
    a_out, na = eng.encode_audio(audio, Ts, 4)
    v_out, nv = eng.encode_video(video, Tv, 2)
    emb, S = eng.build_prompt("audiovisual", 1, na, eng.project("audio", 4, a_out), nv, eng.project("video", 2, v_out))
    logits = eng.prefill("audiovisual", emb, S)
    with torch.no_grad(), O.bf16_emulation():
        ref_emb = orc.prepare_inputs(dict(video=video, audio=audio, lengths=torch.tensor(Ts), tokens=tokens_for(spec, 1)), "audiovisual", 4, 2)
        path = orc.llm.score_path(ref_emb, got, "audiovisual")
    el = (logits[0].cpu() - path[0]).abs().max().item()
    gaps = [float(path[t].max() - path[t][got[t]]) for t in range(len(got))]
    print(f"{name} bf16 vs emulation: prefill logits max err {el:.3e}, worst decoded-token gap {max(gaps):.3e}")
    assert el <= LOGIT_TOL and max(gaps) <= LOGIT_TOL
    # the same utterance inside a batch of 9 (two 16-row blocks of the decode-shaped kernels; another prefill tile count) decodes the same ids
    v9, a9, tv9, ts9 = clips(9, 70, lo=5)
    v9[0], a9[0], tv9[0], ts9[0] = video[0], audio[0], Tv[0], Ts[0]
    ids9, lens9 = eng.generate("audiovisual", v9, tv9, a9, ts9, 4, 2, beams=1, max_new=spec.max_dec_tokens, force_len=True)
    assert ids9[0, : int(lens9[0])].cpu().tolist() == got, f"{name}: an utterance decodes differently inside a batch of 9"
    eng.close()
