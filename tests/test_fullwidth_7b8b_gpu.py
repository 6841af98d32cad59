"""-m gpu: the LLM shapes of BASELINE.json configs[3] / configs[4] at FULL WIDTH (two decoder layers, tiny encoders so the CPU oracle stays
affordable): Qwen2.5-7B -- hidden 3584, 28 query / 4 kv heads x 128, MLP 18 944, vocabulary 152 064 (+4 tags), qkv bias, untied lm_head,
Omni-LoRA task-specific + shared with inner width round(3584 / 32) = 112 (Qwen_LoRA.py:464-475,557-570) -- and Llama-3.1-8B -- hidden 4096,
32 / 8 heads x 128, MLP 14 336, vocabulary 128 256 (+5), llama3 RoPE scaling (Llama_LoRA.py:250-259, lightning_OmniAVSR.py:28-37).
What runs at these widths and nowhere else in the suite: prefill GEMMs with K = 3584 / 4096 and N = 2 x 18 944, the LoRA K segment
K2 = 448 / 512, flash attention with 28 q / 4 kv heads, the decode GEMVs and the two-stage arg-max over 152k columns, beam search over
152k-column rows.

  * f32: greedy AND HF-beam-search (K = 15) token ids bit-exact vs the CPU oracle;
  * bf16: prefill logits / decoded tokens within the bf16-emulation bound of tests/test_bf16_emulation_gpu.py;
  * bf16: three calls of 8 through the decode scheduler (24 rows per step) decode the ids of isolated calls."""
import dataclasses
import functools

import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd.config import LLAMA_IDS, LLM_SPECS, QWEN_IDS, tiny_spec  # noqa: E402
from omni_avsr_amd.modeling import OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

LOGIT_TOL = 0.08
MODELS = {"qwen25_7b": "Qwen/Qwen2.5-7B", "llama31_8b": "meta-llama/Meta-Llama-3.1-8B"}


def wide_spec(name, max_new=6):
    llm = dataclasses.replace(LLM_SPECS[MODELS[name]], layers=2)
    spec = tiny_spec(family=llm.family, head_dim=llm.head_dim, task_specific=True, shared=True)
    spec.llm, spec.ids = llm, dict(QWEN_IDS if llm.family == "qwen2" else LLAMA_IDS)
    spec.lora_rank, spec.lora_alpha, spec.max_dec_tokens = 32, 4, max_new
    return spec


@functools.lru_cache(maxsize=2)
def weights(name):
    """1.6 B parameters per model (2 layers + embedding + untied lm_head): drawn once per session"""
    return synth_state_dict(wide_spec(name), seed=1234)


def clips(B, seed, lo=5):
    g = torch.Generator().manual_seed(seed)
    Tv = [int(x) for x in torch.randint(lo, 13, (B,), generator=g)]
    Ts = [640 * t for t in Tv]
    video = torch.randn(B, 12, 1, 88, 88, generator=g)
    audio = torch.randn(B, 7680, 1, generator=g)
    for b in range(B):
        video[b, Tv[b]:] = 0
        audio[b, Ts[b]:] = 0
    return video, audio, Tv, Ts


def tokens_for(spec, B):
    return torch.full((B, 1), spec.ids["bos"]) if spec.llm.family == "llama" else torch.zeros(B, 0, dtype=torch.long)


@pytest.mark.parametrize("name", sorted(MODELS))
def test_f32_greedy_and_beam15_ids_equal_the_oracle(name):
    spec = wide_spec(name)
    assert spec.lora_inner() == (112 if name == "qwen25_7b" else 128)
    sd = weights(name)
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


@pytest.mark.parametrize("name", sorted(MODELS))
def test_bf16_within_the_emulation_bound_and_scheduler_invariance(name):
    spec = wide_spec(name, max_new=8)
    sd = weights(name)
    eng = OmniEngine(spec, dtype="bf16").load_state_dict(sd)
    orc = O.OmniOracle(sd, spec.oracle_cfg())
    # (a) one utterance: prefill logits and the decoded path against the bf16-emulating oracle
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
    # (b) three calls of 8 share the scheduler's steps (24 rows per step at this width)
    calls = [clips(8, 40 + i) for i in range(3)]
    ref = []
    for v, a, tv, ts in calls:
        i_, l_ = eng.generate("audiovisual", v, tv, a, ts, 4, 2, beams=1, max_new=spec.max_dec_tokens, force_len=True)
        ref.append(i_.cpu())
    eng.sched_lanes = 1
    streams = [torch.cuda.Stream() for _ in range(3)]
    pend = []
    for i, (v, a, tv, ts) in enumerate(calls):
        with torch.cuda.stream(streams[i]):
            pend.append(eng.submit("audiovisual", v, tv, a, ts, 4, 2, max_new=spec.max_dec_tokens, force_len=True, slot=i))
    eng.flush()
    torch.cuda.synchronize()
    assert eng.scheduler_stats()["rows_per_step"] > 16
    for i, (ids_, lens_, _) in enumerate(pend):
        assert torch.equal(ids_.cpu(), ref[i]), f"{name}: call {i} decodes differently inside the shared decode loop"
