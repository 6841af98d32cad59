"""-m gpu: the bf16 engine against the REFERENCE'S OWN bf16 arithmetic.  tests/golden/ref_bf16.npz holds what the reference's classes
produce when they are executed in torch.bfloat16 the way `Trainer(precision='bf16-true')` runs them (eval_OmniAVSR.py:18; generator:
oracle/gen_golden_bf16.py, build container) on seeded tiny weights: compressed audio / video features, `inputs_embeds`, prefill logits,
the greedy token and the logits of one cached decode step.

Both bf16 computations are measured against the same fp32 truth (the oracle, on the same bf16-rounded inputs).  Bar per stage:
    err(engine bf16, fp32 oracle) <= 1.5 x err(reference bf16, fp32 oracle)      (mean and max absolute error)
i.e. the engine's rounding points (fp32 accumulation, fused epilogues) may differ from the reference's (bf16 after every op), but it
must not be further from exact arithmetic than the reference's own evaluation is."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd.config import tiny_spec  # noqa: E402
from omni_avsr_amd.modeling import OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_bf16.npz")
CASES = {      # must match oracle/gen_golden_bf16.py
    "llama": dict(family="llama", task_specific=True, shared=True, matryoshka=True),
    "qwen": dict(family="qwen2", head_dim=128, matryoshka=True, single_projector=True),
}
SLACK = 1.5


def errs(x, truth):
    e = (torch.as_tensor(x).float().cpu() - truth.float()).abs()
    return e.mean().item(), e.max().item()


@pytest.mark.parametrize("name", sorted(CASES))
def test_engine_bf16_is_as_close_to_fp32_as_the_references_bf16(name):
    z = np.load(GOLD)
    spec = tiny_spec(**CASES[name])
    sd = synth_state_dict(spec, seed=int(z["seed"]))
    Tv, ra, rv = int(z[f"{name}_Tv"]), int(z[f"{name}_ra"]), int(z[f"{name}_rv"])
    g = torch.Generator().manual_seed(int(z["input_seed"]))
    video = torch.randn(1, Tv, 1, 88, 88, generator=g).bfloat16().float()           # what HalfPrecision.convert_input hands to forward()
    audio = torch.randn(1, 640 * Tv, 1, generator=g).bfloat16().float()
    tokens = torch.full((1, 1), spec.ids["bos"]) if spec.llm.family == "llama" else torch.zeros(1, 0, dtype=torch.long)
    inp = dict(video=video, audio=audio, lengths=torch.tensor([640 * Tv]), tokens=tokens)
    orc = O.OmniOracle(sd, spec.oracle_cfg())
    mod = "audiovisual" if spec.is_task_specific else None
    with torch.no_grad():
        a32 = orc.encode_audio(audio, 640 * Tv, ra, mel_bf16=True)[0]
        v32 = orc.encode_video(video, rv)[0]
        e32 = orc.prepare_inputs(inp, "audiovisual", ra, rv, mel_bf16=True)
        oc = [None] * spec.llm.layers
        lg32 = orc.llm.forward(e32, 0, oc, mod)[0, -1]
        tok = int(z[f"{name}_token"])
        lg32_1 = orc.llm.forward(orc.llm.embed(torch.tensor([[tok]])), e32.shape[1], oc, mod)[0, -1]
    eng = OmniEngine(spec, dtype="bf16").load_state_dict(sd)
    eng.ensure_workspace(1, Tv, 640 * Tv)
    a_out, na = eng.encode_audio(audio, [640 * Tv], ra)
    v_out, nv = eng.encode_video(video, [Tv], rv)
    emb, S = eng.build_prompt("audiovisual", 1, na, eng.project("audio", ra, a_out), nv, eng.project("video", rv, v_out))
    logits = eng.prefill("audiovisual", emb, S)[0]
    ids, lens = eng.decode("audiovisual", 1, 1, 2, force_len=True)
    torch.cuda.synchronize()
    report = []
    for stage, mine, ref16, truth in (("audio features", a_out, z[f"{name}_audio_feats"], a32), ("video features", v_out, z[f"{name}_video_feats"], v32),
                                      ("inputs_embeds", emb, z[f"{name}_embeds"], e32[0]), ("prefill logits", logits, z[f"{name}_logits"], lg32)):
        (m_e, x_e), (m_r, x_r) = errs(mine, truth), errs(ref16, truth)
        report.append(f"{stage}: engine mean {m_e:.2e} max {x_e:.2e} | reference bf16 mean {m_r:.2e} max {x_r:.2e}")
        assert m_r > 0 and m_e <= SLACK * m_r and x_e <= SLACK * x_r, f"{name}: {report[-1]}"
    print("\n".join(report))
    # greedy tokens: the reference's bf16 run, the fp32 oracle and the engine agree wherever the fp32 margin exceeds both runs' logit error
    got = ids[0].cpu().tolist()
    for step, (lg, ref_lg) in enumerate(((lg32, z[f"{name}_logits"]), (lg32_1, z[f"{name}_step_logits"]))):
        top2 = torch.topk(lg, 2).values
        margin = float(top2[0] - top2[1])
        if margin > 4 * errs(ref_lg, lg)[1]:
            assert got[step] == int(lg.argmax()) == int(torch.as_tensor(ref_lg).argmax()), f"{name}: greedy token {step}: engine {got[step]}, fp32 {int(lg.argmax())}"
        if step == 0 and got[0] != tok:
            break                                              # the stored step logits belong to the reference's first token
