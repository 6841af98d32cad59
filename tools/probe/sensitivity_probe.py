#!/usr/bin/env python3
"""Which seeded-weight recipe makes the greedy ids of the full-size model depend on the clip?  (tests/test_fullsize_gpu.py:
test_input_sensitive_weights_under_the_scheduler needs one.)  Prints, per recipe, how many distinct 32-token hypotheses 16 utterances give."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from test_fullsize_gpu import full_spec  # noqa: E402
from omni_avsr_amd.modeling import OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402

RECIPES = {
    "embed x0.05, o_proj x8": dict(embed=0.05, o=8.0),
    "embed x0.05, o_proj x8, q_proj x6": dict(embed=0.05, o=8.0, q=6.0),
    "embed x0.02, o_proj x16, q_proj x6, v_proj x4": dict(embed=0.02, o=16.0, q=6.0, v=4.0),
}
spec = full_spec()
B, Tv = 8, 400
sets = []
for s_ in range(2):
    g = torch.Generator(device="cuda:0").manual_seed(300 + s_)
    video = torch.randn(B * Tv, 88, 88, generator=g, device="cuda:0")
    audio = torch.nn.functional.layer_norm(torch.randn(B, 640 * Tv, generator=g, device="cuda:0"), (640 * Tv,))
    sets.append((video, audio))
for name, r in RECIPES.items():
    sd = synth_state_dict(spec, seed=4321, device="cuda:0", dtype=torch.bfloat16)
    for k in list(sd):
        if not k.startswith("llm.") and "prompt" not in k:
            continue
        if k.endswith("embed_tokens.weight") or "prompt" in k:
            sd[k] = sd[k] * r.get("embed", 1.0)
        elif k.endswith("self_attn.o_proj.weight"):
            sd[k] = sd[k] * r.get("o", 1.0)
        elif k.endswith("self_attn.q_proj.weight"):
            sd[k] = sd[k] * r.get("q", 1.0)
        elif k.endswith("self_attn.v_proj.weight"):
            sd[k] = sd[k] * r.get("v", 1.0)
    eng = OmniEngine(spec, dtype="bf16").load_state_dict(sd)
    del sd
    rows = []
    for video, audio in sets:
        ids, lens = eng.generate("audiovisual", video, [Tv] * B, audio, [640 * Tv] * B, 4, 4, beams=1, max_new=32, force_len=True, video_is_flat=True)
        torch.cuda.synchronize()
        rows += ids.cpu().tolist()
    distinct = len({tuple(x) for x in rows})
    first = len({x[0] for x in rows})
    print(f"{name}: {distinct} distinct hypotheses / {len(rows)} utterances; {first} distinct first tokens; sample {rows[0][:6]} {rows[9][:6]}", flush=True)
    eng.close()
    del eng
    torch.cuda.empty_cache()
