#!/usr/bin/env python3
"""What ONE decode step costs inside a real call (hipGraph replay on a stream, nothing else on the chip), by row count: the wall time of
`OmniEngine.generate` with max_new = 32 minus the same call with max_new = 4, over 28 steps.  Full-size Llama-3.2-1B AVSR model, 4 s clips
(the encoders' share cancels out).   usage: step_time.py [beams]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from test_fullsize_gpu import full_spec  # noqa: E402
from omni_avsr_amd.modeling import OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402

beams = int(sys.argv[1]) if len(sys.argv) > 1 else 1
spec = full_spec()
eng = OmniEngine(spec, dtype="bf16").load_state_dict(synth_state_dict(spec, seed=1234, device="cuda:0", dtype=torch.bfloat16))
st = torch.cuda.Stream()
Tv = 100
for B in (1, 2, 8, 16, 32, 64, 128):
    if B * beams > 512:
        continue
    g = torch.Generator(device="cuda:0").manual_seed(B)
    video = torch.randn(B * Tv, 88, 88, generator=g, device="cuda:0")
    audio = torch.nn.functional.layer_norm(torch.randn(B, 640 * Tv, generator=g, device="cuda:0"), (640 * Tv,))
    res = {}
    for mn in (32, 4):
        ts = []
        for k in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(st):
                eng.generate("audiovisual", video, [Tv] * B, audio, [640 * Tv] * B, 4, 4, beams=beams, max_new=mn, force_len=(beams == 1), video_is_flat=True,
                             slot=1 + (mn == 4))
            torch.cuda.synchronize()
            if k >= 2:
                ts.append(time.perf_counter() - t0)
        res[mn] = sorted(ts)[len(ts) // 2]
    if beams == 1:
        print(f"B = {B:3d} rows: call with 32 tokens {res[32] * 1e3:7.2f} ms, with 4 tokens {res[4] * 1e3:7.2f} ms -> {(res[32] - res[4]) / 28 * 1e3:6.3f} ms per decode step", flush=True)
    else:
        print(f"B = {B:3d} x {beams} beams = {B * beams:4d} rows: 32-step call {res[32] * 1e3:7.2f} ms (beam search needs max_new == the workspace's: no 4-token leg)", flush=True)
