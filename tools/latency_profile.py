#!/usr/bin/env python3
"""ONE 16 s utterance per call, one call at a time (the reference's evaluation protocol, data_module.py:238-242), greedy: for
`rocprofv3 --kernel-trace --stats` -- where do the ~52 ms of a call go?   usage: latency_profile.py [calls=6] [beams=1]"""
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

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
beams = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else "stream"      # stream: a stream of the caller's; default: torch's default stream; sched: like bench.py's leg (scheduler + caller streams exist)
spec = full_spec()
eng = OmniEngine(spec, dtype="bf16").load_state_dict(synth_state_dict(spec, seed=1234, device="cuda:0", dtype=torch.bfloat16))
eng.launch_objective(True)
Tv = 400
g = torch.Generator(device="cuda:0").manual_seed(1)
video = torch.randn(Tv, 88, 88, generator=g, device="cuda:0")
audio = torch.nn.functional.layer_norm(torch.randn(1, 640 * Tv, generator=g, device="cuda:0"), (640 * Tv,))
st = torch.cuda.Stream()
if mode == "sched":
    eng.reserve_scheduler("audiovisual", Tv, 640 * Tv, 4, 4, 32)
    streams = [eng.caller_stream(i) for i in range(4)]
    v8 = video.repeat(8, 1, 1)
    a8 = audio.repeat(8, 1)
    pend = []
    for i in range(8):
        with torch.cuda.stream(streams[i % 4]):
            pend.append(eng.submit("audiovisual", v8, [Tv] * 8, a8, [640 * Tv] * 8, 4, 4, max_new=32, force_len=True, video_is_flat=True, slot=i % 4))
    eng.flush()
    torch.cuda.synchronize()
ts = []
for k in range(calls + 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode in ("stream", "stream6"):
        with torch.cuda.stream(st):
            eng.generate("audiovisual", video, [Tv], audio, [640 * Tv], 4, 4, beams=beams, max_new=32, force_len=(beams == 1), video_is_flat=True, slot=6 if mode == "stream6" else 0)
    elif mode == "default0":
        eng.generate("audiovisual", video, [Tv], audio, [640 * Tv], 4, 4, beams=beams, max_new=32, force_len=(beams == 1), video_is_flat=True)
    else:
        eng.generate("audiovisual", video, [Tv], audio, [640 * Tv], 4, 4, beams=beams, max_new=32, force_len=(beams == 1), video_is_flat=True, slot=6)
    torch.cuda.synchronize()
    if k >= 3:
        ts.append((time.perf_counter() - t0) * 1e3)
print(f"[{mode}] B = 1, {beams} beam(s), 16 s clip: median {sorted(ts)[len(ts) // 2]:.2f} ms per call over {calls} calls", flush=True)
