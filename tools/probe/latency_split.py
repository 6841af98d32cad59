#!/usr/bin/env python3
"""B = 1 call on the default stream vs on a stream of the caller's: host time until generate() returns, until the device is idle, and the device time between two events on the stream the library call runs on"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_fullsize_gpu import full_spec
from omni_avsr_amd.modeling import OmniEngine
from omni_avsr_amd.synth import synth_state_dict
spec = full_spec()
eng = OmniEngine(spec, dtype="bf16").load_state_dict(synth_state_dict(spec, seed=1234, device="cuda:0", dtype=torch.bfloat16))
eng.launch_objective(True)
Tv = 400
g = torch.Generator(device="cuda:0").manual_seed(1)
video = torch.randn(Tv, 88, 88, generator=g, device="cuda:0")
audio = torch.nn.functional.layer_norm(torch.randn(1, 640 * Tv, generator=g, device="cuda:0"), (640 * Tv,))
st = torch.cuda.Stream()
for mode in ("stream", "default", "stream", "default"):
    rows = []
    for k in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "stream":
            with torch.cuda.stream(st):
                eng.generate("audiovisual", video, [Tv], audio, [640 * Tv], 4, 4, beams=1, max_new=32, force_len=True, video_is_flat=True)
        else:
            eng.generate("audiovisual", video, [Tv], audio, [640 * Tv], 4, 4, beams=1, max_new=32, force_len=True, video_is_flat=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if k >= 3:
            rows.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    rows.sort(key=lambda r: r[1])
    r = rows[len(rows) // 2]
    print(f"[{mode}] generate() returns after {r[0]:.2f} ms, device idle after {r[1]:.2f} ms", flush=True)
