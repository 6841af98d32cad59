#!/usr/bin/env python3
"""B = 1, one 16 s clip, one call at a time (latency objective), on a stream: device-synchronised wall time of the stages alone and of generate()
with 1 and 32 tokens -- where a call's milliseconds go without a profiler attached."""
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
video5 = video.view(1, Tv, 1, 88, 88)
audio = torch.nn.functional.layer_norm(torch.randn(1, 640 * Tv, generator=g, device="cuda:0"), (640 * Tv,))
st = torch.cuda.Stream()


def timed(fn, n=9):
    ts = []
    for k in range(n + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            fn()
        torch.cuda.synchronize()
        if k >= 3:
            ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


gen = lambda mn: eng.generate("audiovisual", video, [Tv], audio, [640 * Tv], 4, 4, beams=1, max_new=mn, force_len=True, video_is_flat=True)
res = {
    "encode_video alone": timed(lambda: eng.encode_video(video5, [Tv], 4)),
    "encode_audio alone": timed(lambda: eng.encode_audio(audio.view(1, -1, 1), [640 * Tv], 4)),
    "generate, 1 token (encoders + prefill + first arg-max)": timed(lambda: gen(1)),
    "generate, 2 tokens": timed(lambda: gen(2)),
    "generate, 32 tokens": timed(lambda: gen(32)),
}
os.environ["OAVSR_ENC_SIDE_BY_SIDE"] = "0"
res["generate, 1 token, encoders one after the other"] = timed(lambda: gen(1))
for k, v in res.items():
    print(f"{k:60s} {v:7.2f} ms")
print(f"per decode step: {(res['generate, 32 tokens'] - res['generate, 2 tokens']) / 30:.3f} ms")
