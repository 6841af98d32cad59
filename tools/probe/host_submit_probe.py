#!/usr/bin/env python3
"""Host time of ONE oavsr_generate_submit on an idle GPU (headline workload, resident inputs): is the caller thread's ~27 ms per call in the
pipelined bench launch work, or waiting for queue space behind the calls in flight?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench as BN
from omni_avsr_amd.modeling import OmniEngine
from omni_avsr_amd.synth import synth_state_dict


class A:
    llm = None; max_new = 32; beams = 1; dtype = "bf16"


pre = BN.PRESETS["c1"]
spec = BN.build_spec(A, pre)
dev = torch.device("cuda:0")
eng = OmniEngine(spec, dtype="bf16").load_state_dict(synth_state_dict(spec, seed=1234, device=dev))
Tvs = [400] * 8
video, audio = BN.make_inputs(Tvs, 100, dev)
Ts = [640 * t for t in Tvs]
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]


def submit(i):
    with torch.cuda.stream(streams[i % 3]):
        return eng.submit("audiovisual", video, Tvs, audio, Ts, 4, 4, max_new=32, force_len=True, video_is_flat=True, slot=i % 3)


for i in range(6):
    submit(i)
eng.flush(); torch.cuda.synchronize()
for i in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    submit(i)
    t1 = time.perf_counter()
    eng.flush()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"idle GPU: submit host {1e3 * (t1 - t0):6.2f} ms, flush host {1e3 * (t2 - t1):6.2f} ms, until done {1e3 * (t3 - t0):6.2f} ms", flush=True)
# back to back: 24 calls without waiting
torch.cuda.synchronize()
t0 = time.perf_counter()
hs = []
for i in range(24):
    th = time.perf_counter()
    submit(i)
    hs.append(1e3 * (time.perf_counter() - th))
eng.flush(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("back to back: host ms per submit " + " ".join(f"{h:.1f}" for h in hs), flush=True)
print(f"24 calls in {1e3 * dt:.1f} ms = {1e3 * dt / 24:.2f} ms per call", flush=True)
