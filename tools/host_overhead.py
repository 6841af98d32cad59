#!/usr/bin/env python3
"""How long does the host spend issuing one full generate() (no sync), with an empty queue and in the three-streams steady state?
Shows whether the pipeline is launch-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
eng = OmniEngine(spec, dtype="bf16", device="cuda:0").load_state_dict(synth_state_dict(spec, seed=1234, device=dev))
Tvs, Ts = [400] * 8, [256000] * 8
sets = [BN.make_inputs(Tvs, 100 + s, dev) for s in range(4)]
streams = [torch.cuda.Stream() for _ in range(3)]
eng.serialize_encoders(True)


def call(i):
    v, a = sets[i % 4]
    with torch.cuda.stream(streams[i % 3]):
        return eng.generate("audiovisual", v, Tvs, a, Ts, 4, 4, max_new=32, force_len=True, video_is_flat=True, slot=i % 3)


for i in range(6):
    call(i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); call(rep); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"empty queue: host issue {1e3 * (t1 - t0):.2f} ms, until GPU done {1e3 * (t2 - t0):.1f} ms")
n = 48
torch.cuda.synchronize()
t0 = time.perf_counter()
host = []
for i in range(n):
    h0 = time.perf_counter(); call(i); host.append(time.perf_counter() - h0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
host.sort()
print(f"steady state, {n} calls: total {1e3 * (t2 - t0) / n:.2f} ms per call; host inside generate(): median {1e3 * host[n // 2]:.2f} ms, "
      f"min {1e3 * host[0]:.2f}, max {1e3 * host[-1]:.2f}; host loop finished after {1e3 * (t1 - t0):.0f} of {1e3 * (t2 - t0):.0f} ms")
