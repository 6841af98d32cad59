#!/usr/bin/env python3
"""How long does the host spend issuing one full generate() (no sync)?  Shows whether the pipeline is launch-bound."""
import os, sys, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
args = argparse.Namespace(max_new=32)
spec = bench.build_spec(args)
from omni_avsr_amd.modeling import OmniEngine
from omni_avsr_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
sd = synth_state_dict(spec, seed=1234, device=dev)
eng = OmniEngine(spec, dtype="bf16", device="cuda:0").load_state_dict(sd)
del sd
video, audio = bench.make_inputs(8, 400, 100, dev)
Tvs, Ts = [400] * 8, [256000] * 8
for _ in range(2):
    eng.generate("audiovisual", video, Tvs, audio, Ts, 4, 4, max_new=32, force_len=True, video_is_flat=True)
torch.cuda.synchronize()
side = torch.cuda.Stream()
for name, stream in (("default stream (eager decode)", torch.cuda.current_stream()), ("side stream (hipGraph decode)", side)):
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            eng.generate("audiovisual", video, Tvs, audio, Ts, 4, 4, max_new=32, force_len=True, video_is_flat=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: host issue {1e3 * (t1 - t0):.1f} ms, until GPU done {1e3 * (t2 - t0):.1f} ms")
