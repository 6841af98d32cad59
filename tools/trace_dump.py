#!/usr/bin/env python3
"""Scheduler event trace of a short bench-like loop, dumped as text: one line per call phase / burst of steps (tools for reading the
overlap by eye).  usage: trace_dump.py [preset] [calls]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as BN
from omni_avsr_amd.modeling import OmniEngine
from omni_avsr_amd.synth import synth_state_dict


class A:
    llm = None; max_new = 32; beams = 1; dtype = "bf16"


preset = sys.argv[1] if len(sys.argv) > 1 else "c1"
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pre = BN.PRESETS[preset]
spec = BN.build_spec(A, pre)
dev = torch.device("cuda:0")
eng = OmniEngine(spec, dtype="bf16").load_state_dict(synth_state_dict(spec, seed=1234, device=dev))
B, Tv = pre["batch"], 400
task = pre["task"]
use_a = task != "video"
sets = [BN.make_inputs([Tv] * B, 100 + s, dev) for s in range(4)]
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
ra, rv = pre["rates"][0]


def call(i):
    video, audio = sets[i % 4]
    with torch.cuda.stream(streams[i % 3]):
        return eng.submit(task, video, [Tv] * B, audio if use_a else None, [640 * Tv] * B if use_a else None, ra, rv, max_new=32, force_len=True, video_is_flat=True, slot=i % 3)


for i in range(12):
    call(i)
eng.flush(); torch.cuda.synchronize()
eng.scheduler_trace(True)
for i in range(ncalls):
    call(12 + i)
eng.flush(); torch.cuda.synchronize()
recs = eng.scheduler_trace_read()
for r in sorted(recs, key=lambda r: r["t0"]):
    what = f"call {r['id']:4d} phase  B={r['rows']}" if r["kind"] == 0 else f"lane {r['id']} burst {r['steps']:2d} steps x {r['rows']:2d} rows"
    print(f"{r['t0']:9.2f} -> {r['t1']:9.2f}  ({r['t1'] - r['t0']:7.2f} ms)  {what}")
