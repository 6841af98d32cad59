#!/usr/bin/env python3
"""harness.evaluate on the headline model with HOST inputs (collate + host-to-device copies inside the timed loop: the PCIe-inclusive rate
of the caller-side loop), sequential vs batches in flight, float samples (host transform done beforehand, as the reference's dataset
does it in its workers) vs the decoded uint8 frames (transform inside the stem kernel).  Synthetic 16 s clips, random-init weights,
greedy, 32 tokens (EOS is not suppressed here, so hypotheses may be shorter)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as BN
from omni_avsr_amd import harness as H
from omni_avsr_amd.modeling import AVSR_LLMs, OmniEngine
from omni_avsr_amd.synth import synth_state_dict


class A:
    llm = None; max_new = 32; beams = 1; dtype = "bf16"


pre = BN.PRESETS["c1"]
spec = BN.build_spec(A, pre)
eng = OmniEngine(spec, dtype="bf16").load_state_dict(synth_state_dict(spec, seed=1234, device="cuda:0"))
m = AVSR_LLMs.__new__(AVSR_LLMs)
m.modality, m.max_dec_tokens, m.num_beams = "audiovisual", 32, 1
m.is_matryoshka, m.is_task_specific, m.spec, m.engine = True, True, spec, eng
m.downsample_ratio_audio, m.downsample_ratio_video = spec.downsample_ratio_audio, spec.downsample_ratio_video
n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 320
g = torch.Generator().manual_seed(0)
n_distinct = min(n_utt, 40)                # distinct clips; the sample list cycles through them (host memory), every batch is still collated and uploaded
base, base_u8 = [], []
for i in range(n_distinct):
    clip = torch.randint(0, 256, (400, 3, 96, 96), generator=g, dtype=torch.uint8)          # decoded mouth ROI, RGB
    wave = torch.randn(256000, 1, generator=g) * 0.1
    base.append(H.make_sample(clip, wave, "audiovisual"))
    base_u8.append(H.make_sample(clip, wave, "audiovisual", device_transform=True))
samples = [(base[i % n_distinct], "gold") for i in range(n_utt)]
samples_u8 = [(base_u8[i % n_distinct], "gold") for i in range(n_utt)]
to_text = lambda ids: [" ".join(str(int(t)) for t in row) for row in ids.cpu()]      # noqa: E731
out = {}
for name, smp, k in (("float samples, sequential", samples[:48], 1), ("float samples, 3 in flight (cold)", samples, 3), ("float samples, 3 in flight", samples, 3),
                     ("uint8 frames, 3 in flight (cold)", samples_u8, 3), ("uint8 frames, 3 in flight", samples_u8, 3), ("uint8 frames, 3 in flight", samples_u8, 3),
                     ("float samples, 3 in flight", samples, 3), ("float samples, 4 in flight", samples, 4), ("uint8 frames, 4 in flight", samples_u8, 4)) + tuple(
                         (f"uint8 frames, {int(k)} in flight", samples_u8, int(k)) for k in os.environ.get("HT_EXTRA_IN_FLIGHT", "").split(",") if k):
    torch.cuda.synchronize()
    s0 = eng.scheduler_stats()
    t0 = time.perf_counter()
    meter = H.evaluate(m, smp, "audiovisual", to_text, batch_size=8, bos_id=spec.ids["bos"], in_flight=k, test_ratio_matry_audio=4, test_ratio_matry_video=4)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = eng.scheduler_stats()
    hyp = sorted(h for _, h, _, _ in meter.per_utt)[: 40]
    rows = (s1["row_steps"] - s0["row_steps"]) / max(1, s1["steps"] - s0["steps"])
    tm = eng.__dict__.pop("_eval_timing", None)
    if tm and tm["batches"]:
        print("    host ms per batch: " + ", ".join(f"{k} {1e3 * v / tm['batches']:.1f}" for k, v in tm.items() if k != "batches"), flush=True)
    print(f"{name:34s}: {len(smp) / dt:7.1f} utt/s  ({dt / (len(smp) / 8) * 1e3:.1f} ms per batch of 8, host inputs; {rows:.1f} rows per decode step)", flush=True)
