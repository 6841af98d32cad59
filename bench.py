#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): utterances/s of the Omni-AVSR inference hot path on synthetic 16 s clips.

  python bench.py [--gpus N --steps K --warmup W]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU; it starts N ranks as a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...` (one rank per GPU over RCCL), relays rank 0's JSON line and
exits with the children's return code (non-zero when fewer than N devices exist).  Under torch.distributed.run (the driver's
form) every rank reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

A "step" is `--calls-per-step` (default 8) consecutive passes of the whole hot path (`AVSR_LLMs.forward(is_trainval=False)`:
log-mel + Whisper encoder, ResNet + AV-HuBERT encoder, compressor + projectors, prompt splice, LLM prefill, 32 greedy decode
steps with EOS suppressed so every run does the same work), each over its own batch of B utterances per GPU, rotating over
`--input-sets` distinct input batches (so no batch's 107 MB of inputs stays resident in the 256 MB Infinity Cache).  Inputs
are in HBM before the timed region.  Workload (default preset c1) = BASELINE.json configs[1]: AVSR, AV-HuBERT-Large +
Whisper-medium.en + Llama-3.2-1B (Omni-LoRA task-specific + shared, rank 32 / alpha 4, LoRA on the video encoder), batch
8 x 16 s per model call, compression rates (4, 4).  Presets c2 / c3 / c4 = configs[2..4] at their per-GPU batch (not the
headline).  Weights are seeded random tensors of the real architecture (no checkpoints offline).  dtype: bf16 storage /
fp32 accumulate (the reference evaluates in bf16-true, eval_OmniAVSR.py:19).

Every call is one batch of B utterances, submitted on one of `--streams` caller streams (`oavsr_generate_submit`: encoders +
prefill there); the ENGINE's decode scheduler serves the decode rows of the calls in flight from its own step loops (up to 128 rows
per weight pass, `--sched-*`), nothing is grouped on the caller's side.  The timed region ends with the scheduler's flush and a device
synchronisation: every hypothesis of every timed call is complete inside it.  `--no-scheduler`: one decode loop per call (round 2).

Besides the contract's fields the line carries `roofline` (dominant MFMA-side kernel family, HIP events, one call at a time),
`roofline_sides` (MFMA side: flops / summed kernel time; decode side: the real step loop replayed alone), `schedule` (event trace
of the scheduler: what overlaps what), `cpu_baseline`, `per_call_decode` (the same job with the scheduler off) and `stages`.

Multi-GPU: the path shards by utterance with no data-path collective: every rank runs its own batches ("weak"
scaling); the only exchange is one all_gather of the last call's hypotheses (omni_avsr_amd.dist.gather_hypotheses), inside the
timed region before the closing barrier.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402   (importing torch / counting devices does not initialise the GPU)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

# BASELINE.json configs[i] -> what a rank runs.  batch = utterances per model call on ONE GPU (configs[3]: 32 over 4 GPUs,
# configs[4]: 64 over 8 GPUs); rates = (audio, video) compression rates cycled call by call.
PRESETS = {
    "c1": dict(idx=1, llm="meta-llama/Llama-3.2-1B", task="audiovisual", batch=8, rates=[(4, 4)], ra=[4, 16], rv=[2, 4, 5], varlen=False),
    "c2": dict(idx=2, llm="meta-llama/Llama-3.2-1B", task="video", batch=16, rates=[(4, 4)], ra=[4, 16], rv=[2, 4, 5], varlen=False),
    "c3": dict(idx=3, llm="Qwen/Qwen2.5-7B", task="audiovisual", batch=8, rates=[(2, 2), (4, 4), (8, 8)], ra=[2, 4, 8], rv=[2, 4, 8], varlen=False),
    "c4": dict(idx=4, llm="meta-llama/Meta-Llama-3.1-8B", task="audiovisual", batch=8, rates=[(4, 4)], ra=[4, 16], rv=[2, 4, 5], varlen=True),
}


def build_spec(args, pre):
    from omni_avsr_amd.config import spec_from_names
    return spec_from_names(
        args.llm or pre["llm"], "openai/whisper-medium.en" if pre["task"] != "video" else None,
        "large_vox_iter5" if pre["task"] != "audio" else None, use_lora_avhubert=True,
        intermediate_size=2048, compression_mode="avg-pooling", downsample_ratio_audio=pre["ra"], downsample_ratio_video=pre["rv"],
        is_matryoshka=True, is_single_matry_projector=False, lora_rank=32, lora_alpha=4, is_task_specific=True, shared_lora=True,
        prompt_len={"audio": 8, "video": 8, "audiovisual": 8}, max_dec_tokens=args.max_new, num_beams=1)


def make_inputs(Tvs, seed, device):
    """video N(0,1) [sum T_v,88,88]; audio N(0,1) [B, 640 max T_v], utterance-level layer-norm over the valid
    samples (transforms.py:123-125), zeros behind them."""
    g = torch.Generator(device=device).manual_seed(seed)
    video = torch.randn(sum(Tvs), 88, 88, generator=g, device=device, dtype=torch.float32)
    audio = torch.zeros(len(Tvs), 640 * max(Tvs), device=device, dtype=torch.float32)
    for b, tv in enumerate(Tvs):
        a = torch.randn(640 * tv, generator=g, device=device, dtype=torch.float32)
        audio[b, : 640 * tv] = torch.nn.functional.layer_norm(a, (640 * tv,))
    return video, audio


def host_cpu():
    """CPU model, physical cores (unique (package, core) pairs) and logical CPUs of the host."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
                cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or logical or os.cpu_count()), (logical or os.cpu_count())


def cpu_baseline(spec, sd_cpu, Tv, max_new, task, rates):
    """The oracle (CPU fp32 restatement, kind 'port': the reference has hard .cuda() calls, modeling_OmniAVSR.py:534) timed on this
    host's cores on ONE utterance of the same workload.  Two decodes over the same encoder output: greedy (like for like with the
    GPU line) and the reference's own algorithmic choices (eval_OmniAVSR.py:215-226: 15 beams; HF expands the prompt K-fold for the
    prefill and Llama_LoRA.py:372-373 materialises the logits of every prompt position)."""
    from oracle import torch_oracle as O
    orc = O.OmniOracle(sd_cpu, spec.oracle_cfg())
    g = torch.Generator().manual_seed(7)
    video = torch.randn(1, Tv, 1, 88, 88, generator=g)
    audio = torch.nn.functional.layer_norm(torch.randn(1, 640 * Tv, generator=g), (640 * Tv,)).unsqueeze(-1)
    inp = dict(video=video, audio=audio, lengths=torch.tensor([640 * Tv]), tokens=torch.full((1, 1), max(spec.ids["bos"], 0)))
    mod = task if spec.is_task_specific else None
    ids = spec.ids
    model, phys, logical = host_cpu()
    # Bounded sample (about 30-40 s of CPU work): the encoders + prompt of one utterance in full, then each decode for its prefill and
    # a few steps; the remaining steps cost the same (one token against a cache that grows by < 10 %), so the decode time is
    # prefill + (max_new - 1) x the mean measured step.
    tg, tb = {}, {}
    with torch.no_grad():
        t0 = time.time()
        emb = orc.prepare_inputs(inp, task, rates[0], rates[1])
        t_enc = time.time() - t0
        orc.llm.generate_greedy(emb, min(max_new, 7), ids["eos"], ids["pad"], mod, force_len=True, timing=tg)
        orc.llm.generate_beam(emb, min(max_new, 4), 15, ids["eos"], ids["pad"], mod, reference_prefill=True, timing=tb)
    t_greedy = tg["prefill"] + (max_new - 1) * sum(tg["steps"]) / max(1, len(tg["steps"]))
    t_beam = tb["prefill"] + (max_new - 1) * sum(tb["steps"]) / max(1, len(tb["steps"]))
    dur = Tv / 25.0
    return dict(value=1.0 / (t_enc + t_greedy), unit="utterances/s", cores=phys, threads=torch.get_num_threads(), logical_cpus=logical,
                cpu_model=model, kind="port",
                sample=f"1 utterance of the same workload ({dur:.0f} s clip), torch CPU fp32: encoders + prompt {t_enc:.1f} s measured in full; greedy decode "
                       f"{t_greedy:.1f} s = prefill {tg['prefill']:.1f} s + {max_new - 1} x {sum(tg['steps']) / max(1, len(tg['steps'])):.2f} s/step "
                       f"({len(tg['steps'])} steps measured)",
                reference_default=dict(value=1.0 / (t_enc + t_beam), unit="utterances/s", beams=15, rtf=round((t_enc + t_beam) / dur, 3),
                                       sample=f"same encoder output, the reference's own decode (HF beam search K=15, K-fold prefill, logits of every prompt "
                                              f"position): {t_beam:.1f} s = prefill {tb['prefill']:.1f} s + {max_new - 1} x "
                                              f"{sum(tb['steps']) / max(1, len(tb['steps'])):.2f} s/step ({len(tb['steps'])} steps measured)"),
                rtf=round((t_enc + t_greedy) / dur, 3))


def env_snapshot():
    """Every environment variable that can steer the library or the runtime under it: the OAVSR_* names (oavsr_env_report: a trailing '?' marks
    a name the library itself does not read -- none of its own switches changes a result, they pick between bit-identical kernel forms) and
    the ROCm runtime's knobs that were set for this process."""
    from omni_avsr_amd import lib as L
    env = dict(L.env_report())
    for k, v in os.environ.items():
        if k.startswith(("GPU_MAX_HW_QUEUES", "HSA_", "HIP_", "AMD_", "ROCR_", "NCCL_", "RCCL_")):
            env[k] = v
    return env


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic_*.json: FETCH_SIZE and
    WRITE_SIZE collected in separate runs, gfx950 read-side correction applied; counters cannot be read from inside this process).
    The newest round's file wins."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel") == kernel:
            best = d.get("traffic_bytes_per_launch")
    return best


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--calls-per-step", type=int, default=8, help="model calls (batches of --batch utterances) per step")
    ap.add_argument("--input-sets", type=int, default=4, help="distinct input batches rotated through the timed loop")
    ap.add_argument("--preset", default="c1", choices=sorted(PRESETS), help="c1 = BASELINE.json configs[1] (headline); c2..c4 = configs[2..4]")
    ap.add_argument("--batch", type=int, default=None, help="utterances per model call per GPU (default: the preset's)")
    ap.add_argument("--frames", type=int, default=400, help="video frames per utterance (400 = 16 s)")
    ap.add_argument("--max-new", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-grouped", action="store_true", help="skip the extra leg with the decode scheduler off (N=1, headline preset only; reported as `per_call_decode`)")
    ap.add_argument("--beams", type=int, default=1, help="1 = greedy (north-star decode); 15 = the reference's default HF beam search")
    ap.add_argument("--llm", default=None, help="override the preset's LLM architecture (random-init)")
    ap.add_argument("--decode-group", type=int, default=1,
                    help="G > 1: G consecutive calls share ONE decode loop (encoders / prefill still run per batch of --batch; "
                         "continuous batching of the decode phase).  Opt-in: not the configuration the headline number is quoted on")
    ap.add_argument("--alone", action="store_true",
                    help="dispatcher objective 'one call at a time on an idle chip' (oavsr_set_launch_objective(1)) for the timed loop: with --streams 1 --no-scheduler "
                         "the reference's own protocol at this batch size; never the headline")
    ap.add_argument("--no-scheduler", action="store_true",
                    help="round-2 mode: every call runs its own decode loop on its own stream (default: the engine's decode scheduler merges the "
                         "decode rows of the calls in flight; greedy only)")
    ap.add_argument("--sched-rows", type=int, default=None, help="rows per decode step of the engine's scheduler (<= 128; default: the engine's)")
    ap.add_argument("--sched-slots", type=int, default=None, help="row slots of the scheduler's pool (default: the engine's)")
    ap.add_argument("--sched-stride", type=int, default=0, help="steps per admission (0 = automatic)")
    ap.add_argument("--sched-lanes", type=int, default=None, help="independent step loops of the scheduler (default: the engine's)")
    ap.add_argument("--pending-per-lane", type=int, default=24,
                    help="submitted calls the loop leaves uncollected: 2 x streams + this x lanes (a stream-side wait beyond that).  8 (rounds 3-4) kept the "
                         "lanes' steps at 74-78 rows; 16: 91-107 rows per step, +2.5-3.8 %% utt/s; 24 with the 1024-slot pool: +2 %% more (round 5)")
    ap.add_argument("--no-stagger", action="store_true", help="let the encoder phases of in-flight batches share the chip (default: chained)")
    ap.add_argument("--testset", type=int, default=0,
                    help="N > 0: the test-set-shaped run instead of the step loop (BASELINE.json configs[4] / SURVEY.md 8d C5: N synthetic utterances, "
                         "lengths U[1 s, 16 s], HOST inputs): duration-sorted sharding over the ranks -> batches -> harness.evaluate(in_flight=4) -> one "
                         "all_gather of the hypotheses + all_reduce of the WER counters.  STRONG scaling: value = N / wall time of the whole set")
    ap.add_argument("--streams", type=int, default=None,
                    help="caller streams = batches whose encoder phases (and, without the scheduler, decode loops) may be in flight; default 4; "
                         "1 = strictly sequential calls")
    return ap.parse_args()


def per_call_leg(eng, args, pre, task, use_a, B, sets, streams, nstream):
    """The same job with the decode scheduler off (round-2 mode): every call runs its own decode loop on its own stream, the encoder +
    prefill phases of the calls in flight chained.  Reported next to the headline as `per_call_decode`, never as `value`."""
    eng.serialize_encoders(True)
    nsets = len(sets)

    def call(i):
        video, audio, Tvs, Ts = sets[i % nsets]
        ra, rv = pre["rates"][i % len(pre["rates"])]
        with torch.cuda.stream(streams[i % nstream]):
            return eng.generate(task, video, Tvs, audio if use_a else None, Ts if use_a else None, ra, rv, beams=1, max_new=args.max_new,
                                force_len=True, video_is_flat=True, slot=i % nstream)
    n = 0
    for _ in range(nstream + 1):
        call(n); n += 1
    torch.cuda.synchronize()
    ncall = max(2 * nstream, (args.steps * max(1, args.calls_per_step)) // 4)      # a quarter of the main leg's calls
    t0 = time.perf_counter()
    for _ in range(ncall):
        call(n); n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.serialize_encoders(False)
    return {"value": round(ncall * B / dt, 3), "unit": "utterances/s", "ms_per_call": round(dt / ncall * 1e3, 3), "calls": ncall,
            "note": "decode scheduler off: one decode loop per call of the same batches (the round-2 configuration); not `value`"}


def latency_leg(eng, args, pre, task, use_a, first_set):
    """One utterance alone (B = 1), one call at a time, host-synchronised after every call: ms per utterance, median of 10, for greedy decode
    (32 tokens, EOS suppressed: like `value`) and for HF beam search K = 15 (the reference's default; natural end)."""
    video, audio, Tvs, Ts = first_set
    t1 = Tvs[0]
    v1, a1 = video[: t1].contiguous(), (audio[:1, : 640 * t1].contiguous() if use_a else None)
    ra, rv = pre["rates"][0]
    out = {"unit": "ms per utterance", "clip_s": round(t1 / 25.0, 2), "calls": 10,
           "note": "B = 1, one call at a time on an idle chip, device-synchronised per call, median of 10 (the reference's evaluation protocol, data_module.py:238-242)"}
    prev_obj = eng.launch_objective(True)   # one call at a time on an idle chip: the dispatcher's latency rule (same bits); restored in `finally`
    try:
        _latency_calls(eng, args, out, task, v1, t1, a1, use_a, ra, rv)
    finally:
        eng.launch_objective(prev_obj)
    return out


def _latency_calls(eng, args, out, task, v1, t1, a1, use_a, ra, rv):
    for name, beams in (("greedy", 1), ("beam15", 15)):
        ts = []
        for k in range(13):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.generate(task, v1, [t1], a1, [640 * t1] if use_a else None, ra, rv, beams=beams, max_new=args.max_new, force_len=(beams == 1),
                         video_is_flat=True, slot=6 if beams == 1 else 7)
            torch.cuda.synchronize()
            if k >= 3:
                ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        out[name] = round(ts[len(ts) // 2], 3)
        out[name + "_min"] = round(ts[0], 3)


def schedule_summary(recs, skip_calls):
    """Occupancy summary of a scheduler trace (OmniEngine.scheduler_trace_read): when were encoder + prefill phases running, when were decode
    steps running, and how fast do steps go beside the phases.  Everything is clipped to the steady window that starts with call
    `skip_calls` (the lanes are full from there on) and ends with the last phase."""
    phases = [r for r in recs if r["kind"] == 0 and r["t1"] >= 0]
    bursts = [r for r in recs if r["kind"] == 1 and r["t1"] >= 0]
    if len(phases) <= skip_calls + 2 or not bursts:
        return None
    w0, w1 = phases[skip_calls]["t0"], max(r["t1"] for r in phases)
    clip = lambda iv: [(max(a, w0), min(b, w1)) for a, b in iv if min(b, w1) > max(a, w0)]      # noqa: E731

    def union(iv):
        tot, end = 0.0, None
        for a, b in sorted(iv):
            if end is None or a > end:
                tot += b - a
                end = b
            elif b > end:
                tot += b - end
                end = b
        return tot

    def both(iv1, iv2):                    # measure of (union iv1) intersected with (union iv2), by sweeping the boundaries
        pts = sorted([(a, 1, 0) for a, _ in iv1] + [(b, -1, 0) for _, b in iv1] + [(a, 0, 1) for a, _ in iv2] + [(b, 0, -1) for _, b in iv2])
        c1 = c2 = 0
        last, tot = None, 0.0
        for t, d1, d2 in pts:
            if last is not None and c1 > 0 and c2 > 0:
                tot += t - last
            c1 += d1
            c2 += d2
            last = t
        return tot
    pi, bi = clip([(r["t0"], r["t1"]) for r in phases]), clip([(r["t0"], r["t1"]) for r in bursts])
    win = w1 - w0
    inside = [r for r in bursts if r["t0"] >= w0 and r["t1"] <= w1]
    steps = sum(r["steps"] for r in inside)
    n_ph = len([r for r in phases[skip_calls:]])
    return dict(window_ms=round(win, 2), calls=n_ph, period_ms_per_call=round(win / max(1, n_ph), 3),
                phase_active_frac=round(union(pi) / win, 4), phases_in_flight_avg=round(sum(b - a for a, b in pi) / win, 3),
                phase_ms_mean=round(sum(r["t1"] - r["t0"] for r in phases[skip_calls:]) / max(1, n_ph), 3),
                decode_active_frac=round(union(bi) / win, 4), lanes_active_avg=round(sum(b - a for a, b in bi) / win, 3),
                both_active_frac=round(both(pi, bi) / win, 4),
                step_ms_beside_phases=round(sum(r["t1"] - r["t0"] for r in inside) / max(1, steps), 4) if steps else None,
                rows_per_step=round(sum(r["rows"] * r["steps"] for r in inside) / max(1, steps), 2) if steps else None,
                note="HIP timing events around every call's encoder + prefill phase (caller streams) and every burst of decode steps (scheduler "
                     "lanes), un-profiled run; fractions of the steady window; step_ms_beside_phases = burst time / steps of one lane")


def testset_run(args, rank, world, local, use_dist, affinity=None):
    """The reference's evaluation loop (eval_OmniAVSR.py:273-357: one Trainer.test pass over the test set) as this repo runs it: every rank
    owns the utterances `dist.shard_utterances` deals it (duration-sorted round-robin), batches them longest first (`make_batches`), and
    `harness.evaluate(in_flight=4)` keeps four batches in flight through the engine's decode scheduler with host inputs (collation and
    host-to-device copies inside the timed region).  The only exchange: one all_gather of {utt_id, n, ids} records and one all_reduce of
    the WER counters, inside the timed region before the closing barrier.  Strong scaling: the set is fixed, `value` = N / wall."""
    import torch.distributed as dist
    from omni_avsr_amd import harness as H
    from omni_avsr_amd.dist import gather_hypotheses, pack_records, reduce_wer_counters, shard_utterances
    from omni_avsr_amd.modeling import AVSR_LLMs
    from omni_avsr_amd.synth import synth_state_dict
    dev = torch.device(f"cuda:{local}")
    pre = PRESETS[args.preset]
    spec = build_spec(args, pre)
    task = pre["task"]
    B = args.batch or pre["batch"]
    ra, rv = pre["rates"][0]
    t0 = time.time()
    model = AVSR_LLMs(task, None, True, args.llm or pre["llm"], spec.llm.hidden, 2048, None, None, None, None, None, pre["ra"], pre["rv"], None,
                      "avg-pooling", None, args.max_new, 1, is_task_specific=True, is_matryoshka=True, spec=spec, dtype=args.dtype, device=str(dev))
    model.load_state_dict(synth_state_dict(spec, seed=1234, device=dev))
    eng = model.engine
    if args.sched_rows is None:
        args.sched_rows = eng.sched_rows
    if args.sched_lanes is None:
        args.sched_lanes = eng.sched_lanes
    if args.sched_slots is None:
        args.sched_slots = eng.sched_slots if args.sched_rows == eng.sched_rows else max(192, 4 * args.sched_rows)
    eng.sched_rows, eng.sched_slots, eng.sched_stride, eng.sched_lanes = args.sched_rows, args.sched_slots, args.sched_stride, args.sched_lanes
    torch.cuda.empty_cache()
    # the test set: N utterance descriptors (length, clip), identical on every rank (seeded); the media of utterance u is clip u % P of a
    # pool of P distinct 16 s clips cut to the utterance's length (host memory: P x 50 MB instead of N x 27 MB)
    N, P = args.testset, 24
    g = torch.Generator().manual_seed(20260)
    Tvs = [int(x) for x in torch.randint(25, args.frames + 1, (N,), generator=g)]           # U[1 s, 16 s] at 25 fps
    owned = set(shard_utterances([float(t) for t in Tvs], world, rank))
    pool = {}
    samples = []
    for u in range(N):
        smp = {}
        if u in owned:                         # only this rank's media are materialised
            c = u % P
            if c not in pool:
                gc = torch.Generator().manual_seed(7000 + c)
                v = torch.randn(args.frames, 1, 88, 88, generator=gc)
                a = torch.randn(640 * args.frames, 1, generator=gc)
                pool[c] = (v, a)
            v, a = pool[c]
            if task != "audio":
                smp["video"] = v[: Tvs[u]]
            if task != "video":
                aw = a[: 640 * Tvs[u]]
                smp["audio"] = torch.nn.functional.layer_norm(aw[:, 0], (aw.shape[0],)).unsqueeze(-1)      # transforms.py:123-125
        else:                                  # a descriptor with the right duration is all evaluate() needs for the shards of other ranks
            smp["video" if task != "audio" else "audio"] = torch.empty(Tvs[u] if task != "audio" else 640 * Tvs[u], 0)
        samples.append((smp, " ".join(f"w{(u * 7 + k) % 23}" for k in range(5 + u % 9))))            # a synthetic transcript: the WER counters must flow
    setup_s = time.time() - t0

    def to_text(ids):                          # "tokenizer": one pseudo-word per token id; keeps the ids for the gather
        out = []
        for r in ids.tolist():
            out.append(" ".join(f"w{t % 23}" for t in r))
        to_text.rows.extend(ids.tolist())
        return out
    to_text.rows = []
    kw = dict(batch_size=B, bos_id=spec.ids["bos"], rank=rank, world=world, in_flight=4, test_ratio_matry_audio=ra, test_ratio_matry_video=rv)
    # warm-up: the first batches of this rank's own shard (workspaces, scheduler pool, step graphs, pinned staging)
    H.evaluate(model, [samples[i] for i in sorted(owned, key=lambda i: -Tvs[i])[: 4 * B]], task, to_text, batch_size=B, bos_id=spec.ids["bos"], in_flight=4,
               test_ratio_matry_audio=ra, test_ratio_matry_video=rv)
    to_text.rows = []
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    eng.__dict__["_eval_timing"] = {"wait_collate": 0.0, "submit": 0.0, "collect": 0.0, "batches": 0}      # harness.evaluate's host-side phase timers
    t0 = time.perf_counter()
    meter = H.evaluate(model, samples, task, to_text, **kw)
    torch.cuda.synchronize()
    tm = dict(eng.__dict__["_eval_timing"])
    busy = time.perf_counter() - t0
    W = args.max_new
    ids_t = torch.full((len(to_text.rows), W), -1, dtype=torch.int32)
    lens_t = torch.zeros(len(to_text.rows), dtype=torch.int32)
    for k, r in enumerate(to_text.rows):
        ids_t[k, : len(r)] = torch.tensor(r, dtype=torch.int32)
        lens_t[k] = len(r)
    rec = pack_records(meter.utt_ids, ids_t.to(dev), lens_t.to(dev), W)
    hyp = gather_hypotheses(rec, N)
    edits, words, failed = reduce_wer_counters(meter.total_edit_distance, meter.total_length, device=dev, failed=len(meter.failed))
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    busy_all = [busy]
    if use_dist:
        t = torch.tensor([elapsed, busy], device=dev, dtype=torch.float64)
        tl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(tl, t)
        elapsed = max(float(x[0]) for x in tl)
        busy_all = [float(x[1]) for x in tl]
    assert len(hyp) == N and failed == 0, (len(hyp), failed)
    # identity check on a sample of the set (outside the timed region): the first and the last batch this rank decoded, each again as ONE
    # isolated call -- same ids as inside the pipelined evaluation
    from omni_avsr_amd.dist import make_batches
    mine = shard_utterances([float(t) for t in Tvs], world, rank)
    bs = list(make_batches(mine, B))
    for idxs in ([bs[0], bs[-1]] if len(bs) > 1 else bs):
        m2 = H.evaluate(model, [samples[i] for i in idxs], task, lambda ids: [" ".join(str(int(t)) for t in r) for r in ids.tolist()], batch_size=B,
                        bos_id=spec.ids["bos"], in_flight=1, test_ratio_matry_audio=ra, test_ratio_matry_video=rv)
        got = {idxs[m2.utt_ids[k]]: [int(t) for t in m2.per_utt[k][1].split()] for k in range(len(idxs))}
        for u_, ids_ in got.items():
            if hyp[u_] != ids_:
                print(f"bench.py --testset: rank {rank}: utterance {u_} decodes differently inside the evaluation than alone", file=sys.stderr, flush=True)
                sys.exit(3)
    if rank == 0:
        secs = sum(Tvs) / 25.0
        llm_name = (args.llm or pre["llm"]).split("/")[-1]
        per_rank = [len(shard_utterances([float(t) for t in Tvs], world, r)) for r in range(world)]
        line = {
            "metric": "utterances/s + real-time-factor, LRS3 AVSR 16s clips, Llama-3.2-1B, 1/2/4/8 GPU",
            "value": round(N / elapsed, 3), "unit": "utterances/s", "n_gpus": world, "steps": 1, "warmup": 1,
            "ms_per_step": round(elapsed * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic", "rtf": round(elapsed / secs, 6),
            "config": {"workload": f"test-set-shaped run (SURVEY.md 8d C5 / BASELINE.json configs[4] shape): {N} utterances, lengths U[1 s, {args.frames / 25:.0f} s], "
                                   f"{task}, {llm_name} (Omni-LoRA ST r32/a4), batches of {B}, rates ({ra}, {rv}), greedy <= {args.max_new} tokens (natural EOS), "
                                   f"HOST inputs (collation + host-to-device copies timed), preset {args.preset} (NOT the headline configuration)",
                       "utterances": N, "audio_seconds": round(secs, 1), "batch_per_call": B, "in_flight": 4,
                       "parallelism": f"utterance-sharded dp{world}, duration-sorted round-robin", "utterances_per_rank": per_rank,
                       "rank_busy_s": [round(b, 3) for b in busy_all], "imbalance": round(max(busy_all) / max(1e-9, min(busy_all)), 4),
                       "host_ms_per_batch": {k: round(tm[k] / max(1, tm["batches"]) * 1e3, 2) for k in ("wait_collate", "submit", "collect")},
                       "gathered_hypotheses": len(hyp), "wer_counters": {"edits": edits, "words": words, "failed": failed},
                       "dist_backend": (dist.get_backend() if use_dist else None), "setup_s": round(setup_s, 1), "ids_verified": True,
                       "cpu_affinity": affinity, "decode_scheduler": dict(eng.scheduler_stats(), pool_bytes=eng.scheduler_bytes()), "env": env_snapshot()},
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def pin_rank(local: int, world_local: int):
    """CPU affinity of this rank, set BEFORE anything touches the GPU (threads created later -- the collation helpers, the runtime's own --
    inherit it): the cores of the NUMA node the rank's GPU hangs off, shared evenly by the ranks whose GPUs sit on the same node; without
    NUMA information an even slice of the host's CPUs.  Reads sysfs only.  Returns what it did (for `config`)."""
    try:
        allc = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None

    def cpulist(txt):
        out = []
        for part in txt.strip().split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            out.extend(range(int(a), int(b or a) + 1))
        return out
    gpus = []                                  # (pci address, numa node) of the AMD display / processing accelerators, in bus order = HIP's device order
    try:
        for dev in sorted(os.listdir("/sys/bus/pci/devices")):
            base = os.path.join("/sys/bus/pci/devices", dev)
            try:
                if open(os.path.join(base, "vendor")).read().strip() != "0x1002":
                    continue
                cls = open(os.path.join(base, "class")).read().strip()
                if not (cls.startswith("0x0302") or cls.startswith("0x0380") or cls.startswith("0x1200")):
                    continue
                gpus.append((dev, int(open(os.path.join(base, "numa_node")).read().strip())))
            except (OSError, ValueError):
                continue
    except OSError:
        pass
    node, cpus = -1, None
    if local < len(gpus) and gpus[local][1] >= 0:
        node = gpus[local][1]
        try:
            ncpus = [c for c in cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) if c in set(allc)]
        except OSError:
            ncpus = []
        peers = [i for i in range(min(world_local, len(gpus))) if gpus[i][1] == node]
        if ncpus and local in peers:
            k, n = peers.index(local), len(peers)
            per = max(1, len(ncpus) // n)
            cpus = ncpus[k * per:(k + 1) * per] or ncpus
    if cpus is None:                           # no NUMA information: an even slice of what the process may use
        per = max(1, len(allc) // max(1, world_local))
        cpus = allc[local * per:(local + 1) * per] or allc
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return {"numa_node": node, "cpus": len(cpus), "first_cpu": cpus[0], "gpus_seen": len(gpus)}


def launch_ranks(args):
    """Parent of a multi-GPU run: never touches the GPU (device_count() does not initialise it), starts the ranks as a child
    process and hands their return code back."""
    n = torch.cuda.device_count()
    if n < args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this node has {n}", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    if os.environ.get("OAVSR_BENCH_PRINT_LAUNCH") == "1":       # tests: show the command instead of running it
        print(json.dumps(cmd), flush=True)
        return 0
    return subprocess.call(cmd)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if args.streams is None:
        args.streams = 4
    if args.beams > 1:
        # beam search runs one decode chain per call on the call's own stream: four chains in flight need more than ROCm's default four
        # hardware queues to advance independently (120.6 -> 133.6 utt/s with 8; the greedy scheduler path is best at the default)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    affinity = pin_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if world > 1 else None      # before any GPU call (one rank: the whole host is its own)
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr, flush=True)
        sys.exit(2)
    use_dist = world > 1 or os.environ.get("OAVSR_BENCH_FORCE_DIST") == "1"   # the latter: exercise the RCCL path under torchrun with one rank
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    if args.testset > 0:
        return testset_run(args, rank, world, local, use_dist, affinity)
    from omni_avsr_amd.dist import gather_hypotheses, pack_records
    from omni_avsr_amd.modeling import OmniEngine
    from omni_avsr_amd.synth import synth_state_dict

    pre = PRESETS[args.preset]
    spec = build_spec(args, pre)
    task = pre["task"]
    use_a = task in ("audio", "audiovisual")
    t0 = time.time()
    big = spec.llm.hidden > 2048                     # 7B / 8B: no CPU-baseline leg (the fp32 oracle would run for minutes)
    sd = synth_state_dict(spec, seed=1234, device=dev)
    eng = OmniEngine(spec, dtype=args.dtype, device=str(dev)).load_state_dict(sd)
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not big:
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    B, Tv = args.batch or pre["batch"], args.frames
    G = max(1, args.decode_group)
    calls = max(1, args.calls_per_step)
    assert (calls % G == 0 and args.beams == 1) or G == 1, "--decode-group needs greedy decode and --calls-per-step divisible by G"
    nsets = max(1, args.input_sets)
    sets = []
    for s in range(nsets):
        if pre["varlen"]:     # configs[4]: utterance lengths U[1 s, 16 s]
            gl = torch.Generator().manual_seed(1000 * rank + s)
            Tvs = [int(x) for x in torch.randint(25, Tv + 1, (B * G,), generator=gl)]
        else:
            Tvs = [Tv] * (B * G)
        video, audio = make_inputs(Tvs, 100 + 17 * rank + s, dev)
        sets.append((video, audio, Tvs, [640 * t for t in Tvs]))
    setup_s = time.time() - t0

    # Independent batches are pipelined over `--streams` HIP streams (each with its own workspace): the latency-bound decode
    # loop of one batch overlaps the MFMA-bound encoders / prefill of the next.  Every call still is one complete pass.
    nstream = max(1, args.streams)
    results = [None] * nstream
    # The decode scheduler (default): every call is still one batch of B utterances submitted on its own stream; the engine merges the
    # decode rows of the calls in flight into one step loop (oavsr_generate_submit).  Caller-side nothing is grouped (decode_group 1).
    if args.sched_rows is None:
        args.sched_rows = eng.sched_rows
    if args.sched_lanes is None:
        args.sched_lanes = eng.sched_lanes
    if args.sched_slots is None:
        args.sched_slots = eng.sched_slots if args.sched_rows == eng.sched_rows else max(192, 4 * args.sched_rows)
    use_sched = not args.no_scheduler and args.beams == 1 and G == 1 and B <= args.sched_rows
    eng.launch_objective(bool(args.alone))
    eng.sched_rows, eng.sched_slots, eng.sched_stride, eng.sched_lanes = args.sched_rows, args.sched_slots, args.sched_stride, args.sched_lanes
    if use_sched and nstream <= 8:
        # the ENGINE's caller streams (created before its lanes' streams: the hardware-queue binding order no longer depends on the caller)
        eng.reserve_scheduler(task, max(max(s[2]) for s in sets), max(max(s[3]) for s in sets),
                              min(r[0] for r in pre["rates"]), min(r[1] for r in pre["rates"]), args.max_new)
        streams = [eng.caller_stream(i) for i in range(nstream)]
    else:
        streams = [torch.cuda.Stream(device=dev) for _ in range(nstream)]
    pending = []
    timed = None                                     # inside the timed region: [(call index, ids, lens, digests)] of every call, for the identity check after it
    # per-utterance {inputs_embeds, first-step logits} digests of every timed call (oavsr_workspace_set_digests): allocated up front, two tiny
    # launches per call inside the timed region
    dg_pool = torch.zeros(args.steps * max(1, calls // G) + 1, B * G, 2, dtype=torch.int64, device=dev)
    # Without the scheduler the encoder + prefill phases of the calls in flight are chained (only the decode loops overlap them); with it
    # the callers' streams carry nothing but those phases, and letting the tail of one overlap the head of the next is worth 5 %.
    if nstream > 1 and not args.no_stagger and not use_sched:
        eng.serialize_encoders(True)

    def call(i):
        s = i % nstream
        video, audio, Tvs, Ts = sets[i % nsets]
        ra, rv = pre["rates"][i % len(pre["rates"])]
        dg = dg_pool[len(timed)] if timed is not None and len(timed) < dg_pool.shape[0] else None
        with torch.cuda.stream(streams[s]):
            if use_sched:
                ids_, lens_, ticket = eng.submit(task, video, Tvs, audio if use_a else None, Ts if use_a else None, ra, rv, max_new=args.max_new,
                                                 force_len=True, video_is_flat=True, slot=s, digests=dg)
                pending.append((s, ticket))
                if len(pending) > 2 * nstream + args.pending_per_lane * args.sched_lanes:            # every step of that call was issued long ago: a stream-side wait, no host sync
                    s0, t0_ = pending.pop(0)
                    with torch.cuda.stream(streams[s0]):
                        eng.wait(t0_)
                results[s] = (ids_, lens_)
                if timed is not None:
                    timed.append((i, ids_, lens_, dg))
            else:
                results[s] = eng.generate(task, video, Tvs, audio if use_a else None, Ts if use_a else None, ra, rv, beams=args.beams,
                                          max_new=args.max_new, force_len=(args.beams == 1), video_is_flat=True, slot=s, chunk=B if G > 1 else None,
                                          digests=dg)
                if timed is not None:
                    timed.append((i, results[s][0], results[s][1], dg))
        return results[s]

    def drain():
        if use_sched:
            eng.flush()
            pending.clear()
        torch.cuda.synchronize()

    ncall = 0
    torch.cuda.synchronize()
    for _ in range(max(args.warmup * (calls // G), nstream, nsets * len(pre["rates"]))):
        ids, lens = call(ncall)
        ncall += 1
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    timed = []
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps * (calls // G)):       # one call = G batches of B when --decode-group G
        th = time.perf_counter()
        ids, lens = call(ncall)
        host_s += time.perf_counter() - th            # host time inside the call (launches; includes waiting for queue / staging space)
        ncall += 1
    drain()
    busy = time.perf_counter() - t0               # this rank's own work (before the exchange and the closing barrier)
    n_last = B * G
    # the path's only exchange step, inside the timed region: gather the hypotheses of the last call
    hyp = gather_hypotheses(pack_records([rank * n_last + j for j in range(n_last)], ids, lens, args.max_new), world * n_last)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rank_busy = [round(busy, 3)]
    if use_dist:
        t = torch.tensor([elapsed, busy], device=dev, dtype=torch.float64)
        tl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(tl, t)
        elapsed = max(float(x[0]) for x in tl)
        rank_busy = [round(float(x[1]), 3) for x in tl]
    assert len(hyp) == world * n_last
    assert args.beams > 1 or int(lens.min()) == args.max_new
    # ---- identity check (outside the timed region): the input sets and rates rotate, so every timed call must have produced exactly the
    # ids of ONE ISOLATED oavsr_generate of its (input set, rates) -- nothing else on the chip, no scheduler, default stream.  A mismatch
    # means the pipelined / scheduled configuration the number was measured in does not compute what a single call computes: no line then.
    timed_calls, timed = timed, None
    torch.cuda.synchronize()
    iso = {}
    bad, bad_dg = [], []
    for i, ids_t, lens_t, dg_t in timed_calls:
        key = (i % nsets, i % len(pre["rates"]))
        if key not in iso:
            video, audio, Tvs, Ts = sets[key[0]]
            ra, rv = pre["rates"][key[1]]
            r_dg = torch.zeros(B * G, 2, dtype=torch.int64, device=dev)
            r_ids, r_lens = eng.generate(task, video, Tvs, audio if use_a else None, Ts if use_a else None, ra, rv, beams=args.beams, max_new=args.max_new,
                                         force_len=(args.beams == 1), video_is_flat=True, slot=0, chunk=B if G > 1 else None, digests=r_dg)
            torch.cuda.synchronize()
            iso[key] = (r_ids.clone(), r_lens.clone(), r_dg)
        if not (torch.equal(ids_t, iso[key][0]) and torch.equal(lens_t, iso[key][1])):
            bad.append(i)
        if dg_t is not None and not torch.equal(dg_t, iso[key][2]):
            bad_dg.append(i)
    # the ids of seeded random weights hardly depend on the clip; the digests do: every utterance of every input set must have its own pair,
    # otherwise "equal to the isolated call" could not tell a call that was handed another call's inputs, KV rows or logits
    alld = torch.cat([v[2] for v in iso.values()])
    distinct = len({int(x) for x in alld[:, 0]}) == alld.shape[0] and len({int(x) for x in alld[:, 1]}) == alld.shape[0]
    n_dg = sum(1 for t in timed_calls if t[3] is not None)
    ok_t = torch.tensor([0 if (bad or bad_dg or not distinct) else 1], device=dev, dtype=torch.int32)
    if use_dist:
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
    if bad:
        print(f"bench.py: rank {rank}: {len(bad)} of {len(timed_calls)} timed calls returned ids that differ from an isolated call of the same inputs "
              f"(first: call {bad[0]})", file=sys.stderr, flush=True)
    if bad_dg:
        print(f"bench.py: rank {rank}: {len(bad_dg)} of {n_dg} timed calls report input digests that differ from the isolated call's (first: call {bad_dg[0]}): "
              "the call did not see its own inputs_embeds / first-step logits", file=sys.stderr, flush=True)
    if not distinct:
        print(f"bench.py: rank {rank}: the input sets' digests are not pairwise different: the identity check is not input-sensitive", file=sys.stderr, flush=True)
    if int(ok_t.item()) != 1:
        if use_dist:
            dist.destroy_process_group()
        sys.exit(3)
    n_verified = len(timed_calls)
    del timed_calls, iso, alld

    dsched = eng.scheduler_stats() if use_sched else None      # the timed region's (the passes below run more calls)
    sched_bytes = eng.scheduler_bytes() if use_sched else 0
    schedule = None
    if rank == 0 and use_sched and not args.no_roofline:
        # the same loop for 32 more calls with the scheduler's trace on: what overlaps what (a kernel trace cannot show it: the profiler
        # takes the hardware queues in turn)
        eng.scheduler_trace(True)
        for _ in range(32):
            call(ncall)
            ncall += 1
        drain()
        schedule = schedule_summary(eng.scheduler_trace_read(), skip_calls=4 * args.sched_lanes)
        eng.scheduler_trace(False)
    roofline = None
    stages = None
    sides = None
    if rank == 0 and not args.no_roofline:
        torch.cuda.synchronize()
        # (1) per-kernel-family HIP-event brackets, one call at a time (a bracket times whatever shares the chip with its kernel, so the
        # calls of this pass do not overlap): what each encoder / projector / prefill kernel costs.  Decode families: pass (3).
        nprof = max(2, len(pre["rates"]))
        eng.profile(True)
        for _ in range(nprof):
            call(ncall)
            ncall += 1
            drain()
        ents = eng.profile_read()
        eng.profile(False)
        tot_ms = sum(e["ms"] for e in ents) or 1.0
        stages = {}
        for e in sorted(ents, key=lambda e: -e["ms"]):
            avg_us = e["ms"] / max(1, e["launches"]) * 1e3
            stages[e["name"]] = dict(ms_per_call=round(e["ms"] / nprof, 3), cu_ms_per_call=round(e.get("cu_ms", e["ms"]) / nprof, 3),
                                     launches_per_call=round(e["launches"] / nprof, 1), avg_us=round(avg_us, 1),
                                     tflops=round(e["flops"] / e["ms"] / 1e9, 1) if e["ms"] > 0 and e["flops"] > 0 else None,
                                     gbs=round(e["bytes"] / e["ms"] / 1e6, 1) if e["ms"] > 0 else None)
            if avg_us < 20.0:      # an event bracket resolves a launch to about +-2 us: rates of these families are indicative only
                stages[e["name"]]["short"] = True
        is_dec = lambda n: n.startswith(("decode_", "argmax", "sched_", "beam_"))          # noqa: E731   (decode-shaped launches: HBM-bound, weights streamed once per step)
        is_mfma = lambda e: e["flops"] > 0 and not is_dec(e["name"]) and "logmel" not in e["name"]           # noqa: E731
        dom = max((e for e in ents if not is_dec(e["name"])), key=lambda e: e["ms"], default=None) or max(ents, key=lambda e: e["ms"])
        traffic = pmc_traffic(dom["name"])
        if is_mfma(dom):
            ach = dom["flops"] / dom["ms"] / 1e9
            peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
            roofline = dict(bound="mfma", kernel=dom["name"], achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                            traffic=traffic, avg_launch_ms=round(dom["ms"] / dom["launches"], 4), launches=dom["launches"],
                            share_of_step=round(dom["ms"] / tot_ms, 3))
            # the same flops over the family's CU-time (duration x fraction of the CUs a launch occupies): the dispatcher runs launches of
            # 64 .. 255 tiles on as many CUs as they have tiles -- the rest of the chip belongs to the other calls in flight -- so `frac`
            # (whole-chip peak, one call at a time) understates what a CU delivers while it works on this family
            if dom.get("cu_ms", 0) > 0:
                roofline["cu_frac"] = round(dom["flops"] / dom["cu_ms"] / 1e9 / peak, 4)
        else:
            ach = dom["bytes"] / dom["ms"] / 1e6
            roofline = dict(bound="hbm", kernel=dom["name"], achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                            traffic=traffic, avg_launch_ms=round(dom["ms"] / dom["launches"], 4), launches=dom["launches"],
                            share_of_step=round(dom["ms"] / tot_ms, 3))
        # (2) the two sides of the job.  MFMA side: every encoder / projector / prefill kernel that does flops (long kernels: the brackets are
        # accurate), algorithmic flops / their summed time.  Decode side: what a weight pass costs is measured on the real thing -- the
        # scheduler's step loop replayed from its hipGraph with nothing else on the chip -- algorithmic bytes per step / wall time per step.
        peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
        mf = [e for e in ents if is_mfma(e)]
        mfma_ms = sum(e["ms"] for e in mf) / nprof
        mfma_fl = sum(e["flops"] for e in mf) / nprof
        other_ms = sum(e["ms"] for e in ents if not is_mfma(e) and not is_dec(e["name"])) / nprof
        # CU-time: duration x the fraction of the chip's CUs the launch can occupy (GEMMs: by the tile the dispatcher picked; every other
        # kernel counts as the whole chip).  In the pipelined job the CUs a coarse launch leaves free run other calls' kernels and the decode
        # steps, so THIS, not the one-call-at-a-time latency above, is what a call costs the job (the same box
        # went 319 -> 327 utt/s while mfma_ms_per_call went 22.3 -> 23.4).
        mfma_cu_ms = sum(e.get("cu_ms", e["ms"]) for e in mf) / nprof
        sides = dict(mfma_ms_per_call=round(mfma_ms, 3), mfma_tflop_per_call=round(mfma_fl / 1e12, 3), mfma_frac=round(mfma_fl / mfma_ms / 1e9 / peak, 4) if mfma_ms > 0 else None,
                     mfma_cu_ms_per_call=round(mfma_cu_ms, 3), mfma_cu_frac=round(mfma_fl / mfma_cu_ms / 1e9 / peak, 4) if mfma_cu_ms > 0 else None,
                     other_ms_per_call=round(other_ms, 3))
        if use_sched:
            l = spec.llm
            esz = 2 if args.dtype == "bf16" else 4
            nqkv = (l.heads + 2 * l.kv_heads) * l.head_dim
            from omni_avsr_amd.weights import lora_k2
            k2 = lora_k2(spec)
            w_layer = nqkv * (l.hidden + k2) + k2 * l.hidden + l.hidden * l.heads * l.head_dim + 3 * l.mlp * l.hidden
            step_bytes = esz * (l.layers * w_layer + l.vocab * l.hidden)             # weights streamed once per step (KV and activations: < 3 %)
            k = max(1, args.sched_rows // B)
            eng.sched_lanes, eng.sched_stride = 1, 1                                 # ONE step loop, filled as fast as the calls arrive

            def fill():
                for j in range(k):
                    call(ncall + j)
                torch.cuda.synchronize()
                return eng.scheduler_stats()
            # (2) wall time of the real step loop: hipGraph replays of the steps the k admitted calls still need, nothing else on the chip
            st1 = fill()
            t1 = time.perf_counter()
            drain()
            dt = time.perf_counter() - t1
            st2 = eng.scheduler_stats()
            nst = st2["steps"] - st1["steps"]
            if nst > 0:
                step_ms = dt * 1e3 / nst
                sides.update(decode_step_ms=round(step_ms, 4), decode_rows_per_step=round((st2["row_steps"] - st1["row_steps"]) / nst, 1),
                             decode_gb_per_step=round(step_bytes / 1e9, 3), decode_hbm_frac=round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             decode_steps_per_call=round(dsched["steps"] / max(1, dsched["calls"]), 2),
                             decode_note=f"{nst} hipGraph-replayed steps of one step loop ({k} calls admitted), nothing else on the chip; hbm_frac = weight "
                                         f"bytes per step / step time / {HBM_PEAK_GBS:.0f} GB/s; steps_per_call: the timed region's")
            # (3) the same steps eagerly with brackets: the decode kernel families at the row count they run at under the scheduler
            st3 = fill()
            eng.profile(True)
            drain()
            dents = eng.profile_read()
            eng.profile(False)
            for name in [n for n in stages if is_dec(n)]:
                del stages[name]
            nsteps = max(1, eng.scheduler_stats()["steps"] - st3["steps"])
            for e in sorted(dents, key=lambda e: -e["ms"]):
                if not is_dec(e["name"]) or e["launches"] == 0:
                    continue
                avg_us = e["ms"] / e["launches"] * 1e3
                stages[e["name"]] = dict(ms_per_step=round(e["ms"] / nsteps, 4), launches_per_step=round(e["launches"] / nsteps, 1), avg_us=round(avg_us, 1),
                                         gbs=round(e["bytes"] / e["ms"] / 1e6, 1) if e["ms"] > 0 else None)
                if avg_us < 20.0:
                    stages[e["name"]]["short"] = True
            eng.sched_lanes, eng.sched_stride = args.sched_lanes, args.sched_stride

    # Extra, reported next to the headline and never as `value`: the same job with the engine's decode scheduler switched off
    grouped = None
    if rank == 0 and world == 1 and use_sched and not args.no_grouped and args.preset == "c1" and not args.llm:
        try:
            grouped = per_call_leg(eng, args, pre, task, use_a, B, sets, streams, nstream)
        except Exception as e:                               # the headline line must not depend on this leg
            grouped = {"error": f"{type(e).__name__}: {e}"[:200]}

    # The reference's own protocol (datamodule/data_module.py:238-242: batch 1, one call at a time): what ONE 16 s utterance costs alone on an
    # otherwise idle chip, greedy and with the reference's default K = 15 (eval_OmniAVSR.py:215-226), median of 10 calls each
    latency = None
    if rank == 0 and world == 1 and not args.no_grouped and args.beams == 1:
        try:
            latency = latency_leg(eng, args, pre, task, use_a, sets[0])
        except Exception as e:                               # the headline line must not depend on this leg
            latency = {"error": f"{type(e).__name__}: {e}"[:200]}

    cpu = None
    if sd_cpu is not None:
        cpu = cpu_baseline(spec, sd_cpu, Tv, args.max_new, task, pre["rates"][0])

    if rank == 0:
        n_utt = world * B * calls * args.steps
        value = n_utt / elapsed
        secs = sum(sum(s[2]) for s in sets) / (nsets * G) / 25.0            # mean audio seconds per model call
        llm_name = (args.llm or pre["llm"]).split("/")[-1]
        what = {"audiovisual": "AVSR, AV-HuBERT-Large + Whisper-medium.en", "video": "VSR, AV-HuBERT-Large visual branch", "audio": "ASR, Whisper-medium.en"}[task]
        line = {
            "metric": "utterances/s + real-time-factor, LRS3 AVSR 16s clips, Llama-3.2-1B, 1/2/4/8 GPU",
            "value": round(value, 3), "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "rtf": round(elapsed / (world * calls * args.steps * secs), 6),
            "config": {"workload": f"{what} + {llm_name} (Omni-LoRA ST r32/a4), batch={B} x "
                                   + ("U[1,16]s" if pre["varlen"] else f"{Tv / 25:.0f}s") + " clips per model call per GPU, "
                                   f"compression rates {pre['rates']}, "
                                   + (f"greedy {args.max_new} new tokens (EOS suppressed)" if args.beams == 1 else f"HF beam search K={args.beams}, {args.max_new} steps")
                                   + f", BASELINE.json configs[{pre['idx']}]" + ("" if args.preset == "c1" and not args.llm else " (NOT the headline configuration)"),
                       "batch_per_call": B, "calls_per_step": calls, "ms_per_call": round(elapsed / (args.steps * calls) * 1e3, 3),
                       "global_batch": world * B * calls, "input_sets": nsets, "timed_s": round(elapsed, 2),
                       "host_ms_per_call": round(host_s / (args.steps * (calls // G)) * 1e3, 3),
                       "parallelism": f"utterance-sharded dp{world}", "gathered_hypotheses": len(hyp), "dist_backend": (dist.get_backend() if use_dist else None),
                       "streams_per_gpu": nstream, "decode_group": G, "setup_s": round(setup_s, 1),
                       "ids_verified": True, "ids_verified_calls": n_verified, "digests_verified": n_dg == n_verified and n_dg > 0, "digests_verified_calls": n_dg,
                       "cpu_affinity": affinity, "rank_busy_s": rank_busy, "env": env_snapshot(),
                       "decode_scheduler": (dict(rows_per_step_cap=args.sched_rows, slots=args.sched_slots, lanes=args.sched_lanes, pool_bytes=sched_bytes, **dsched) if use_sched else None)},
            "roofline": roofline, "roofline_sides": sides, "schedule": schedule, "cpu_baseline": cpu, "per_call_decode": grouped, "latency_b1": latency,
            "stages": stages,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
