#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): utterances/s of the Omni-AVSR inference hot path on synthetic 16 s clips.

  python bench.py [--gpus N --steps K --warmup W]          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the whole hot path (`AVSR_LLMs.forward(is_trainval=False)`: log-mel + Whisper encoder, ResNet +
AV-HuBERT encoder, compressor + projectors, prompt splice, LLM prefill, 32 greedy decode steps with EOS suppressed so
every run does the same work) over one batch of B utterances per GPU.  Inputs are resident in HBM before the timed
region.  Workload = BASELINE.json configs[1]: AVSR, AV-HuBERT-Large + Whisper-medium.en + Llama-3.2-1B (Omni-LoRA
task-specific + shared, rank 32 / alpha 4, LoRA on the video encoder), batch 8 x 16 s, compression rates (4, 4).
Weights are seeded random tensors of the real architecture (no checkpoints offline).  dtype: bf16 storage / fp32
accumulate (the reference evaluates in bf16-true, eval_OmniAVSR.py:19).

Multi-GPU: the path shards by utterance with no data-path collective: every rank runs its own batch
("weak" scaling); the only exchange is one all_gather of the hypotheses after the timed loop.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def build_spec(args):
    from omni_avsr_amd.config import spec_from_names
    return spec_from_names(
        args.llm, "openai/whisper-medium.en", "large_vox_iter5", use_lora_avhubert=True,
        intermediate_size=2048, compression_mode="avg-pooling", downsample_ratio_audio=[4, 16], downsample_ratio_video=[2, 4, 5],
        is_matryoshka=True, is_single_matry_projector=False, lora_rank=32, lora_alpha=4, is_task_specific=True, shared_lora=True,
        prompt_len={"audio": 8, "video": 8, "audiovisual": 8}, max_dec_tokens=args.max_new, num_beams=1)


def make_inputs(B, Tv, seed, device):
    """video N(0,1) [B,T_v,88,88]; audio N(0,1), utterance-level layer-norm (transforms.py:123-125)."""
    g = torch.Generator(device=device).manual_seed(seed)
    video = torch.randn(B * Tv, 88, 88, generator=g, device=device, dtype=torch.float32)
    audio = torch.randn(B, 640 * Tv, generator=g, device=device, dtype=torch.float32)
    audio = torch.nn.functional.layer_norm(audio, (audio.shape[1],))
    return video, audio


def cpu_baseline(spec, sd_cpu, Tv, max_new):
    """The oracle (CPU fp32 restatement, kind 'port') timed on this host's cores on ONE utterance of the same workload."""
    from oracle import torch_oracle as O
    orc = O.OmniOracle(sd_cpu, spec.oracle_cfg())
    g = torch.Generator().manual_seed(7)
    video = torch.randn(1, Tv, 1, 88, 88, generator=g)
    audio = torch.nn.functional.layer_norm(torch.randn(1, 640 * Tv, generator=g), (640 * Tv,)).unsqueeze(-1)
    inp = dict(video=video, audio=audio, lengths=torch.tensor([640 * Tv]), tokens=torch.full((1, 1), spec.ids["bos"]))
    t0 = time.time()
    with torch.no_grad():
        out = orc.forward(inp, "audiovisual", 4, 4, num_beams=1, force_len=True)
    dt = time.time() - t0
    assert len(out[0]) == max_new
    return dict(value=1.0 / dt, unit="utterances/s", cores=torch.get_num_threads(), kind="port",
                sample=f"1 utterance of the same workload (16 s clip, greedy, {max_new} tokens) in {dt:.1f} s, torch CPU fp32")


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic_*.json: FETCH_SIZE and
    WRITE_SIZE collected in separate runs, gfx950 read-side correction applied; counters cannot be read from inside this process)."""
    import glob
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic_*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel") == kernel:
            return d.get("traffic_bytes_per_launch")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=400, help="video frames per utterance (400 = 16 s)")
    ap.add_argument("--max-new", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--beams", type=int, default=1, help="1 = greedy (north-star decode); 15 = the reference's default HF beam search")
    ap.add_argument("--llm", default="meta-llama/Llama-3.2-1B",
                    help="LLM architecture (random-init); the headline number is quoted on Llama-3.2-1B.  E.g. meta-llama/Meta-Llama-3.1-8B, Qwen/Qwen2.5-7B")
    ap.add_argument("--decode-group", type=int, default=1,
                    help="G > 1: G consecutive steps (batches) share ONE decode loop (encoders / prefill still run per batch of --batch; "
                         "continuous batching of the decode phase).  Opt-in: not the configuration the headline number is quoted on")
    ap.add_argument("--no-stagger", action="store_true", help="let the encoder phases of in-flight batches share the chip (default: chained)")
    ap.add_argument("--streams", type=int, default=3, help="batches in flight (one HIP stream + workspace each); 1 = strictly sequential steps")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("OAVSR_BENCH_FORCE_DIST") == "1"   # the latter: exercise the RCCL path under torchrun with one rank
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    from omni_avsr_amd.modeling import OmniEngine
    from omni_avsr_amd.synth import synth_state_dict

    spec = build_spec(args)
    t0 = time.time()
    sd = synth_state_dict(spec, seed=1234, device=dev)
    eng = OmniEngine(spec, dtype=args.dtype, device=str(dev)).load_state_dict(sd)
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    B, Tv = args.batch, args.frames
    G = max(1, args.decode_group)
    assert args.steps % G == 0 and args.beams == 1 or G == 1, "--decode-group needs greedy decode and steps divisible by G"
    video, audio = make_inputs(B * G, Tv, 100 + rank, dev)
    Tvs, Ts = [Tv] * (B * G), [640 * Tv] * (B * G)
    setup_s = time.time() - t0

    # Independent batches are pipelined over `--streams` HIP streams (each with its own workspace): the latency-bound decode
    # loop of one batch overlaps the MFMA-bound encoders / prefill of the next.  Every step still is one complete pass.
    nstream = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstream)]
    if nstream > 1 and not args.no_stagger:
        eng.serialize_encoders(True)
    results = [None] * nstream

    def step(i=0):
        s = i % nstream
        with torch.cuda.stream(streams[s]):
            results[s] = eng.generate("audiovisual", video, Tvs, audio, Ts, 4, 4, beams=args.beams, max_new=args.max_new, force_len=(args.beams == 1),
                                      video_is_flat=True, slot=s, chunk=B if G > 1 else None)
        return results[s]

    torch.cuda.synchronize()
    for i in range(max(args.warmup, nstream)):
        ids, lens = step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps // G):                 # one call = G steps (G batches of B) when --decode-group G
        ids, lens = step(i)
    torch.cuda.synchronize()
    if use_dist:
        # the path's only exchange step, inside the timed region: gather the hypotheses of the last batch (K16)
        rec = torch.cat([lens.view(-1, 1), ids], 1).contiguous()
        gathered = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(gathered, rec)
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert args.beams > 1 or int(lens.min()) == args.max_new

    roofline = None
    stages = None
    if rank == 0 and not args.no_roofline:
        eng.profile(True)
        step()
        ents = eng.profile_read()
        eng.profile(False)
        tot_ms = sum(e["ms"] for e in ents) or 1.0
        stages = {e["name"]: dict(ms=round(e["ms"], 3), launches=e["launches"],
                                   tflops=round(e["flops"] / e["ms"] / 1e9, 1) if e["ms"] > 0 and e["flops"] > 0 else None,
                                   gbs=round(e["bytes"] / e["ms"] / 1e6, 1) if e["ms"] > 0 else None) for e in sorted(ents, key=lambda e: -e["ms"])}
        dom = max(ents, key=lambda e: e["ms"])
        traffic = pmc_traffic(dom["name"])
        is_mfma = dom["flops"] > 0 and (("gemm" in dom["name"] and "skinny" not in dom["name"] and "splitk" not in dom["name"]) or dom["name"].startswith("attention_"))
        if is_mfma:
            ach = dom["flops"] / dom["ms"] / 1e9
            peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
            roofline = dict(bound="mfma", kernel=dom["name"], achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                            traffic=traffic, avg_launch_ms=round(dom["ms"] / dom["launches"], 4), launches=dom["launches"],
                            share_of_step=round(dom["ms"] / tot_ms, 3))
        else:
            ach = dom["bytes"] / dom["ms"] / 1e6
            roofline = dict(bound="hbm", kernel=dom["name"], achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                            traffic=traffic, avg_launch_ms=round(dom["ms"] / dom["launches"], 4), launches=dom["launches"],
                            share_of_step=round(dom["ms"] / tot_ms, 3))

    cpu = None
    if sd_cpu is not None:
        cpu = cpu_baseline(spec, sd_cpu, Tv, args.max_new)

    if rank == 0:
        n_utt = world * B * args.steps
        value = n_utt / elapsed
        line = {
            "metric": "utterances/s + real-time-factor, LRS3 AVSR 16s clips, Llama-3.2-1B, 1/2/4/8 GPU",
            "value": round(value, 3), "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "rtf": round(elapsed / (n_utt * Tv / 25.0), 6),
            "config": {"workload": f"AVSR, AV-HuBERT-Large + Whisper-medium.en + {args.llm.split('/')[-1]} (Omni-LoRA ST r32/a4), batch={B} x {Tv / 25:.0f}s clips per GPU, "
                                   f"compression (4,4), " + (f"greedy {args.max_new} new tokens (EOS suppressed)" if args.beams == 1 else f"HF beam search K={args.beams}, {args.max_new} steps") + (", BASELINE.json configs[1]" if args.llm.endswith("Llama-3.2-1B") else ", NOT the headline configuration (other LLM)"),
                       "global_batch": world * B, "parallelism": f"utterance-sharded dp{world}", "streams_per_gpu": nstream, "decode_group": G, "setup_s": round(setup_s, 1)},
            "roofline": roofline, "cpu_baseline": cpu, "stages": stages,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
