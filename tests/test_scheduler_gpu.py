"""-m gpu: the decode scheduler (continuous batching of the greedy decode inside the engine: oavsr_generate_submit / _wait).

The reference decodes one utterance per forward() (eval_OmniAVSR.py:18-24, datamodule/data_module.py:238-242), so what has to hold is
the per-utterance result: a call's ids must be BIT-IDENTICAL to the same call decoded on its own (oavsr_generate, itself pinned to the
oracle in tests/test_stages_gpu.py), whatever shares its decode steps, wherever its rows sit in a step, whenever it joined -- in f32
and in bf16."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd.config import tiny_spec  # noqa: E402
from omni_avsr_amd.modeling import AVSR_LLMs, OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402


def clips(B, seed):
    g = torch.Generator().manual_seed(seed)
    Tv = [int(x) for x in torch.randint(5, 13, (B,), generator=g)]
    Ts = [640 * t for t in Tv]
    video = torch.randn(B, 12, 1, 88, 88, generator=g)
    audio = torch.randn(B, 7680, 1, generator=g)
    for b in range(B):
        video[b, Tv[b]:] = 0
        audio[b, Ts[b]:] = 0
    return video, audio, Tv, Ts


def alone(eng, c, max_new, force_len, task="audiovisual"):
    video, audio, Tv, Ts = c
    ids, lens = eng.generate(task, video if task != "audio" else None, Tv if task != "audio" else None, audio if task != "video" else None,
                             Ts if task != "video" else None, 4, 2, beams=1, max_new=max_new, force_len=force_len)
    torch.cuda.synchronize()
    return ids.cpu().tolist(), lens.cpu().tolist()


def submit(eng, c, max_new, force_len, slot, stream, task="audiovisual"):
    video, audio, Tv, Ts = c
    with torch.cuda.stream(stream):
        return eng.submit(task, video if task != "audio" else None, Tv if task != "audio" else None, audio if task != "video" else None,
                          Ts if task != "video" else None, 4, 2, max_new=max_new, force_len=force_len, slot=slot)


def collect(eng, pend, stream):
    ids, lens, ticket = pend
    with torch.cuda.stream(stream):
        eng.wait(ticket)
    stream.synchronize()
    return ids.cpu().tolist(), lens.cpu().tolist()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("stride", [0, 1, 3])
def test_three_staggered_calls_equal_isolated_calls(dtype, stride):
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=1234))
    eng.sched_stride, eng.sched_lanes = stride, 1            # ONE step loop: the three calls must share its steps
    mn = spec.max_dec_tokens
    calls = [clips(8, 1), clips(3, 2), clips(8, 3)]
    ref = [alone(eng, c, mn, True) for c in calls]
    streams = [torch.cuda.Stream() for _ in range(3)]
    pend = []
    for i, c in enumerate(calls):                       # staggered: every call joins `stride` steps (or 1/k of a hypothesis) after the previous one
        pend.append(submit(eng, c, mn, True, i, streams[i]))
        time.sleep(0.02 * i)
    got = [collect(eng, pend[i], streams[i]) for i in (2, 0, 1)]          # results asked for out of order
    assert got[1] == ref[0] and got[2] == ref[1] and got[0] == ref[2], f"{dtype}: ids of calls that shared decode steps differ from isolated calls"
    st = eng.scheduler_stats()
    assert st["calls"] == 3 and st["rows_per_step"] > 8.0, f"the calls did not share steps: {st}"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_eos_lengths_and_different_max_new(dtype):
    """no EOS suppression: rows finish at their own EOS (pad afterwards, lens as HF greedy), calls with different max_new leave at different steps"""
    spec = tiny_spec(family="qwen2", head_dim=128)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=77))
    calls = [(clips(5, 11), 6), (clips(8, 12), spec.max_dec_tokens), (clips(2, 13), 1), (clips(7, 14), 3)]
    ref = [alone(eng, c, mn, False) for c, mn in calls]
    streams = [torch.cuda.Stream() for _ in range(2)]
    pend = [submit(eng, c, mn, False, i % 2, streams[i % 2]) for i, (c, mn) in enumerate(calls)]
    got = [collect(eng, pend[i], streams[i % 2]) for i in range(len(calls))]
    assert got == ref


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("family,hd", [("llama", 64), ("qwen2", 128)])
def test_64_row_steps_equal_isolated_calls(dtype, family, hd):
    """max_rows = 64: eleven calls of 3..12 rows through ONE lane whose steps carry up to 64 rows -- the decode-shaped GEMMs give a row the
    same bits at every row count, so every call must decode exactly the ids of the same call run alone; also a call of 39 rows on its own
    (generate, no scheduler)."""
    spec = tiny_spec(family=family, head_dim=hd, task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=1234))
    eng.sched_lanes, eng.sched_rows, eng.sched_slots, eng.sched_stride = 1, 64, 128, 1
    mn = spec.max_dec_tokens
    sizes = [8, 12, 8, 3, 8, 8, 12, 5, 8, 8, 7]
    calls = [clips(b, 300 + i) for i, b in enumerate(sizes)]
    ref = [alone(eng, c, mn, True) for c in calls]
    streams = [torch.cuda.Stream() for _ in range(3)]
    pend = [submit(eng, c, mn, True, i % 3, streams[i % 3]) for i, c in enumerate(calls)]
    eng.flush()
    got = [collect(eng, pend[i], streams[i % 3]) for i in range(len(calls))]
    for i in range(len(calls)):
        assert got[i] == ref[i], f"{dtype}: call {i} ({sizes[i]} rows) decodes differently inside 64-row steps than alone"
    st = eng.scheduler_stats()
    assert st["rows_per_step"] > 32.0, f"the steps never carried more than 32 rows: {st}"
    # 40 utterances in ONE plain call: rows 0..7 are the first call's utterances
    video = torch.cat([c[0] for c in calls[:5]], 0)
    audio = torch.cat([c[1] for c in calls[:5]], 0)
    Tv = sum((c[2] for c in calls[:5]), [])
    Ts = sum((c[3] for c in calls[:5]), [])
    ids40, lens40 = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 2, beams=1, max_new=mn, force_len=True)
    torch.cuda.synchronize()
    assert len(Tv) == 39 and ids40[:8].cpu().tolist() == ref[0][0], "39 rows in one decode loop: rows 0..7 differ from the 8-row call"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("B", [65, 100, 128])
def test_one_call_of_65_to_128_rows_through_the_scheduler(dtype, B):
    """the scheduler's default step holds 128 rows and `oavsr_generate_submit` accepts B <= max_rows: a single call that brings more than 64
    rows (round 4 refused it in admission: sched_init_slots checked B <= 64) decodes, and its first 8 rows are the ids of those 8 utterances
    submitted as a call of their own"""
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=1234))
    eng.sched_lanes, eng.sched_rows, eng.sched_slots = 1, 128, 256
    mn = spec.max_dec_tokens
    big = clips(B, 900 + B)
    head = tuple(x[:8] for x in big)
    ref8 = alone(eng, head, mn, True)
    s = torch.cuda.Stream()
    p = submit(eng, big, mn, True, 0, s)
    p8 = submit(eng, head, mn, True, 1, s)          # joins the same lane when B + 8 <= 128, else the lane makes room first
    ids, lens = collect(eng, p, s)
    assert len(ids) == B and all(n == mn for n in lens)
    assert (ids[:8], lens[:8]) == ref8, f"{dtype}: rows 0..7 of a {B}-row call differ from the 8-row call of the same utterances"
    assert collect(eng, p8, s) == ref8


def test_more_calls_than_rows_per_step_and_oracle_f32():
    """7 calls of 8 rows against a 32-row step: the host issues the steps that make room; ids equal the CPU oracle's (f32)"""
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    sd = synth_state_dict(spec, seed=1234)
    eng = OmniEngine(spec, dtype="f32").load_state_dict(sd)
    eng.sched_lanes, eng.sched_rows, eng.sched_slots = 1, 32, 64
    orc = O.OmniOracle(sd, spec.oracle_cfg())
    mn = spec.max_dec_tokens
    calls = [clips(8, 20 + i) for i in range(7)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    pend = [submit(eng, c, mn, False, i % 3, streams[i % 3]) for i, c in enumerate(calls)]
    eng.flush()
    got = [collect(eng, pend[i], streams[i % 3]) for i in range(7)]
    st = eng.scheduler_stats()
    assert st["rows_per_step"] > 16.0, st
    for i in (0, 3, 6):
        video, audio, Tv, Ts = calls[i]
        with torch.no_grad():
            want = orc.forward(dict(video=video, audio=audio, lengths=torch.tensor(Ts), video_lens=torch.tensor(Tv), tokens=torch.full((8, 1), spec.ids["bos"])),
                               "audiovisual", 4, 2, num_beams=1)
        ids, lens = got[i]
        assert [ids[b][: lens[b]] for b in range(8)] == want, f"call {i}: scheduler ids differ from the oracle"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_two_lanes_share_the_pool(dtype):
    """two step loops (own streams) over one pool of row slots: calls alternate between them; same ids as isolated calls"""
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=1234))
    eng.sched_lanes, eng.sched_rows, eng.sched_slots = 2, 32, 40             # a small pool: later calls wait for blocks of the earlier ones
    mn = spec.max_dec_tokens
    calls = [clips(8 if i % 3 else 6, 60 + i) for i in range(9)]
    ref = [alone(eng, c, mn, False) for c in calls]
    streams = [torch.cuda.Stream() for _ in range(3)]
    pend = [submit(eng, c, mn, False, i % 3, streams[i % 3]) for i, c in enumerate(calls)]
    got = [collect(eng, pend[i], streams[i % 3]) for i in range(9)]
    assert got == ref
    assert eng.scheduler_stats()["graphs"] >= 2


def test_task_switch_drains_the_loop_and_pool_growth():
    """task-specific LoRA: rows of one step share the adapter set -> a call of another task waits for the loop to drain; a longer prompt
    than the pool was built for rebuilds the pool (tickets of the old one stay valid)"""
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype="bf16").load_state_dict(synth_state_dict(spec, seed=5))
    mn = spec.max_dec_tokens
    a, v, av = clips(4, 31), clips(6, 32), clips(8, 33)
    ref = [alone(eng, a, mn, True, "audio"), alone(eng, v, mn, True, "video"), alone(eng, av, mn, True, "audiovisual")]
    s = torch.cuda.Stream()
    p0 = submit(eng, a, mn, True, 0, s, "audio")          # pool sized for the audio-only prompt
    p1 = submit(eng, v, mn, True, 0, s, "video")
    gen0 = eng._sched_gen
    p2 = submit(eng, av, mn, True, 0, s, "audiovisual")   # longer prompt: the pool grows
    assert eng._sched_gen >= gen0
    got = [collect(eng, p, s) for p in (p0, p1, p2)]
    assert got == ref


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_forward_sync_false_goes_through_the_scheduler(dtype):
    spec = tiny_spec(family="llama", task_specific=True, shared=True, matryoshka=True)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=1234))
    model = AVSR_LLMs.__new__(AVSR_LLMs)
    model.modality, model.max_dec_tokens, model.num_beams = "audiovisual", spec.max_dec_tokens, 1
    model.is_matryoshka, model.is_task_specific, model.spec, model.engine = True, True, spec, eng
    model.downsample_ratio_audio, model.downsample_ratio_video = spec.downsample_ratio_audio, spec.downsample_ratio_video
    batches = []
    for i in range(4):
        video, audio, Tv, Ts = clips(8 if i % 2 == 0 else 5, 40 + i)
        batches.append(dict(video=video, audio=audio, lengths=torch.tensor(Ts), video_lengths=torch.tensor(Tv)))
    want = [model(b, is_trainval=False, modality="audiovisual", test_ratio_matry_audio=4, test_ratio_matry_video=2).cpu() for b in batches]
    streams = [torch.cuda.Stream() for _ in range(2)]
    pend = []
    for i, b in enumerate(batches):
        with torch.cuda.stream(streams[i % 2]):
            pend.append(model(b, is_trainval=False, modality="audiovisual", test_ratio_matry_audio=4, test_ratio_matry_video=2, slot=i % 2, sync=False))
    assert model.engine.scheduler_stats()["calls"] == 4
    got = [p.result().cpu() for p in reversed(pend)][::-1]
    for w, g in zip(want, got):
        assert torch.equal(w, g)
    model.decode_scheduler = False                        # opt-out: the per-call decode loop, same ids
    with torch.cuda.stream(streams[0]):
        p = model(batches[0], is_trainval=False, modality="audiovisual", test_ratio_matry_audio=4, test_ratio_matry_video=2, slot=0, sync=False)
    assert torch.equal(p.result().cpu(), want[0]) and model.engine.scheduler_stats()["calls"] == 4


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_schedules_never_change_a_result(seed):
    """Seeded stress: 14 calls with random sizes (1..12 utterances), tasks, max_new, EOS suppression, pacing, lane count and pool size,
    submitted in bursts with results collected in random order and at random moments (some early, forcing steps; some only after the
    final flush) -- every call must return the ids of the same call decoded alone."""
    import random
    rng = random.Random(1000 + seed)
    spec = tiny_spec(family="llama" if seed != 1 else "qwen2", head_dim=64 if seed != 1 else 128, task_specific=(seed != 1), shared=(seed != 1))
    dtype = "bf16" if seed != 2 else "f32"
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=99 + seed))
    eng.sched_lanes, eng.sched_slots, eng.sched_rows = rng.choice([1, 2, 3]), rng.choice([32, 48, 96]), rng.choice([16, 24, 32])
    tasks = ["audiovisual", "audiovisual", "audiovisual", "audio", "video"]
    calls = []
    for i in range(14):
        B = rng.randint(1, min(12, eng.sched_rows))
        calls.append(dict(c=clips(B, 500 + 31 * seed + i), mn=rng.choice([1, 2, 5, spec.max_dec_tokens, spec.max_dec_tokens]), force=rng.random() < 0.5,
                          task=rng.choice(tasks) if spec.is_task_specific else "audiovisual"))
    ref = [alone(eng, k["c"], k["mn"], k["force"], k["task"]) for k in calls]
    streams = [torch.cuda.Stream() for _ in range(3)]
    pend, got = {}, {}
    for i, k in enumerate(calls):
        eng.sched_stride = rng.choice([0, 0, 1, 2, 5])
        pend[i] = (submit(eng, k["c"], k["mn"], k["force"], i % 3, streams[i % 3], k["task"]), streams[i % 3])
        if rng.random() < 0.3:                                 # collect something now, out of order
            j = rng.choice(sorted(pend))
            p, s = pend.pop(j)
            got[j] = collect(eng, p, s)
        if rng.random() < 0.2:
            time.sleep(0.01)
    eng.flush()
    for j in rng.sample(sorted(pend), len(pend)):
        p, s = pend.pop(j)
        got[j] = collect(eng, p, s)
    for i in range(len(calls)):
        assert got[i] == ref[i], f"seed {seed}: call {i} ({calls[i]['task']}, B={len(calls[i]['c'][2])}, max_new={calls[i]['mn']}) differs from the isolated call"


def test_a_refused_call_leaves_the_scheduler_consistent():
    """a call the engine refuses half way (unknown compression rate: KeyError like the reference's matry_map lookup, modeling_OmniAVSR.py:429) has
    taken no slots and joined no lane; calls before and after it decode as if it had never been submitted"""
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype="bf16").load_state_dict(synth_state_dict(spec, seed=1234))
    eng.sched_lanes = 1
    mn = spec.max_dec_tokens
    a, b = clips(8, 71), clips(6, 72)
    ref = [alone(eng, a, mn, True), alone(eng, b, mn, True)]
    s = torch.cuda.Stream()
    p0 = submit(eng, a, mn, True, 0, s)
    video, audio, Tv, Ts = clips(4, 73)
    with pytest.raises(KeyError):
        with torch.cuda.stream(s):
            eng.submit("audiovisual", video, Tv, audio, Ts, 3, 2, max_new=mn, force_len=True, slot=0)       # rate 3 is not configured
    p1 = submit(eng, b, mn, True, 0, s)
    assert eng.scheduler_stats()["calls"] == 2
    assert [collect(eng, p0, s), collect(eng, p1, s)] == ref


def test_poll_never_forces_and_pinned_host_results_and_trace():
    """oavsr_generate_poll: 0 while steps are still to be issued (it issues none), 1 once later submissions / a flush have completed the call --
    and then the PINNED host buffers the scheduler's stream filled hold exactly the device results.  oavsr_scheduler_trace: one record per
    call phase and per burst of steps, times ordered and non-negative."""
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype="bf16").load_state_dict(synth_state_dict(spec, seed=1234))
    eng.sched_lanes = 1
    mn = spec.max_dec_tokens
    s = torch.cuda.Stream()
    c0 = clips(8, 81)
    p0 = submit(eng, c0, mn, True, 0, s)
    torch.cuda.synchronize()
    assert eng.poll(p0[2]) is False, "a call alone in its lane cannot be complete: nobody has issued its steps yet"
    steps_before = eng.scheduler_stats()["steps"]
    assert eng.poll(p0[2]) is False and eng.scheduler_stats()["steps"] == steps_before, "poll must not issue steps"
    eng.scheduler_trace(True)
    pend = [submit(eng, clips(8, 82 + i), mn, True, 0, s) for i in range(4)]        # their admissions issue p0's remaining steps
    eng.flush()
    torch.cuda.synchronize()
    assert eng.poll(p0[2]) is True
    host = eng.host_result(p0[2])
    ids, lens = p0[0].cpu(), p0[1].cpu()
    assert torch.equal(host[: 8 * mn].view(8, mn), ids) and torch.equal(host[8 * mn:], lens)
    assert (ids.tolist(), lens.tolist()) == alone(eng, c0, mn, True)
    recs = eng.scheduler_trace_read()
    eng.scheduler_trace(False)
    phases, bursts = [r for r in recs if r["kind"] == 0], [r for r in recs if r["kind"] == 1]
    assert len(phases) == 4 and len(bursts) >= 2 and sum(r["steps"] for r in bursts) >= mn - 1
    assert all(0 <= r["t0"] <= r["t1"] for r in recs), recs
    for p in pend:
        collect(eng, p, s)
