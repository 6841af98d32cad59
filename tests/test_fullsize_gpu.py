"""-m gpu: parity at BASELINE.json's full model sizes (AV-HuBERT-Large + Whisper-medium + Llama-3.2-1B, Omni-LoRA ST).
 * f32 parity mode against the CPU oracle on a 4 s clip: encoder activations within the north-star 1e-3 (relative to the
   activation scale) and the 32 greedy token ids bit-exact;
 * size-independent properties at 16 s / batch sizes the oracle cannot afford: an utterance decodes to the same ids and the
   same encoder activations (bit for bit) alone and inside a ragged batch (no cross-utterance leakage, the reference's
   batch-1 semantics), and repeated runs are deterministic."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd.config import spec_from_names  # noqa: E402
from omni_avsr_amd.modeling import OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402


def full_spec():
    return spec_from_names("meta-llama/Llama-3.2-1B", "openai/whisper-medium.en", "large_vox_iter5", use_lora_avhubert=True,
                           intermediate_size=2048, compression_mode="avg-pooling", downsample_ratio_audio=[4, 16], downsample_ratio_video=[2, 4, 5],
                           is_matryoshka=True, lora_rank=32, lora_alpha=4, is_task_specific=True, shared_lora=True,
                           prompt_len={"audio": 8, "video": 8, "audiovisual": 8}, max_dec_tokens=32, num_beams=1)


def clips(Tv, seed=0):
    g = torch.Generator().manual_seed(seed)
    B = len(Tv)
    video = torch.randn(B, max(Tv), 1, 88, 88, generator=g)
    audio = torch.zeros(B, 640 * max(Tv), 1)
    for b, t in enumerate(Tv):
        video[b, t:] = 0
        audio[b, : 640 * t, 0] = torch.nn.functional.layer_norm(torch.randn(640 * t, generator=g), (640 * t,))
    return video, audio, [640 * t for t in Tv]


@pytest.fixture(scope="module")
def bf16_engine():
    """ONE full-size bf16 engine for the cases of this file (1.9 B parameters of seeded weights: building it per test was a third of
    the file's run time).  Tests that change the scheduler's shape restore the defaults."""
    spec = full_spec()
    sd = synth_state_dict(spec, seed=1234, device="cuda:0", dtype=torch.bfloat16)
    eng = OmniEngine(spec, dtype="bf16").load_state_dict(sd)
    del sd
    torch.cuda.empty_cache()
    yield eng
    eng.close()


def test_fullsize_f32_matches_oracle():
    spec = full_spec()
    sd = synth_state_dict(spec, seed=1234, device="cuda:0")
    eng = OmniEngine(spec, dtype="f32").load_state_dict(sd)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    Tv = [100]
    video, audio, Ts = clips(Tv, seed=5)
    _, venc, _ = eng.encode_video(video, Tv, 4, return_enc=True)
    _, aenc, _, _ = eng.encode_audio(audio, Ts, 4, return_enc=True)
    ids, lens = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 4, beams=1, max_new=32, force_len=True)
    orc = O.OmniOracle(sd_cpu, spec.oracle_cfg())
    with torch.no_grad():
        _, vs = orc.encode_video(video, 4, return_stages=True)
        _, as_ = orc.encode_audio(audio[:, : Ts[0]], Ts[0], 4, return_stages=True)
        ref = orc.forward(dict(video=video, audio=audio, lengths=torch.tensor(Ts), tokens=torch.full((1, 1), spec.ids["bos"])), "audiovisual", 4, 4,
                          num_beams=1, force_len=True)
    ev = (venc.float().cpu() - vs["enc"][0]).abs().max().item() / vs["enc"][0].abs().max().item()
    ea = (aenc[0].float().cpu() - as_["enc"][0]).abs().max().item() / as_["enc"][0].abs().max().item()
    print(f"full-size f32: video encoder rel err {ev:.2e}, whisper encoder rel err {ea:.2e}")
    assert ev < 1e-3 and ea < 1e-3
    assert ids[0].cpu().tolist() == ref[0]


def test_fullsize_batch_invariance_and_determinism(bf16_engine):
    eng = bf16_engine
    Tv = [400, 250, 100, 61]
    video, audio, Ts = clips(Tv, seed=9)
    ids_b, lens_b = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 4, beams=1, max_new=32, force_len=True)
    ids_b2, _ = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 4, beams=1, max_new=32, force_len=True)
    assert torch.equal(ids_b, ids_b2), "two runs of the same batch differ"
    _, venc_b, _ = eng.encode_video(video, Tv, 4, return_enc=True)
    off = 0
    for b, t in enumerate(Tv):
        v1, a1 = video[b:b + 1, :t], audio[b:b + 1, : 640 * t]
        ids1, _ = eng.generate("audiovisual", v1, [t], a1, [640 * t], 4, 4, beams=1, max_new=32, force_len=True)
        assert ids1[0].tolist() == ids_b[b].tolist(), f"utterance {b} (T_v={t}) decodes differently alone vs in the ragged batch"
        _, venc1, _ = eng.encode_video(v1, [t], 4, return_enc=True)
        assert torch.equal(venc1, venc_b[off:off + t]), f"utterance {b}: video encoder output depends on its batch mates"
        off += t
    # ... and inside a batch of 12 (the decode-shaped GEMMs give a row the same bits whatever shares the launch)
    Tv12 = Tv + [61, 40, 33, 61, 50, 61, 27, 61]
    video12, audio12, Ts12 = clips(Tv12, seed=9)
    video12[:4, : video.shape[1]], audio12[:4, : audio.shape[1]] = video, audio
    ids12, _ = eng.generate("audiovisual", video12, Tv12, audio12, Ts12, 4, 4, beams=1, max_new=32, force_len=True)
    for b in range(4):
        assert ids12[b].tolist() == ids_b[b].tolist(), f"utterance {b} decodes differently inside a batch of 12"


def test_fullsize_launch_objective_changes_no_result(bf16_engine):
    """oavsr_set_launch_objective (harness.evaluate sets it: several batches in flight vs one call at a time) moves launches between tile
    shapes -- incl. prefill GEMMs that lose their fused SwiGLU epilogue under the other rule (a ragged batch: 2 x 16 s, a short batch: 2 short
    clips, one utterance) -- and never a bit: same ids, same video encoder output."""
    eng = bf16_engine
    for Tv in ([400, 400], [100, 61], [250], [400, 250, 100, 61, 61, 40, 33, 27]):
        video, audio, Ts = clips(Tv, seed=21)
        got = {}
        try:
            for alone in (False, True):
                eng.launch_objective(alone)
                ids, _ = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 4, beams=1, max_new=16, force_len=True)
                _, venc, _ = eng.encode_video(video, Tv, 4, return_enc=True)
                got[alone] = (ids.clone(), venc.clone())
        finally:
            eng.launch_objective(False)
        assert torch.equal(got[False][0], got[True][0]), f"T_v={Tv}: ids depend on the launch objective"
        assert torch.equal(got[False][1], got[True][1]), f"T_v={Tv}: video encoder output depends on the launch objective"


def test_fullsize_encoders_side_by_side_change_no_bit(bf16_engine):
    """One call at a time (the latency objective), an audio-visual call runs its video encoder on a second stream beside the audio
    encoder, each on its own scratch set (model.hip: encode_prefill).  Same launches: the ids and the digests of inputs_embeds / first-step
    logits are those of the one-stream order (OAVSR_ENC_SIDE_BY_SIDE=0) and of the pipelined objective, call after call."""
    import os
    eng = bf16_engine
    for Tv in ([250], [400, 61], [100, 400, 33, 250, 61, 61, 400, 27]):
        video, audio, Ts = clips(Tv, seed=33)
        B = len(Tv)

        def run():
            dg = torch.zeros(B, 2, dtype=torch.int64, device="cuda:0")
            ids, _ = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 4, beams=1, max_new=8, force_len=True, digests=dg)
            torch.cuda.synchronize()
            return ids.clone(), dg.clone()

        ref = run()      # calls share the chip: one stream
        prev = eng.launch_objective(True)
        try:
            os.environ["OAVSR_ENC_SIDE_BY_SIDE"] = "0"
            one = run()
            os.environ["OAVSR_ENC_SIDE_BY_SIDE"] = "1"
            for rep in range(4):
                two = run()
                assert torch.equal(two[0], one[0]) and torch.equal(two[1], one[1]), f"T_v={Tv} run {rep}: side-by-side encoders changed a result"
        finally:
            os.environ.pop("OAVSR_ENC_SIDE_BY_SIDE", None)
            eng.launch_objective(prev)
        assert torch.equal(ref[0], one[0]) and torch.equal(ref[1], one[1]), f"T_v={Tv}: results depend on the launch objective"
        assert (ref[1] != 0).all()


def test_fullsize_chained_decode_launches_change_no_id(bf16_engine):
    """<= 16 decode rows: the strip GEMMs between two attention launches go out as the phases of one launch (model.hip: chain_flush;
    OAVSR_DECODE_CHAIN=0: one by one).  Same ids from the captured graph and from eager launches, greedy (1 and 8 rows) and beam search
    (15 rows).  (Against the tiled kernels of larger steps: the batch / scheduler invariance suites.)"""
    import os
    eng = bf16_engine
    Tv = [100, 61, 250, 40, 33, 61, 50, 27]
    video, audio, Ts = clips(Tv, seed=44)
    got = {}
    try:
        for chain in ("0", "1"):
            os.environ["OAVSR_DECODE_CHAIN"] = chain
            for graph in ("1", "0"):
                os.environ["OAVSR_DECODE_GRAPH"] = graph
                a, _ = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 4, beams=1, max_new=24, force_len=True)
                b, _ = eng.generate("audiovisual", video[:1, :100], Tv[:1], audio[:1, :64000], Ts[:1], 4, 4, beams=1, max_new=24, force_len=True)
                c, lc = eng.generate("audiovisual", video[:1, :100], Tv[:1], audio[:1, :64000], Ts[:1], 4, 4, beams=15, max_new=24)
                got[(chain, graph)] = (a.clone(), b.clone(), c.clone(), lc.clone())
    finally:
        os.environ.pop("OAVSR_DECODE_CHAIN", None)
        os.environ.pop("OAVSR_DECODE_GRAPH", None)
    ref = got[("0", "1")]
    for k, v in got.items():
        for what, x, y in zip(("greedy, 8 rows", "greedy, 1 row", "beam 15", "beam 15 lengths"), v, ref):
            assert torch.equal(x, y), f"OAVSR_DECODE_CHAIN={k[0]} OAVSR_DECODE_GRAPH={k[1]}: {what} differs"
    assert torch.equal(ref[0][0], ref[1][0]), "an utterance decodes differently alone and in its batch"


def test_fullsize_grouped_decode_equals_separate_calls(bf16_engine):
    """harness decode_group / bench --decode-group: 24 utterances in one call (encoders / prefill in chunks of 8, ONE decode loop over 24 rows)
    decode the ids of three separate calls of 8."""
    eng = bf16_engine
    g = torch.Generator().manual_seed(21)
    Tv = [int(x) for x in torch.randint(30, 101, (24,), generator=g)]
    video, audio, Ts = clips(Tv, seed=22)
    ids_g, _ = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 4, beams=1, max_new=32, force_len=True, chunk=8)
    ids_g = ids_g.cpu()
    for c in range(3):
        sl = slice(8 * c, 8 * c + 8)
        tmax = max(Tv[sl])
        ids_c, _ = eng.generate("audiovisual", video[sl, :tmax], Tv[sl], audio[sl, : 640 * tmax], Ts[sl], 4, 4, beams=1, max_new=32, force_len=True)
        assert torch.equal(ids_c.cpu(), ids_g[sl]), f"utterances {8 * c}..{8 * c + 7}: grouped decode differs from the separate call"


def test_fullsize_scheduler_equals_separate_calls(bf16_engine):
    """The headline configuration's own mechanism at the full model size: three calls of 8 utterances (ragged lengths) submitted one after another
    through the decode scheduler -- encoders / prefill per call on three streams, decode rows sharing the steps of one lane -- decode the
    ids of the same calls run on their own (bf16, EOS suppressed and free-running)."""
    eng = bf16_engine
    lanes0 = eng.sched_lanes
    g = torch.Generator().manual_seed(31)
    calls = []
    for c in range(3):
        Tv = [int(x) for x in torch.randint(30, 101, (8,), generator=g)]
        video, audio, Ts = clips(Tv, seed=40 + c)
        calls.append((video, audio, Tv, Ts))
    for force in (True, False):
        ref = []
        for video, audio, Tv, Ts in calls:
            ids, lens = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 4, beams=1, max_new=32, force_len=force)
            ref.append((ids.cpu(), lens.cpu()))
        eng.sched_lanes = 1
        streams = [torch.cuda.Stream() for _ in range(3)]
        pend = []
        for i, (video, audio, Tv, Ts) in enumerate(calls):
            with torch.cuda.stream(streams[i]):
                pend.append(eng.submit("audiovisual", video, Tv, audio, Ts, 4, 4, max_new=32, force_len=force, slot=i))
        eng.flush()
        torch.cuda.synchronize()
        for i, (ids, lens, _) in enumerate(pend):
            assert torch.equal(lens.cpu(), ref[i][1]) and torch.equal(ids.cpu(), ref[i][0]), f"call {i} (force_len={force}) decodes differently inside the shared decode loop"
    assert eng.scheduler_stats()["rows_per_step"] > 12
    eng.sched_lanes = lanes0


def test_headline_configuration_returns_the_ids_of_isolated_calls(bf16_engine):
    """bench.py's headline configuration as a test (BASELINE.json configs[1]): batches of 8 x 16 s clips, 32 greedy tokens with EOS
    suppressed, the decode scheduler at its DEFAULT shape (3 lanes x 128 rows, 512 slots), calls submitted back to back on three of the
    engine's caller streams, four input sets rotating -- 28 calls, and every one of them must return exactly the ids of ONE isolated
    oavsr_generate of its input set (what bench.py itself verifies after its timed region before it prints a line)."""
    eng = bf16_engine
    assert (eng.sched_lanes, eng.sched_rows, eng.sched_slots) == (3, 128, 1024), "this case pins the engine's default scheduler shape"
    B, Tv = 8, 400
    sets = []
    for s_ in range(4):
        g = torch.Generator(device="cuda:0").manual_seed(100 + s_)
        video = torch.randn(B * Tv, 88, 88, generator=g, device="cuda:0")
        audio = torch.nn.functional.layer_norm(torch.randn(B, 640 * Tv, generator=g, device="cuda:0"), (640 * Tv,))
        sets.append((video, audio))
    Tvs, Ts = [Tv] * B, [640 * Tv] * B
    iso = []
    for video, audio in sets:
        dg = torch.zeros(B, 2, dtype=torch.int64, device="cuda:0")
        ids, lens = eng.generate("audiovisual", video, Tvs, audio, Ts, 4, 4, beams=1, max_new=32, force_len=True, video_is_flat=True, digests=dg)
        torch.cuda.synchronize()
        iso.append((ids.clone(), lens.clone(), dg.clone()))
    # With seeded random weights the 32 forced tokens hardly depend on the clip (the residual stream at the last prompt position is dominated
    # by the prompt's own embedding), so the ids alone guard state handling only.  The DIGESTS are input-sensitive: every utterance of every
    # input set has its own {inputs_embeds, first-step logits} pair, and a call that received another call's encoder output, prompt rows,
    # KV slots or logits row reports a different pair (test_input_sensitive_weights_under_the_scheduler adds weights whose ids depend on the clip).
    alld = torch.cat([d for _, _, d in iso])
    assert len({int(x) for x in alld[:, 0]}) == 4 * B and len({int(x) for x in alld[:, 1]}) == 4 * B, "digests of different clips collide"
    eng.reserve_scheduler("audiovisual", Tv, 640 * Tv, 4, 4, 32)
    streams = [eng.caller_stream(i) for i in range(3)]
    pend = []
    dgs = torch.zeros(28, B, 2, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()              # (the zero fill ran on torch's default stream: done before a caller stream writes there)
    for i in range(28):
        video, audio = sets[i % 4]
        dg = dgs[i]
        with torch.cuda.stream(streams[i % 3]):
            pend.append((i, eng.submit("audiovisual", video, Tvs, audio, Ts, 4, 4, max_new=32, force_len=True, video_is_flat=True, slot=i % 3, digests=dg), dg))
    eng.flush()
    torch.cuda.synchronize()
    for i, (ids, lens, _), dg in pend:
        assert torch.equal(lens, iso[i % 4][1]) and torch.equal(ids, iso[i % 4][0]), f"call {i} (input set {i % 4}) differs from the isolated call of that set"
        assert torch.equal(dg, iso[i % 4][2]), f"call {i} (input set {i % 4}): digests differ from the isolated call's -- the call did not see its own inputs"
    st = eng.scheduler_stats()
    assert st["rows_per_step"] > 20, st


def test_input_sensitive_weights_under_the_scheduler():
    """The same pipelined schedule with weights that make the IDS depend on the clip (recipe found with tools/probe/sensitivity_probe.py): token
    and prompt embeddings scaled down (x 0.05), the LLM's attention output projections up (x 8) and its query projections up (x 6: peaked
    attention), so the residual stream at a decode position is dominated by what attention picks from the ~300 audio / video tokens instead
    of by the token's own embedding.  Asserted first: most utterances decode to a hypothesis of their own and the four input sets differ.
    Then 16 calls through the decode scheduler must return exactly the ids -- and digests -- of the isolated call of their input set: a call
    that got another call's encoder output, KV rows or slots fails on ids here."""
    spec = full_spec()
    sd = synth_state_dict(spec, seed=4321, device="cuda:0", dtype=torch.bfloat16)
    for k in list(sd):
        if k.endswith("embed_tokens.weight") or k.startswith("prompt_"):
            sd[k] = sd[k] * 0.05            # (a new tensor: lm_head keeps the unscaled copy)
        elif k.startswith("llm.") and k.endswith("self_attn.o_proj.weight"):
            sd[k] = sd[k] * 8
        elif k.startswith("llm.") and k.endswith("self_attn.q_proj.weight"):
            sd[k] = sd[k] * 6
    eng = OmniEngine(spec, dtype="bf16").load_state_dict(sd)
    del sd
    torch.cuda.empty_cache()
    B, Tv = 8, 400
    Tvs, Ts = [Tv] * B, [640 * Tv] * B
    sets, iso = [], []
    for s_ in range(4):
        g = torch.Generator(device="cuda:0").manual_seed(300 + s_)
        video = torch.randn(B * Tv, 88, 88, generator=g, device="cuda:0")
        audio = torch.nn.functional.layer_norm(torch.randn(B, 640 * Tv, generator=g, device="cuda:0"), (640 * Tv,))
        sets.append((video, audio))
        dg = torch.zeros(B, 2, dtype=torch.int64, device="cuda:0")
        ids, lens = eng.generate("audiovisual", video, Tvs, audio, Ts, 4, 4, beams=1, max_new=32, force_len=True, video_is_flat=True, digests=dg)
        torch.cuda.synchronize()
        iso.append((ids.clone(), lens.clone(), dg.clone()))
    rows = {tuple(r) for ids, _, _ in iso for r in ids.cpu().tolist()}
    assert len(rows) >= 2 * B, f"the ids of these weights do not depend on the clip: only {len(rows)} distinct hypotheses among {4 * B} utterances"
    assert len({tuple(ids.flatten().tolist()) for ids, _, _ in iso}) == 4
    eng.reserve_scheduler("audiovisual", Tv, 640 * Tv, 4, 4, 32)
    streams = [eng.caller_stream(i) for i in range(3)]
    pend = []
    dgs = torch.zeros(16, B, 2, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    for i in range(16):
        video, audio = sets[(3 * i) % 4]
        dg = dgs[i]
        with torch.cuda.stream(streams[i % 3]):
            pend.append(((3 * i) % 4, eng.submit("audiovisual", video, Tvs, audio, Ts, 4, 4, max_new=32, force_len=True, video_is_flat=True, slot=i % 3, digests=dg), dg))
    eng.flush()
    torch.cuda.synchronize()
    for k, (ids, lens, _), dg in pend:
        assert torch.equal(ids, iso[k][0]) and torch.equal(lens, iso[k][1]) and torch.equal(dg, iso[k][2]), f"a scheduled call of input set {k} differs from the isolated call"
    eng.close()
