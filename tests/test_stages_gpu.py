"""-m gpu: each stage of the hot path (through the C ABI) against the CPU oracle on the same seeded inputs and
weights.  f32 mode must hold the north-star tolerance (encoder activations within 1e-3, token ids bit-exact);
bf16 mode (the throughput mode, = the reference's bf16-true) is checked against a looser bound."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd.config import tiny_spec  # noqa: E402
from omni_avsr_amd.modeling import AVSR_LLMs, OmniEngine, whisper_keep_tokens  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

ENC_TOL = {"f32": 1e-3, "bf16": 0.12}


def make(dtype, **kw):
    spec = tiny_spec(**kw)
    sd = synth_state_dict(spec, seed=1234)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(sd)
    orc = O.OmniOracle(sd, spec.oracle_cfg())
    return spec, sd, eng, orc


def inputs(B=2, Tv=(12, 9), Ts=(7680, 5000), seed=0):
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(B, max(Tv), 1, 88, 88, generator=g)
    audio = torch.randn(B, max(Ts), 1, generator=g)
    for b in range(B):
        video[b, Tv[b]:] = 0
        audio[b, Ts[b]:] = 0
    return video, audio


def maxerr(a, b):
    return (a.float().cpu() - b.float()).abs().max().item()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("use_lora", [True, False])
def test_encode_video(dtype, use_lora):
    spec, sd, eng, orc = make(dtype, use_lora_avhubert=use_lora)
    Tv = [12, 9]
    video, _ = inputs(Tv=Tv)
    out, enc, n_out = eng.encode_video(video, Tv, rate=2, return_enc=True)
    refs_enc, refs_out = [], []
    for b in range(2):
        v = video[b:b + 1, :Tv[b]]
        o, st = orc.encode_video(v, 2, return_stages=True)
        refs_enc.append(st["enc"][0])
        refs_out.append(o[0])
    e1 = maxerr(enc, torch.cat(refs_enc))
    e2 = maxerr(out, torch.cat(refs_out))
    assert n_out == [6, 4]
    assert e1 < ENC_TOL[dtype] and e2 < ENC_TOL[dtype], f"video encoder max err {e1:.3e} / compressed {e2:.3e}"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_encode_audio(dtype):
    spec, sd, eng, orc = make(dtype)
    Ts = [7680, 5000]
    _, audio = inputs(Ts=Ts)
    out, enc, mel, n_out = eng.encode_audio(audio, Ts, rate=4, return_enc=True)
    refs_out = []
    for b in range(2):
        a = audio[b:b + 1, :Ts[b]]
        if dtype == "bf16":      # bf16 mode = the reference's bf16-true run, where Lightning hands forward() the WAVEFORM in bf16:
            a = a.bfloat16().float()                       # same input for the fp32 oracle, so that the bound below is about arithmetic
        o, st = orc.encode_audio(a, Ts[b], 4, return_stages=True)
        em = maxerr(mel[b].transpose(0, 1), st["mel"][0])
        ee = maxerr(enc[b], st["enc"][0])
        assert em < (2e-3 if dtype == "f32" else 3e-2), f"log-mel max err {em:.3e}"
        assert ee < ENC_TOL[dtype], f"whisper encoder max err {ee:.3e}"
        refs_out.append(o[0])
    assert n_out == [whisper_keep_tokens(t) // 4 for t in Ts]
    e2 = maxerr(out, torch.cat(refs_out))
    assert e2 < ENC_TOL[dtype], f"compressed audio max err {e2:.3e}"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("family", ["llama", "qwen2"])
def test_prompt_prefill_logits(dtype, family):
    spec, sd, eng, orc = make(dtype, family=family, task_specific=(family == "llama"), shared=(family == "llama"))
    Tv, Ts = [12, 9], [7680, 5000]
    video, audio = inputs(Tv=Tv, Ts=Ts)
    eng.ensure_workspace(2, 12, 8000)
    a_out, na = eng.encode_audio(audio, Ts, 4)
    v_out, nv = eng.encode_video(video, Tv, 2)
    atok = eng.project("audio", 4, a_out)
    vtok = eng.project("video", 2, v_out)
    emb, S = eng.build_prompt("audiovisual", 2, na, atok, nv, vtok)
    logits = eng.prefill("audiovisual", emb, S)
    tokens = torch.full((1, 1), spec.ids["bos"]) if family == "llama" else torch.zeros(1, 0, dtype=torch.long)
    off = 0
    for b in range(2):
        one = dict(video=video[b:b + 1, :Tv[b]], audio=audio[b:b + 1, :Ts[b]], lengths=torch.tensor([Ts[b]]), tokens=tokens)
        orc_mod = orc
        ref_emb = orc.prepare_inputs(one, "audiovisual", 4, 2)
        assert ref_emb.shape[1] == S[b]
        ee = maxerr(emb[off:off + S[b]], ref_emb[0])
        assert ee < ENC_TOL[dtype], f"inputs_embeds max err {ee:.3e}"
        mod = "audiovisual" if spec.is_task_specific else None
        ref_logits = orc_mod.llm.forward(ref_emb, 0, [None] * spec.llm.layers, mod, last_only=True)[0, -1]
        el = maxerr(logits[b], ref_logits)
        assert el < (2e-3 if dtype == "f32" else 0.25), f"prefill logits max err {el:.3e}"
        if dtype == "f32":
            assert int(logits[b].argmax()) == int(ref_logits.argmax())
        off += S[b]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("task,chunk", [("audiovisual", 2), ("audiovisual", 3), ("video", 1), ("audio", 2)])
def test_generate_chunked_equals_plain(dtype, task, chunk):
    """oavsr_generate_chunked (encoders / prefill per chunk, one decode loop for all rows) returns exactly what generate does on the
    whole batch and what per-chunk generate calls do: ragged lengths, a ragged last chunk, natural EOS and forced length."""
    spec, sd, eng, orc = make(dtype, family="llama", task_specific=True, shared=True)
    Tv, Ts = [12, 9, 7, 12, 5], [7680, 5000, 4480, 7000, 3200]
    video, audio = inputs(B=5, Tv=Tv, Ts=Ts, seed=3)
    for force in (False, True):
        kw = dict(beams=1, max_new=spec.max_dec_tokens, force_len=force)
        ids0, len0 = eng.generate(task, video, Tv, audio, Ts, 4, 2, **kw)
        ids1, len1 = eng.generate(task, video, Tv, audio, Ts, 4, 2, chunk=chunk, **kw)
        assert torch.equal(ids0.cpu(), ids1.cpu()) and torch.equal(len0.cpu(), len1.cpu()), f"chunk={chunk} force={force}"
        for b0 in range(0, 5, chunk):
            sl = slice(b0, min(5, b0 + chunk))
            ids2, len2 = eng.generate(task, video[sl], Tv[sl], audio[sl], Ts[sl], 4, 2, **kw)
            assert torch.equal(ids2.cpu(), ids0[sl].cpu()) and torch.equal(len2.cpu(), len0[sl].cpu()), f"separate calls differ at {b0}"


@pytest.mark.parametrize("family,task", [("llama", "audiovisual"), ("llama", "audio"), ("llama", "video"), ("qwen2", "audiovisual")])
def test_generate_token_ids_f32(family, task):
    """Integer token ids bit-exact against the oracle's greedy decode (parity mode), natural EOS and forced length."""
    spec, sd, eng, orc = make("f32", family=family, task_specific=True, shared=(family == "llama"))
    Tv, Ts = [12, 9], [7680, 5000]
    video, audio = inputs(Tv=Tv, Ts=Ts)
    ra, rv = 4, 2
    for force in (False, True):
        ids, lens = eng.generate(task, video, Tv, audio, Ts, ra, rv, beams=1, max_new=spec.max_dec_tokens, force_len=force)
        ids, lens = ids.cpu(), lens.cpu()
        tokens = torch.full((2, 1), spec.ids["bos"]) if family == "llama" else torch.zeros(2, 0, dtype=torch.long)
        ref = orc.forward(dict(video=video, audio=audio, lengths=torch.tensor(Ts), video_lens=torch.tensor(Tv), tokens=tokens), task, ra, rv,
                          num_beams=1, force_len=force)
        for b in range(2):
            got = ids[b, : int(lens[b])].tolist()
            assert got == ref[b], f"utt {b} force={force}: got {got} want {ref[b]}"
            assert all(int(t) == spec.ids["pad"] for t in ids[b, int(lens[b]):])


@pytest.mark.parametrize("family,task,K", [("llama", "audiovisual", 3), ("llama", "audiovisual", 15), ("llama", "video", 15), ("qwen2", "audio", 4)])
def test_generate_beam_search_token_ids_f32(family, task, K):
    """HF-equivalent beam search on device (the reference's default decode, num_beams=15): token ids vs the oracle's
    restatement of transformers' BeamSearchScorer (itself pinned to transformers.generate by tests/golden/llm_*.npz)."""
    spec, sd, eng, orc = make("f32", family=family, task_specific=True, shared=(family == "llama"))
    Tv, Ts = [12, 9], [7680, 5000]
    video, audio = inputs(Tv=Tv, Ts=Ts)
    ids, lens = eng.generate(task, video, Tv, audio, Ts, 4, 2, beams=K, max_new=spec.max_dec_tokens)
    ids, lens = ids.cpu(), lens.cpu()
    tokens = torch.full((2, 1), spec.ids["bos"]) if family == "llama" else torch.zeros(2, 0, dtype=torch.long)
    ref = orc.forward(dict(video=video, audio=audio, lengths=torch.tensor(Ts), video_lens=torch.tensor(Tv), tokens=tokens), task, 4, 2, num_beams=K)
    for b in range(2):
        got = ids[b, : int(lens[b])].tolist()
        assert got == ref[b], f"utt {b} K={K}: got {got} want {ref[b]}"


def test_beam_search_eos_paths_f32():
    """Declare a token the model likes to emit to be EOS so hypotheses finish early (add / is_done / finalize paths)."""
    spec, sd, eng, orc = make("f32", family="llama", task_specific=True, shared=True)
    Tv, Ts = [12, 9], [7680, 5000]
    video, audio = inputs(Tv=Tv, Ts=Ts)
    tokens = torch.full((2, 1), spec.ids["bos"])
    batch = dict(video=video, audio=audio, lengths=torch.tensor(Ts), video_lens=torch.tensor(Tv), tokens=tokens)
    greedy = orc.forward(dict(batch), "audiovisual", 4, 2, num_beams=1, force_len=True)
    for alt in sorted(set(greedy[0][1:4] + greedy[1][1:4]))[:3]:
        spec2, sd2, eng2, orc2 = make("f32", family="llama", task_specific=True, shared=True)
        spec2.ids["eos"] = alt
        eng2 = OmniEngine(spec2, dtype="f32").load_state_dict(sd2)
        orc2 = O.OmniOracle(sd2, spec2.oracle_cfg())
        for K in (3, 15):
            ids, lens = eng2.generate("audiovisual", video, Tv, audio, Ts, 4, 2, beams=K, max_new=spec2.max_dec_tokens)
            ids, lens = ids.cpu(), lens.cpu()
            ref = orc2.forward(dict(batch), "audiovisual", 4, 2, num_beams=K)
            for b in range(2):
                assert ids[b, : int(lens[b])].tolist() == ref[b], f"eos={alt} K={K} utt {b}: {ids[b].tolist()} vs {ref[b]}"


@pytest.mark.parametrize("kw,ra,rv", [
    (dict(compression_mode="stack"), 4, 2),                                             # frame stacking, projector in = D*rate (:507-514)
    (dict(compression_mode="stack", matryoshka=True), 2, 1),                            # per-rate stack projectors, audio LN polarity quirk (:74-77)
    (dict(matryoshka=True, single_projector=True), 4, 2),                               # one projector + LayerNorm for every rate (:102,:186)
    (dict(matryoshka=True), 2, 1),                                                      # per-rate projectors, LayerNorm swallowed (:104,:188), rate 1
    (dict(remove_layernorm=True), 4, 2),                                                # --no-layernorm-projector
    (dict(lora_rank=0, use_lora_avhubert=False), 4, 2),                                 # plain LLM / plain AV-HuBERT (PETF_LLM_name=None)
    (dict(family="qwen2", head_dim=128), 4, 2),                                         # head_dim 128, qkv bias, no BOS, non-task-specific LoRA
])
def test_generate_config_variants_f32(kw, ra, rv):
    """Every constructor variant of AVSR_LLMs that changes the arithmetic, end to end, token ids vs the oracle."""
    spec, sd, eng, orc = make("f32", **kw)
    Tv, Ts = [11, 8], [7040, 5120]
    video, audio = inputs(Tv=Tv, Ts=Ts)
    fam = spec.llm.family
    tokens = torch.full((2, 1), spec.ids["bos"]) if fam == "llama" else torch.zeros(2, 0, dtype=torch.long)
    ids, lens = eng.generate("audiovisual", video, Tv, audio, Ts, ra, rv, beams=1, max_new=spec.max_dec_tokens, force_len=True)
    ref = orc.forward(dict(video=video, audio=audio, lengths=torch.tensor(Ts), video_lens=torch.tensor(Tv), tokens=tokens), "audiovisual", ra, rv,
                      num_beams=1, force_len=True)
    ids = ids.cpu()
    for b in range(2):
        assert ids[b].tolist() == ref[b], f"{kw}: utt {b}: {ids[b].tolist()} vs {ref[b]}"


def test_greedy_hipgraph_replay_matches_eager():
    """On a non-default stream the decode step is captured into a hipGraph and replayed; ids must equal the eager path
    (default stream, where capture is not allowed) and a second call must reuse the instantiated graph."""
    spec, sd, eng, orc = make("f32", family="llama", task_specific=True, shared=True)
    Tv, Ts = [12, 9], [7680, 5000]
    video, audio = inputs(Tv=Tv, Ts=Ts)
    ids0, lens0 = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 2, beams=1, max_new=spec.max_dec_tokens)
    side = torch.cuda.Stream()
    outs = []
    for _ in range(2):
        with torch.cuda.stream(side):
            ids1, lens1 = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 2, beams=1, max_new=spec.max_dec_tokens, slot=1)
        side.synchronize()
        outs.append((ids1.cpu(), lens1.cpu()))
    torch.cuda.synchronize()
    for ids1, lens1 in outs:
        assert torch.equal(ids0.cpu(), ids1) and torch.equal(lens0.cpu(), lens1), f"eager {ids0.cpu().tolist()} {lens0.cpu().tolist()} vs graph {ids1.tolist()} {lens1.tolist()}"


def test_workspace_created_while_another_call_is_in_flight():
    """A second slot's workspace is created (device-side tables written on the null stream) while the first call of the same engine is
    still running; the non-blocking stream that uses it must not start before that initialisation is done.  Repeated over several
    engines because whether the two calls really overlap depends on which hardware queue the new stream lands on."""
    Tv, Ts = [12, 9], [7680, 5000]
    video, audio = inputs(Tv=Tv, Ts=Ts)
    ref = None
    for k in range(6):
        spec, sd, eng, orc = make("f32", family="llama", task_specific=True, shared=True)
        vd, ad = video.cuda(), audio.cuda()
        torch.cuda.synchronize()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(sa):
            ids0, _ = eng.generate("audiovisual", vd, Tv, ad, Ts, 4, 2, beams=1, max_new=spec.max_dec_tokens, slot=0)
        with torch.cuda.stream(sb):
            ids1, _ = eng.generate("audiovisual", vd, Tv, ad, Ts, 4, 2, beams=1, max_new=spec.max_dec_tokens, slot=1)
        torch.cuda.synchronize()
        if ref is None:
            with torch.no_grad():
                ref = orc.forward(dict(video=video, audio=audio, lengths=torch.tensor(Ts), video_lens=torch.tensor(Tv),
                                       tokens=torch.full((2, 1), spec.ids["bos"])), "audiovisual", 4, 2, num_beams=1)
        for name, ids in (("first", ids0), ("second", ids1)):
            got = [row[: len(r)] for row, r in zip(ids.cpu().tolist(), ref)]
            assert got == ref, f"engine {k}, {name} call: {got} vs {ref}"
        del eng


# bf16 end to end: tests/test_bf16_emulation_gpu.py holds the bf16 engine to the bf16-emulating oracle (quantitative bounds on
# activations, logits and a tolerance-aware id check) on the tiny and on the full-size configuration.


def test_shim_forward_matches_engine_and_errors():
    spec, sd, eng, orc = make("f32", family="llama", task_specific=True, shared=True, matryoshka=True)
    m = AVSR_LLMs.__new__(AVSR_LLMs)
    m.modality, m.max_dec_tokens, m.num_beams = "audiovisual", spec.max_dec_tokens, 1
    m.is_matryoshka, m.is_task_specific, m.spec, m.engine = True, True, spec, eng
    m.downsample_ratio_audio, m.downsample_ratio_video = spec.downsample_ratio_audio, spec.downsample_ratio_video
    video, audio = inputs(B=1, Tv=(10,), Ts=(6400,))
    batch = dict(video=video, audio=audio, lengths=torch.tensor([6400]), tokens=torch.full((1, 1), spec.ids["bos"]))
    out = m(batch, is_trainval=False, modality="audiovisual", test_ratio_matry_audio=4, test_ratio_matry_video=2)
    ref = orc.forward(dict(batch), "audiovisual", 4, 2, num_beams=1)
    assert out.dtype == torch.long and out[0].tolist() == ref[0]
    with pytest.raises(KeyError):
        m(batch, is_trainval=False, modality="audiovisual", test_ratio_matry_audio=3, test_ratio_matry_video=2)
    with pytest.raises(NotImplementedError):
        m(batch, is_trainval=True)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_digests_are_the_hashes_of_inputs_embeds_and_first_logits(dtype):
    """oavsr_workspace_set_digests: per utterance {digest of its spliced inputs_embeds rows, digest of its first-step logits row}.  The values
    are exactly the host-side hash of what the STAGE entry points return for the same inputs; different clips give different digests; an
    utterance keeps its digests whatever its position / batch (batch invariance); a call through the decode scheduler reports the same pair."""
    from gpu_util import digest_words
    spec, sd, eng, orc = make(dtype)
    Tv, Ts = [12, 9, 7], [7680, 5000, 4480]
    video, audio = inputs(B=3, Tv=Tv, Ts=Ts, seed=3)
    dg = torch.zeros(3, 2, dtype=torch.int64, device="cuda:0")
    ids, lens = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 2, beams=1, max_new=spec.max_dec_tokens, digests=dg)
    torch.cuda.synchronize()
    a_out, na = eng.encode_audio(audio, Ts, 4)
    v_out, nv = eng.encode_video(video, Tv, 2)
    emb, S = eng.build_prompt("audiovisual", 3, na, eng.project("audio", 4, a_out), nv, eng.project("video", 2, v_out))
    logits = eng.prefill("audiovisual", emb, S)
    torch.cuda.synchronize()
    off = 0
    for b in range(3):
        assert int(dg[b, 0]) == digest_words(emb[off:off + S[b]]), f"utterance {b}: inputs_embeds digest"
        assert int(dg[b, 1]) == digest_words(logits[b]), f"utterance {b}: first-step logits digest"
        off += S[b]
    assert len({int(x) for x in dg[:, 0]}) == 3 and len({int(x) for x in dg[:, 1]}) == 3, "different clips must give different digests"
    # utterance 1 alone, and as row 0 of another batch: same digests
    d1 = torch.zeros(1, 2, dtype=torch.int64, device="cuda:0")
    eng.generate("audiovisual", video[1:2, : Tv[1]], [Tv[1]], audio[1:2, : Ts[1]], [Ts[1]], 4, 2, beams=1, max_new=spec.max_dec_tokens, digests=d1)
    d2 = torch.zeros(2, 2, dtype=torch.int64, device="cuda:0")
    eng.generate("audiovisual", video[[1, 0]], [Tv[1], Tv[0]], audio[[1, 0]], [Ts[1], Ts[0]], 4, 2, beams=1, max_new=spec.max_dec_tokens, digests=d2)
    torch.cuda.synchronize()
    assert torch.equal(d1[0], dg[1]) and torch.equal(d2[0], dg[1]) and torch.equal(d2[1], dg[0])
    # through the decode scheduler
    ds = torch.zeros(3, 2, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()              # (the zero fill ran on the default stream)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ids_s, lens_s, ticket = eng.submit("audiovisual", video, Tv, audio, Ts, 4, 2, max_new=spec.max_dec_tokens, digests=ds)
        eng.wait(ticket)
    st.synchronize()
    assert torch.equal(ds, dg) and torch.equal(ids_s, ids)
    # off again: a later call leaves the buffer alone
    dg2 = dg.clone()
    eng.generate("audiovisual", video[:1, : Tv[0]], [Tv[0]], audio[:1, : Ts[0]], [Ts[0]], 4, 2, beams=1, max_new=spec.max_dec_tokens)
    torch.cuda.synchronize()
    assert torch.equal(dg, dg2)
