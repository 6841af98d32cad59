"""CPU: host-side L4 counterparts (manifest, transforms, WER) against hand-computed expectations."""
import torch

from omni_avsr_amd import harness as H


def test_manifest_and_cut_or_pad(tmp_path):
    p = tmp_path / "m.csv"
    p.write_text("lrs3,test/a/0001.mp4,57,12,HELLO WORLD\nlrs3,test/b/0002.mp4,400,30,IT IS\n")
    assert H.load_manifest(str(p)) == [("lrs3", "test/a/0001.mp4", 57, "HELLO WORLD"), ("lrs3", "test/b/0002.mp4", 400, "IT IS")]
    a = torch.arange(10.0).reshape(5, 2)
    assert H.cut_or_pad(a, 3).shape == (3, 2) and torch.equal(H.cut_or_pad(a, 3), a[:3])
    pd = H.cut_or_pad(a, 7)
    assert pd.shape == (7, 2) and torch.equal(pd[:5], a) and float(pd[5:].abs().sum()) == 0


def test_video_transform_matches_formula():
    g = torch.Generator().manual_seed(0)
    fr = torch.randint(0, 256, (4, 3, 96, 96), generator=g, dtype=torch.uint8)
    out = H.video_test_transform(fr)
    assert out.shape == (4, 1, 88, 88) and out.dtype == torch.float32
    x = fr.float()[:, :, 4:92, 4:92] / 255
    ref = ((0.2989 * x[:, 0] + 0.587 * x[:, 1] + 0.114 * x[:, 2]) - 0.421) / 0.165
    assert torch.allclose(out[:, 0], ref, atol=1e-6)
    mono = H.video_test_transform(fr[:, :1])
    assert torch.allclose(mono[:, 0], (fr[:, 0, 4:92, 4:92].float() / 255 - 0.421) / 0.165, atol=1e-6)


def test_audio_transform_and_noise():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(16000, 1, generator=g) * 0.3 + 0.1
    out = H.audio_test_transform(w)
    assert abs(float(out.mean())) < 1e-5 and abs(float(out.var(unbiased=False)) - 1) < 1e-4
    noise = torch.randn(1, 40000, generator=g)
    for snr in (-5.0, 0.0, 15.0):
        noisy = H.add_noise(w.t(), noise[:, :16000], snr)
        resid = noisy - w.t()
        got = 10 * torch.log10((w.t() ** 2).sum() / (resid ** 2).sum())
        assert abs(float(got) - snr) < 1e-3
    s = H.make_sample(torch.zeros(10, 1, 96, 96, dtype=torch.uint8), torch.randn(5000, 1, generator=g), "audiovisual")
    assert s["video"].shape == (10, 1, 88, 88) and s["audio"].shape == (6400, 1)      # cut_or_pad to 640 * frames


def test_wer_and_collate():
    assert H.edit_distance("kitten", "sitting") == 3
    assert H.compute_word_level_distance("The cat sat", "the cat sat") == 0
    assert H.compute_word_level_distance("a b c d", "a x c") == 2
    m = H.WerMeter()
    m.update("hello big world", "hello world")
    m.update("one two", "one two three")
    assert m.total_edit_distance == 2 and m.total_length == 5 and abs(m.wer - 0.4) < 1e-9
    assert H.strip_special([5, 1, 7, 2, 2], [1, 2]) == [5, 7]
    b = H.collate_test([{"video": torch.ones(3, 1, 88, 88), "audio": torch.ones(1920, 1)}, {"video": torch.ones(5, 1, 88, 88), "audio": torch.ones(3200, 1)}],
                       "audiovisual", 128000)
    assert b["video"].shape == (2, 5, 1, 88, 88) and b["audio"].shape == (2, 3200, 1)
    assert b["lengths"].tolist() == [1920, 3200] and b["video_lengths"].tolist() == [3, 5] and b["tokens"].tolist() == [[128000], [128000]]
    assert H.collate_test([{"audio": torch.ones(10, 1)}], "audio", -1)["tokens"].shape == (1, 0)      # Qwen: no BOS


def _toy_tokenizer(family):
    """A fast tokenizer with the special tokens of the real families (no hub access offline): word-level vocabulary."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    words = ["hello", "world", "the", "cat", "sat", "[UNK]"]
    specials = ["<|begin_of_text|>", "<|end_of_text|>"] if family == "llama" else ["<|endoftext|>"]
    vocab = {w: i for i, w in enumerate(words + specials)}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    kw = dict(bos_token="<|begin_of_text|>", eos_token="<|end_of_text|>") if family == "llama" else dict(eos_token="<|endoftext|>")
    return PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="[UNK]", **kw), len(vocab)


def test_tokenizer_surgery_follows_lightning_module():
    """lightning_OmniAVSR.py:53-91: post-processor template, added tokens (+5 Llama / +4 Qwen), right padding, pad id; and the ids the
    inference branch reads from tokenizer.vocab (modeling_OmniAVSR.py:312-322)."""
    tok, n0 = _toy_tokenizer("llama")
    pad_id = H.prepare_tokenizer(tok, "meta-llama/Llama-3.2-1B")
    assert len(tok) == n0 + 5 and pad_id == tok.vocab["<pad>"] == n0 and tok.padding_side == "right"
    assert [tok.vocab[t] for t in ("<audio>", "</audio>", "<video>", "</video>")] == [n0 + 1, n0 + 2, n0 + 3, n0 + 4]
    ids = tok("hello world").input_ids
    assert ids == [tok.bos_token_id, 0, 1, tok.eos_token_id]                      # bos $A eos  (:59-71)
    sp = H.special_ids_from_tokenizer(tok, "meta-llama/Llama-3.2-1B")
    assert sp == dict(bos=tok.vocab["<|begin_of_text|>"], eos=tok.vocab["<|end_of_text|>"], pad=n0, audio_sos=n0 + 1, audio_eos=n0 + 2,
                      video_sos=n0 + 3, video_eos=n0 + 4)
    assert tok.batch_decode([[tok.bos_token_id, 2, 3, 4, tok.eos_token_id, n0]], skip_special_tokens=True, clean_up_tokenization_spaces=False) == ["the cat sat"]
    tq, q0 = _toy_tokenizer("qwen")
    assert H.prepare_tokenizer(tq, "Qwen/Qwen2.5-7B") is None and len(tq) == q0 + 4
    assert tq("the cat").input_ids == [2, 3, tq.eos_token_id]                     # $A eos, no BOS  (:72-81)
    sq = H.special_ids_from_tokenizer(tq, "Qwen/Qwen2.5-7B")
    assert sq["bos"] == -1 and sq["eos"] == sq["pad"] == tq.vocab["<|endoftext|>"] and sq["video_eos"] == q0 + 3


def test_resize_token_embeddings_and_assemble(tmp_path):
    from omni_avsr_amd import ingest
    emb = torch.arange(12.0).reshape(4, 3)
    sd = {"llm.model.embed_tokens.weight": emb, "llm.lm_head.weight": emb}           # tied
    out = ingest.resize_token_embeddings(sd, 6)
    assert out["llm.model.embed_tokens.weight"].shape == (6, 3) and out["llm.lm_head.weight"].data_ptr() == out["llm.model.embed_tokens.weight"].data_ptr()
    assert torch.allclose(out["llm.model.embed_tokens.weight"][4:], emb.mean(0, keepdim=True).expand(2, 3))
    added = torch.ones(2, 3) * 7
    import pytest
    untied = {"llm.model.embed_tokens.weight": emb, "llm.lm_head.weight": emb + 1}
    with pytest.raises(ValueError, match="untied"):          # one tensor cannot serve two different matrices
        ingest.resize_token_embeddings(untied, 6, added)
    out2 = ingest.resize_token_embeddings(untied, 6, {"embed": added, "lm_head": added * 2})
    assert torch.equal(out2["llm.model.embed_tokens.weight"][4:], added) and torch.equal(out2["llm.lm_head.weight"][:4], emb + 1)
    assert torch.equal(out2["llm.lm_head.weight"][4:], added * 2)
    with pytest.raises(ValueError):
        ingest.resize_token_embeddings(sd, 3)


def test_evaluate_reports_failed_batches():
    class Boom:
        is_task_specific = False

        def __call__(self, batch, **kw):
            if batch["video"].shape[1] == 7:
                raise RuntimeError("bad clip")
            return torch.tensor([[1, 2]] * batch["video"].shape[0])

    import pytest
    from omni_avsr_amd.lib import DeviceError
    samples = [({"video": torch.zeros(t, 1, 88, 88)}, "w1 w2") for t in (5, 7, 5, 5)]
    # default: like the reference's loop, the first failing batch ends the evaluation
    with pytest.raises(RuntimeError, match="bad clip"):
        H.evaluate(Boom(), samples, "video", lambda ids: ["w1 w2"] * ids.shape[0], batch_size=1)
    # opt-in tolerance: the batch is reported, left out of the counters, and shows up in the summary
    m = H.evaluate(Boom(), samples, "video", lambda ids: ["w1 w2"] * ids.shape[0], batch_size=1, allow_failures=True)
    assert m.failed == [1] and m.total_length == 6 and m.total_edit_distance == 0
    assert m.summary() == dict(wer=0.0, edits=0, words=6, utterances=3, failed=1)

    class Sticky(Boom):
        def __call__(self, batch, **kw):
            raise DeviceError("HIP error 719")
    with pytest.raises(DeviceError):      # a device fault is never "one bad clip": every later batch would fail the same way
        H.evaluate(Sticky(), samples, "video", lambda ids: ["w1 w2"] * ids.shape[0], batch_size=1, allow_failures=True)


def test_evaluate_with_an_empty_shard_returns_an_empty_meter():
    """A rank that owns no utterance (world > utterances) or an empty sample list must come back with an empty meter -- it still has to
    reach the collectives behind evaluate() -- and the process-wide launch objective must be what the caller had set."""
    class Obj:
        def __init__(self):
            self.alone, self.calls = False, []

        def launch_objective(self, alone):
            prev, self.alone = self.alone, bool(alone)
            self.calls.append(bool(alone))
            return prev

    class M:
        is_task_specific = False

        def __init__(self):
            self.engine = Obj()

        def __call__(self, batch, **kw):
            return torch.tensor([[1, 2]] * batch["video"].shape[0])

    m = M()
    meter = H.evaluate(m, [], "video", lambda ids: ["w1 w2"] * ids.shape[0], batch_size=2)
    assert meter.total_length == 0 and meter.total_edit_distance == 0 and meter.per_utt == {} or len(meter.per_utt) == 0
    samples = [({"video": torch.zeros(5, 1, 88, 88)}, "w1 w2")]
    meter = H.evaluate(m, samples, "video", lambda ids: ["w1 w2"] * ids.shape[0], batch_size=2, rank=1, world=2)      # rank 1 of 2 owns nothing
    assert meter.total_length == 0 and len(meter.per_utt) == 0
    # the owning rank scores it, and restores the objective it found (here: "alone" set by the caller)
    m.engine.alone = True
    meter = H.evaluate(m, samples, "video", lambda ids: ["w1 w2"] * ids.shape[0], batch_size=2, rank=0, world=2)
    assert meter.total_length == 2 and m.engine.alone is True


def test_reduce_wer_counters_carries_the_failed_count():
    from omni_avsr_amd.dist import reduce_wer_counters
    assert reduce_wer_counters(3, 10) == (3, 10) and reduce_wer_counters(3, 10, failed=2) == (3, 10, 2)


def test_flat_collator_matches_collate_test():
    """FlatCollator (pinned staging, engine layout) carries the same data as the reference-format collate: the valid frames back to back,
    the waveforms right-padded with zeros, the same lengths -- also when its buffers are reused by a shorter batch."""
    import torch
    from omni_avsr_amd import harness as H
    g = torch.Generator().manual_seed(0)
    col = H.FlatCollator(3, 12, 12 * 640, pin=False)
    for frames in ([9, 12, 7], [5, 3]):
        samples = [dict(video=torch.randn(n, 1, 88, 88, generator=g), audio=torch.randn(n * 640, 1, generator=g)) for n in frames]
        ref = H.collate_test(samples, "audiovisual", 5)
        got = col(samples, "audiovisual", 5)
        assert torch.equal(got["tokens"], ref["tokens"]) and torch.equal(got["lengths"], ref["lengths"]) and torch.equal(got["video_lengths"], ref["video_lengths"])
        assert torch.equal(got["audio"], ref["audio"][..., 0])
        flat = torch.cat([ref["video"][b, :n, 0] for b, n in enumerate(frames)], 0)
        assert torch.equal(got["video_flat"], flat)


def test_raw_uint8_samples_and_flat_collation():
    """make_sample(device_transform=True) keeps the decoded uint8 frames; FlatCollator carries them back to back ("video_u8_flat")"""
    import torch
    from omni_avsr_amd import harness as H
    g = torch.Generator().manual_seed(1)
    clips = [torch.randint(0, 256, (n, 3, 96, 96), generator=g, dtype=torch.uint8) for n in (6, 4)]
    waves = [torch.randn(n * 640 + 17, 1, generator=g) for n in (6, 4)]
    samples = [H.make_sample(c, w, "audiovisual", device_transform=True) for c, w in zip(clips, waves)]
    assert all(s["video_u8"].dtype == torch.uint8 and s["audio"].shape[0] == 640 * n for s, n in zip(samples, (6, 4)))
    col = H.FlatCollator(2, 6, 6 * 640, pin=False, u8_shape=(3, 96, 96))
    b = col(samples, "audiovisual", 5)
    assert b["video_u8_flat"].shape == (10, 3, 96, 96) and torch.equal(b["video_u8_flat"], torch.cat(clips)) and b["video_lengths"].tolist() == [6, 4]
    ref = torch.cat([H.video_test_transform(c) for c in clips])           # what the host path would have uploaded instead
    assert ref.shape == (10, 1, 88, 88)
    import pytest
    with pytest.raises(ValueError):
        H.make_sample(clips[0].float(), waves[0], "audiovisual", device_transform=True)
