"""-m gpu: the callers and data formats either side of the hot path, end to end on the HIP engine.

f-2  manifest -> test-time transforms -> collate (B > 1, ragged) -> `AVSR_LLMs.forward` -> text -> WER  (datamodule/av_dataset.py:78-114,
     transforms.py:93-126, data_module.py:19-79, lightning_OmniAVSR.py:194-219) through `harness.evaluate`: the decoded ids equal the
     oracle's on the same transformed tensors, the WER counters equal hand-computed values.
f-3  a checkpoint written in each layout a reference user has on disk -- `model_avg_N.pth` (utils/avg_checkpoints.py:14-44), or the
     AV-HuBERT fairseq pickle (checkpoint_utils.py:303-329) + Whisper / LLM safetensors directories + trained adapter file with
     `resize_token_embeddings` (modeling_OmniAVSR.py:214) -- ingested and decoded: same ids as the direct state dict."""
import json
import os
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd import harness as H  # noqa: E402
from omni_avsr_amd import ingest  # noqa: E402
from omni_avsr_amd.config import tiny_spec  # noqa: E402
from omni_avsr_amd.modeling import AVSR_LLMs, OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402


def _shim(spec, eng, modality):
    m = AVSR_LLMs.__new__(AVSR_LLMs)
    m.modality, m.max_dec_tokens, m.num_beams = modality, spec.max_dec_tokens, 1
    m.is_matryoshka, m.is_task_specific, m.spec, m.engine = spec.is_matryoshka, spec.is_task_specific, spec, eng
    m.downsample_ratio_audio, m.downsample_ratio_video = spec.downsample_ratio_audio, spec.downsample_ratio_video
    return m


def test_manifest_to_wer_through_the_engine(tmp_path):
    spec = tiny_spec(family="llama", task_specific=True, shared=True, matryoshka=True)
    sd = synth_state_dict(spec, seed=1234)
    eng = OmniEngine(spec, dtype="f32").load_state_dict(sd)
    orc = O.OmniOracle(sd, spec.oracle_cfg())
    model = _shim(spec, eng, "audiovisual")
    # a manifest in the reference's format and "decoded media": uint8 RGB 96 x 96 clips, raw waveforms of unequal length
    frames = [9, 12, 7, 11, 10]
    man = tmp_path / "test.csv"
    man.write_text("".join(f"lrs3,test/spk/{i:04d}.mp4,{n},{n * 4},GOLD {i}\n" for i, n in enumerate(frames)))
    rows = H.load_manifest(str(man))
    assert [r[2] for r in rows] == frames
    g = torch.Generator().manual_seed(3)
    samples, raw = [], []
    for _, _, n, _ in rows:
        clip = torch.randint(0, 256, (n, 3, 96, 96), generator=g, dtype=torch.uint8)
        wave = torch.randn(n * 640 + int(torch.randint(-300, 300, (1,), generator=g)), 1, generator=g) * 0.1
        samples.append(H.make_sample(clip, wave, "audiovisual"))
        raw.append((clip, wave))
    for s_, n in zip(samples, frames):
        assert s_["video"].shape == (n, 1, 88, 88) and s_["audio"].shape == (640 * n, 1)          # cut_or_pad to 640 * frames
    # a toy vocabulary: token t <-> word "w<t>"; special ids are dropped as skip_special_tokens does
    special = set(spec.ids.values())
    to_text = lambda ids: [" ".join(f"w{int(t)}" for t in row if int(t) not in special) for row in ids]      # noqa: E731
    # the oracle on the same transformed tensors, one utterance at a time (the reference's batch-1 evaluation)
    ref_ids = []
    with torch.no_grad():
        for s_ in samples:
            one = dict(video=s_["video"][None], audio=s_["audio"][None], lengths=torch.tensor([s_["audio"].shape[0]]),
                       tokens=torch.full((1, 1), spec.ids["bos"]))
            ref_ids.append(orc.forward(one, "audiovisual", 4, 2, num_beams=1)[0])
    ref_text = to_text(ref_ids)
    # gold transcripts: utterances 0 and 3 exact, 1 with its first word replaced, 2 with one extra word, 4 empty hypothesis edits
    gold = list(ref_text)
    w1 = gold[1].split()
    gold[1] = " ".join(["zzz"] + w1[1:])
    gold[2] = gold[2] + " extra"
    gold[4] = "only these words"
    pairs = [(s_, t) for s_, t in zip(samples, gold)]
    meter = H.evaluate(model, pairs, "audiovisual", to_text, batch_size=3, bos_id=spec.ids["bos"], test_ratio_matry_audio=4, test_ratio_matry_video=2)
    by_gold = {g_: h for g_, h, _, _ in meter.per_utt}
    for i in range(5):
        assert by_gold[gold[i]] == ref_text[i], f"utterance {i}: engine text differs from the oracle's: {by_gold[gold[i]]!r} vs {ref_text[i]!r}"
    e4 = H.edit_distance(gold[4].split(), ref_text[4].split())
    assert meter.total_edit_distance == 0 + 1 + 1 + 0 + e4
    assert meter.total_length == sum(len(t.split()) for t in gold) and not meter.failed
    # two "ranks" (utterance sharding): the union of their meters is the single-rank meter
    m0 = H.evaluate(model, pairs, "audiovisual", to_text, batch_size=2, bos_id=spec.ids["bos"], rank=0, world=2, test_ratio_matry_audio=4, test_ratio_matry_video=2)
    m1 = H.evaluate(model, pairs, "audiovisual", to_text, batch_size=2, bos_id=spec.ids["bos"], rank=1, world=2, test_ratio_matry_audio=4, test_ratio_matry_video=2)
    assert m0.total_edit_distance + m1.total_edit_distance == meter.total_edit_distance and m0.total_length + m1.total_length == meter.total_length
    # batches kept in flight on their own streams / workspaces (what bench.py measures): same hypotheses, same counters
    for k, grp in ((2, 1), (3, 1), (1, 2), (2, 2)):
        mp = H.evaluate(model, pairs, "audiovisual", to_text, batch_size=2, bos_id=spec.ids["bos"], in_flight=k, decode_group=grp,
                        test_ratio_matry_audio=4, test_ratio_matry_video=2)
        assert sorted(mp.per_utt) == sorted(meter.per_utt), f"in_flight={k}, decode_group={grp}: hypotheses differ from the sequential loop"
        assert (mp.total_edit_distance, mp.total_length, mp.failed) == (meter.total_edit_distance, meter.total_length, [])
    assert getattr(model, "encoder_chunk", None) is None
    # the decoded uint8 frames themselves as samples (make_sample(device_transform=True)): the test-time transform runs inside the stem kernel,
    # the host uploads [sum T, 3, 96, 96] bytes -- same hypotheses and counters, sequential and with batches in flight (f32 AND bf16 engines)
    pairs_u8 = [(H.make_sample(clip, wave, "audiovisual", device_transform=True), t) for (clip, wave), t in zip(raw, gold)]
    assert pairs_u8[0][0]["video_u8"].dtype == torch.uint8 and "video" not in pairs_u8[0][0]
    for k in (1, 3):
        mu = H.evaluate(model, pairs_u8, "audiovisual", to_text, batch_size=2, bos_id=spec.ids["bos"], in_flight=k, test_ratio_matry_audio=4, test_ratio_matry_video=2)
        assert sorted(mu.per_utt) == sorted(meter.per_utt), f"uint8 ingest (in_flight={k}): hypotheses differ from the host-transform path"
    eng16 = OmniEngine(spec, dtype="bf16").load_state_dict(sd)
    m16 = _shim(spec, eng16, "audiovisual")
    a16 = H.evaluate(m16, pairs, "audiovisual", to_text, batch_size=2, bos_id=spec.ids["bos"], in_flight=2, test_ratio_matry_audio=4, test_ratio_matry_video=2)
    b16 = H.evaluate(m16, pairs_u8, "audiovisual", to_text, batch_size=2, bos_id=spec.ids["bos"], in_flight=2, test_ratio_matry_audio=4, test_ratio_matry_video=2)
    assert sorted(a16.per_utt) == sorted(b16.per_utt), "bf16: uint8 ingest decodes differently from the host-transform path"


def _decode(eng, spec, seed=5):
    g = torch.Generator().manual_seed(seed)
    Tv, Ts = [11, 8], [7040, 5120]
    video = torch.randn(2, 11, 1, 88, 88, generator=g)
    audio = torch.randn(2, 7040, 1, generator=g)
    video[1, 8:] = 0
    audio[1, 5120:] = 0
    ids, lens = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 2, beams=1, max_new=spec.max_dec_tokens, force_len=True)
    return ids.cpu().tolist()


def test_checkpoint_formats_decode_like_the_direct_state_dict(tmp_path):
    import omni_avsr_amd.config as C
    spec = tiny_spec(family="llama", task_specific=True, shared=True, matryoshka=True)
    sd = synth_state_dict(spec, seed=1234)
    want = _decode(OmniEngine(spec, dtype="f32").load_state_dict(sd), spec)
    C.LLM_SPECS["tiny/llama"] = spec.llm
    kw = dict(lora_alpha=spec.lora_alpha, downsample_ratio_audio=[2, 4], downsample_ratio_video=[1, 2], ids=spec.ids, max_dec_tokens=8, num_beams=1)

    # (a) model_avg_N.pth: the whole AVSR_LLMs state dict, written with the Lightning "model." prefix the averaging script strips
    p = tmp_path / "model_avg_10.pth"
    torch.save({"model." + k: v for k, v in sd.items()}, p)
    got = ingest.load_omni_checkpoint(str(p))
    s2 = ingest.infer_spec(got, "tiny/llama", **kw)
    assert _decode(OmniEngine(s2, dtype="f32").load_state_dict(got), s2) == want

    # (b) the separate artefacts: fairseq pickle (with an un-importable cfg class), HF safetensors directories for Whisper and the
    #     LLM at its BASE vocabulary, and the trained adapter file that carries the added embedding rows
    from safetensors.torch import save_file
    mod = types.ModuleType("fake_omegaconf_pkg2")
    exec("class DictConfig:\n    def __init__(self):\n        self.content = {'x': 1}\n", mod.__dict__)
    mod.DictConfig.__module__ = "fake_omegaconf_pkg2"
    sys.modules["fake_omegaconf_pkg2"] = mod
    torch.save({"cfg": {"model": mod.DictConfig()}, "model": {k[len("video_encoder."):]: v for k, v in sd.items() if k.startswith("video_encoder.")},
                "task_state": {}}, tmp_path / "large_vox_iter5.pt")
    del sys.modules["fake_omegaconf_pkg2"]
    wdir, ldir = tmp_path / "whisper", tmp_path / "llm"
    os.makedirs(wdir)
    os.makedirs(ldir)
    save_file({"model.encoder." + k[len("audio_encoder."):]: v.contiguous() for k, v in sd.items() if k.startswith("audio_encoder.")}, str(wdir / "model.safetensors"))
    n_added = 5
    base_rows = spec.llm.vocab - n_added
    llm = {k[len("llm."):]: v for k, v in sd.items() if k.startswith("llm.") and "lora_" not in k and k != "llm.lm_head.weight"}
    llm["model.embed_tokens.weight"] = sd["llm.model.embed_tokens.weight"][:base_rows].contiguous()       # tie_word_embeddings: no lm_head tensor
    names = sorted(llm)
    half = len(names) // 2
    save_file({k: llm[k].contiguous() for k in names[:half]}, str(ldir / "model-00001-of-00002.safetensors"))
    save_file({k: llm[k].contiguous() for k in names[half:]}, str(ldir / "model-00002-of-00002.safetensors"))
    json.dump({"weight_map": {**{k: "model-00001-of-00002.safetensors" for k in names[:half]}, **{k: "model-00002-of-00002.safetensors" for k in names[half:]}}},
              open(ldir / "model.safetensors.index.json", "w"))
    adapter = {k: v for k, v in sd.items() if "lora_" in k and k.startswith("llm.") or k.startswith(("audio_proj.", "video_proj.", "prompt_"))}
    adapter["llm.added_rows"] = sd["llm.model.embed_tokens.weight"][base_rows:].clone()
    torch.save(adapter, tmp_path / "adapter.pth")
    asm = ingest.assemble_state_dict(str(tmp_path / "large_vox_iter5.pt"), str(wdir), str(ldir), str(tmp_path / "adapter.pth"), vocab_rows=spec.llm.vocab)
    assert set(asm) == set(sd), sorted(set(asm) ^ set(sd))[:10]
    assert all(torch.equal(asm[k], sd[k]) for k in sd)
    s3 = ingest.infer_spec(asm, "tiny/llama", **kw)
    assert _decode(OmniEngine(s3, dtype="f32").load_state_dict(asm), s3) == want

    # (c) a checkpoint whose vocabulary was never resized is refused like the reference's strict load (not read out of bounds)
    bad = dict(sd)
    bad["llm.model.embed_tokens.weight"] = sd["llm.model.embed_tokens.weight"][:base_rows]
    bad["llm.lm_head.weight"] = bad["llm.model.embed_tokens.weight"]
    with pytest.raises(RuntimeError, match="size mismatch"):
        OmniEngine(spec, dtype="f32").load_state_dict(bad)
