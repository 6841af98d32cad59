"""-m gpu: what an utterance's result may depend on.  The reference evaluates with batch 1 (data_module.py:238-242); this engine
batches, so "alone" and "inside a batch" must agree.

Contract:
  * encoders, projector, prompt and prefill: bit-identical for any batch (static split-K, var-len kernels) -- asserted on the
    prefill logits in both dtypes;
  * decode: the kernel family follows the decode regime, never the data: rows <= 32 (weight-streaming kernels; the 9..32-row
    kernel sums its partials exactly as the <= 8-row kernel does) | > 32 (tiled split-K GEMMs, split factors functions of N and K
    only); decode attention: one wave layout per mode (greedy / beam).  Inside a regime the ids are bit-identical in both dtypes;
    across the 32-row boundary the fp32 summation order differs, so bit-identity is asserted in f32 mode (1e-6-level differences do
    not move an arg-max here) and not claimed for bf16."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd.config import tiny_spec  # noqa: E402
from omni_avsr_amd.modeling import OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402


def batch(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    Tv = [12] + [int(x) for x in torch.randint(5, 13, (B - 1,), generator=g)]
    Ts = [640 * t for t in Tv]
    video = torch.randn(B, 12, 1, 88, 88, generator=g)
    audio = torch.randn(B, 7680, 1, generator=g)
    for b in range(B):
        video[b, Tv[b]:] = 0
        audio[b, Ts[b]:] = 0
    return video, audio, Tv, Ts


def gen(eng, B, beams, max_new, base):
    """utterance 0 is always `base`'s utterance 0; the other B - 1 rows are different clips"""
    video, audio, Tv, Ts = batch(B, seed=100 + B)
    video[0], audio[0] = base[0][0], base[1][0]
    ids, lens = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 2, beams=beams, max_new=max_new, force_len=(beams == 1))
    return ids[0].cpu().tolist(), int(lens[0])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_greedy_alone_vs_batches_crossing_kernel_regimes(dtype):
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=1234))
    base = batch(1, seed=7)
    alone = gen(eng, 1, 1, spec.max_dec_tokens, base)
    same_regime = [gen(eng, B, 1, spec.max_dec_tokens, base) for B in (2, 5, 8)]            # <= 8 rows: 8-column tiles
    assert all(r == alone for r in same_regime), f"greedy ids depend on the batch inside the <= 8-row regime: {alone} vs {same_regime}"
    mid = [gen(eng, B, 1, spec.max_dec_tokens, base) for B in (9, 20, 32)]                  # 9..32 rows: 8-column tiles, 16-row operands
    assert all(r == alone for r in mid), f"greedy ids of an utterance change between 1 and 9..32 rows: {alone} vs {mid}"
    mid2 = [gen(eng, B, 1, spec.max_dec_tokens, base) for B in (33, 48, 64)]               # 33..64 rows (round 4): the same tiles, four 16-row operands
    assert all(r == alone for r in mid2), f"greedy ids of an utterance change between 1 and 33..64 rows: {alone} vs {mid2}"
    big = [gen(eng, B, 1, spec.max_dec_tokens, base) for B in (65, 70, 130)]               # > 64 rows: split-K tiles, 1 and 2 M-tiles
    assert all(r == big[0] for r in big), f"greedy ids depend on the batch inside the > 64-row regime: {big}"
    if dtype == "f32":
        assert big[0] == alone, f"f32 ids differ across the 64-row boundary: {alone} / {big[0]}"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_beam15_alone_vs_batches(dtype):
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=1234))
    base = batch(1, seed=11)
    small = [gen(eng, B, 15, spec.max_dec_tokens, base) for B in (1, 2)]                    # 15 / 30 rows: weight-streaming kernels
    assert small[0] == small[1], f"beam-15 ids depend on the batch (<= 32 rows): {small}"
    big = [gen(eng, B, 15, spec.max_dec_tokens, base) for B in (3, 8, 9, 35)]               # 45 / 120 / 135 / 525 rows: split-K tiles
    assert all(r == big[0] for r in big), f"beam-15 ids depend on the batch (> 32 rows): {big}"
    if dtype == "f32":
        assert big[0] == small[0], f"f32 beam ids differ across decode regimes: {small[0]} vs {big[0]}"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_prefill_logits_bitwise_independent_of_batch(dtype):
    spec = tiny_spec(family="qwen2", head_dim=128)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=4321))
    base = batch(1, seed=3)

    def logits0(B):
        video, audio, Tv, Ts = batch(B, seed=50 + B)
        video[0], audio[0] = base[0][0], base[1][0]
        eng.ensure_workspace(40, 12, 7680)
        a_out, na = eng.encode_audio(audio, Ts, 4)
        v_out, nv = eng.encode_video(video, Tv, 2)
        emb, S = eng.build_prompt("audiovisual", B, na, eng.project("audio", 4, a_out), nv, eng.project("video", 2, v_out))
        return eng.prefill("audiovisual", emb, S)[0].clone()

    ref = logits0(1)
    for B in (2, 7, 33):
        assert torch.equal(logits0(B), ref), f"prefill logits of an utterance change with batch size {B}"
