"""-m gpu: what an utterance's result may depend on.  The reference evaluates with batch 1 (data_module.py:238-242); this engine
batches, so "alone" and "inside a batch" must agree.

Contract:
  * encoders, projector, prompt and prefill: bit-identical for any batch (var-len kernels, no split-K; the pruned last prefill layer and
    the first lm_head run on the decode-shaped path) -- asserted on the prefill logits in both dtypes;
  * decode: ONE regime at every row count, greedy and beam search alike -- the tiled LDS-DMA GEMMs with split factors that are functions
    of N and K only and the slabs summed in slab order inside the launch; rows are independent, launches carry <= 128 rows (more rows:
    more launches of the same kernels).  An utterance's ids are bit-identical alone, in a batch of any size and as K beams x B utterances,
    in f32 AND in bf16 -- 1 / 8 / 33 / 128 / 130 rows below."""
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
def test_greedy_ids_do_not_depend_on_the_batch(dtype):
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=1234))
    base = batch(1, seed=7)
    alone = gen(eng, 1, 1, spec.max_dec_tokens, base)
    for B in (2, 8, 9, 33, 64, 128, 130):               # 130 rows: two launches per decode GEMM (128 + 2 rows)
        got = gen(eng, B, 1, spec.max_dec_tokens, base)
        assert got == alone, f"{dtype}: greedy ids of an utterance change between 1 and {B} rows: {alone} vs {got}"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_beam15_ids_do_not_depend_on_the_batch(dtype):
    """the reference's own protocol is B = 1 (15 rows); a batch of 8 utterances runs 120 rows, 9 utterances 135 (two launches), 35 utterances 525"""
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    eng = OmniEngine(spec, dtype=dtype).load_state_dict(synth_state_dict(spec, seed=1234))
    base = batch(1, seed=11)
    alone = gen(eng, 1, 15, spec.max_dec_tokens, base)
    for B in (2, 3, 8, 9, 35):
        got = gen(eng, B, 15, spec.max_dec_tokens, base)
        assert got == alone, f"{dtype}: beam-15 ids of an utterance change between B = 1 and B = {B}: {alone} vs {got}"


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
