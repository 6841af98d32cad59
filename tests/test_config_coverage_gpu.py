"""-m gpu: the architecture combinations of BASELINE.json's configs that the tiny default never exercised:
  * Qwen2 with task-specific + shared LoRA  (configs[3], "Omni-LoRA-hybrid"; Qwen_LoRA.py:557-570)
  * Llama with head_dim 128, task-specific + shared LoRA and llama3 RoPE scaling  (configs[4], Llama-3.1-8B; Llama_LoRA.py:250-259)
  * compression rates 5, 8 and 16 (README rates (4,16) / (2,5); configs[3] cycles {2,4,8})
  * a Whisper-small-shaped audio encoder (768 wide, 12 heads x 64, FFN 3072; configs[0]) at its full 12 layers, ASR only
  * VSR only with batch 16  (configs[2])
All in f32 parity mode: token ids bit-exact against the oracle (greedy and HF beam search), encoder activations within 1e-3."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd.config import EncoderSpec, tiny_spec  # noqa: E402
from omni_avsr_amd.modeling import OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402


def clips(Tv, Ts, seed):
    g = torch.Generator().manual_seed(seed)
    B = len(Tv)
    video = torch.randn(B, max(Tv), 1, 88, 88, generator=g)
    audio = torch.randn(B, max(Ts), 1, generator=g)
    for b in range(B):
        video[b, Tv[b]:] = 0
        audio[b, Ts[b]:] = 0
    return video, audio


def check_ids(spec, task, Tv, Ts, ra, rv, beams=(1,), seed=0, force=True):
    sd = synth_state_dict(spec, seed=1234)
    eng = OmniEngine(spec, dtype="f32").load_state_dict(sd)
    orc = O.OmniOracle(sd, spec.oracle_cfg())
    video, audio = clips(Tv, Ts, seed)
    B = len(Tv)
    tokens = torch.full((B, 1), spec.ids["bos"]) if spec.llm.family == "llama" else torch.zeros(B, 0, dtype=torch.long)
    batch = dict(video=video, audio=audio, lengths=torch.tensor(Ts), video_lens=torch.tensor(Tv), tokens=tokens)
    for K in beams:
        fl = force and K == 1
        ids, lens = eng.generate(task, video, Tv, audio, Ts, ra, rv, beams=K, max_new=spec.max_dec_tokens, force_len=fl)
        ids, lens = ids.cpu(), lens.cpu()
        with torch.no_grad():
            ref = orc.forward(dict(batch), task, ra, rv, num_beams=K, force_len=fl)
        for b in range(B):
            got = ids[b, : int(lens[b])].tolist()
            assert got == ref[b], f"{task} K={K} utt {b}: got {got} want {ref[b]}"
    return eng, orc, video, audio


@pytest.mark.parametrize("task", ["audiovisual", "audio", "video"])
def test_qwen2_task_specific_plus_shared_lora(task):
    spec = tiny_spec(family="qwen2", head_dim=128, task_specific=True, shared=True)
    check_ids(spec, task, [12, 9], [7680, 5000], 4, 2, beams=(1, 4))


@pytest.mark.parametrize("task", ["audiovisual", "video"])
def test_llama_head_dim_128_st_lora_llama3_rope(task):
    spec = tiny_spec(family="llama", head_dim=128, task_specific=True, shared=True)
    assert spec.llm.rope_scaling is not None and spec.llm.head_dim == 128
    check_ids(spec, task, [12, 9], [7680, 5000], 4, 2, beams=(1, 15))


@pytest.mark.parametrize("ra,rv", [(8, 8), (16, 5), (2, 8)])
def test_compression_rates_5_8_16(ra, rv):
    spec = dataclasses.replace(tiny_spec(family="llama", task_specific=True, shared=True, matryoshka=True),
                               downsample_ratio_audio=[2, 4, 8, 16], downsample_ratio_video=[2, 5, 8])
    check_ids(spec, "audiovisual", [43, 29], [27520, 18560], ra, rv)        # 86 / 58 audio tokens; remainders dropped by the pooling


def test_whisper_small_shaped_asr_only():
    """configs[0]: ASR only, Whisper-small dimensions (12 layers x 768, 12 heads) in front of the LLM, one 4 s utterance."""
    spec = dataclasses.replace(tiny_spec(family="llama", task_specific=True, shared=True), whisper=EncoderSpec(768, 12, 12, 3072))
    eng, orc, video, audio = check_ids(spec, "audio", [100], [64000], 4, 2, beams=(1, 15), seed=4)
    _, enc, _, _ = eng.encode_audio(audio, [64000], 4, return_enc=True)
    with torch.no_grad():
        ref = orc.encode_audio(audio, 64000, 4, return_stages=True)[1]["enc"][0]
    err = (enc[0].float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"whisper-small-shaped encoder rel err {err:.2e}")
    assert err < 1e-3


def test_vsr_only_batch_16():
    """configs[2]: video branch only, 16 ragged utterances in one call."""
    spec = tiny_spec(family="llama", task_specific=True, shared=True)
    g = torch.Generator().manual_seed(2)
    Tv = [int(x) for x in torch.randint(5, 15, (16,), generator=g)]
    check_ids(spec, "video", Tv, [640 * t for t in Tv], 4, 2)
