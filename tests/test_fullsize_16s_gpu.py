"""-m gpu: oracle parity AT THE BENCHMARKED CLIP LENGTH (BASELINE configs[1]: 16 s clips => T_v = 400 frames, 256 000 samples, 800 -> 200 audio
tokens + 100 video tokens, prompt S = 313) on the full model sizes (AV-HuBERT-Large + Whisper-medium + Llama-3.2-1B, Omni-LoRA ST).
The path checked is modeling_OmniAVSR.py:308-323 on [1,400,1,88,88] / [1,256000,1] (encode_video :461-483, encode_audio :528-558,
prepare_inputs :397-458, generate :313-322).  tests/test_fullsize_gpu.py pins the same sizes on a 4 s clip; at 16 s the positional
convolution, both flash-attention kernels, the 13-row-tile AV-HuBERT GEMMs and the S = 313 prefill take tile counts and ragged edges that
no 4 s case reaches.

  * f32 parity mode, a ragged pair [400, 250] in ONE engine call: video / Whisper encoder activations of the 16 s clip within 1e-3 of the
    activation scale, the 32 greedy ids of BOTH utterances bit-exact, and the HF beam-search (K = 15, the reference's default decode,
    eval_OmniAVSR.py:215-226) ids of the 16 s clip bit-exact -- each against the CPU oracle run per utterance (the reference's batch 1);
  * bf16 (the benchmarked arithmetic) on the 16 s clip inside the bf16-emulation bounds of tests/test_bf16_emulation_gpu.py.
The oracle costs ~35-60 s per utterance on the GPU box's host cores."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from omni_avsr_amd.modeling import OmniEngine  # noqa: E402
from omni_avsr_amd.synth import synth_state_dict  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

from test_bf16_emulation_gpu import run_case  # noqa: E402
from test_fullsize_gpu import clips, full_spec  # noqa: E402


def test_16s_f32_encoders_greedy_and_beam15_match_the_oracle():
    spec = full_spec()
    sd = synth_state_dict(spec, seed=1234, device="cuda:0")
    eng = OmniEngine(spec, dtype="f32").load_state_dict(sd)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    Tv = [400, 250]
    video, audio, Ts = clips(Tv, seed=11)
    _, venc, _ = eng.encode_video(video, Tv, 4, return_enc=True)
    _, aenc, _, _ = eng.encode_audio(audio, Ts, 4, return_enc=True)
    ids, lens = eng.generate("audiovisual", video, Tv, audio, Ts, 4, 4, beams=1, max_new=32, force_len=True)
    v0, a0 = video[:1, :400].contiguous(), audio[:1, : Ts[0]].contiguous()
    ids_k, lens_k = eng.generate("audiovisual", v0, [400], a0, [Ts[0]], 4, 4, beams=15, max_new=32)
    got_k = ids_k[0, : int(lens_k[0])].cpu().tolist()
    orc = O.OmniOracle(sd_cpu, spec.oracle_cfg())
    c = spec.oracle_cfg()
    with torch.no_grad():
        for b, t in enumerate(Tv):
            vb, ab = video[b:b + 1, :t], audio[b:b + 1, : Ts[b]]
            v_feat, vs = orc.encode_video(vb, 4, return_stages=True)
            a_feat, as_ = orc.encode_audio(ab, Ts[b], 4, return_stages=True)
            if b == 0:      # the 16 s clip's encoder activations (north star: within 1e-3 in fp32)
                ev = (venc[:400].float().cpu() - vs["enc"][0]).abs().max().item() / vs["enc"][0].abs().max().item()
                ea = (aenc[0].float().cpu() - as_["enc"][0]).abs().max().item() / as_["enc"][0].abs().max().item()
                print(f"16 s clip, f32: video encoder rel err {ev:.2e}, whisper encoder rel err {ea:.2e}")
                assert ev < 1e-3 and ea < 1e-3
            emb = orc.prepare_inputs(dict(video=vb, audio=ab, lengths=torch.tensor([Ts[b]]), tokens=torch.full((1, 1), spec.ids["bos"])),
                                     "audiovisual", 4, 4, feats=(a_feat, v_feat))
            assert emb.shape[1] == 1 + 2 + (int(Ts[b] / 16000 * 50) // 4) + 2 + t // 4 + 8 and (t != 400 or emb.shape[1] == 313)
            ref = orc.llm.generate_greedy(emb, 32, c["ids"]["eos"], c["ids"]["pad"], "audiovisual", force_len=True)
            assert ids[b].cpu().tolist() == ref, f"utterance {b} (T_v = {t}): greedy ids differ from the oracle's"
            if b == 0:
                ref_k = orc.llm.generate_beam(emb, 32, 15, c["ids"]["eos"], c["ids"]["pad"], "audiovisual")
                assert got_k == ref_k, f"16 s clip, num_beams = 15: {got_k} vs oracle {ref_k}"
    eng.close()


def test_16s_clip_bf16_engine_matches_bf16_emulation():
    spec = full_spec()
    sd = synth_state_dict(spec, seed=1234, device="cuda:0")
    run_case(spec, sd, 400, 256000, 4, 4, seed=11, max_new=32)
