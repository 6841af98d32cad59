import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_fullsize_gpu import full_spec, clips
from omni_avsr_amd.modeling import OmniEngine
from omni_avsr_amd.synth import synth_state_dict
spec = full_spec()
sd = synth_state_dict(spec, seed=1234, device="cuda:0", dtype=torch.bfloat16)
eng = OmniEngine(spec, dtype="bf16").load_state_dict(sd)
del sd
Tv = [400, 250, 100, 61]
video, audio, Ts = clips(Tv, seed=9)
_, e1, _ = eng.encode_video(video, Tv, 4, return_enc=True)
_, e2, _ = eng.encode_video(video, Tv, 4, return_enc=True)
print("video batch run-to-run equal:", torch.equal(e1, e2), (e1.float() - e2.float()).abs().max().item())
_, s1, _ = eng.encode_video(video[:1, :400], [400], 4, return_enc=True)
_, s2, _ = eng.encode_video(video[:1, :400], [400], 4, return_enc=True)
print("video single run-to-run equal:", torch.equal(s1, s2))
print("single vs batch:", torch.equal(s1, e1[:400]), (s1.float() - e1[:400].float()).abs().max().item())
_, a1, _, _ = eng.encode_audio(audio, Ts, 4, return_enc=True)
_, a2, _, _ = eng.encode_audio(audio, Ts, 4, return_enc=True)
print("audio batch run-to-run equal:", torch.equal(a1, a2))
_, b1, _, _ = eng.encode_audio(audio[:1, :Ts[0]], Ts[:1], 4, return_enc=True)
print("audio single vs batch:", torch.equal(b1[0], a1[0]), (b1[0].float() - a1[0].float()).abs().max().item())
