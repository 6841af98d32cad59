"""Drop-in for `Omni_AVSR.modeling_LlamaAVSR.AVSR_LLMs` (inference branch): the single-prompt predecessor of
`modeling_OmniAVSR` with the same encoders, compressor, projectors and LoRA-adapted LLM, hence the same kernels.  Differences
handled here (all glue):

  * one `prompt` string for every modality, tokenised at run time and embedded with the LLM's own table
    (modeling_LlamaAVSR.py:283-287) instead of three stored prompt buffers;
  * LoRA is never task specific (`LoRA_config(RANK, ALPHA, ...)`, `lora_down_Q` / `lora_up_Q` / `lora_down_V` / `lora_up_V`);
  * the projector LayerNorm is a real `nn.LayerNorm` at Sequential index 3 unless `remove_layernorm_from_projector`
    (modeling_LlamaAVSR.py:60-108) -- none of the Omni variant's quirks;
  * `forward(inputs, is_trainval, test_ratio_matry)`: for audiovisual matryoshka models `test_ratio_matry` is the pair
    (video rate, audio rate) (modeling_LlamaAVSR.py:334,339), otherwise one int.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .config import OmniSpec, spec_from_names
from .modeling import AVSR_LLMs as _OmniShim, OmniEngine


class AVSR_LLMs(_OmniShim):
    def __init__(self, modality, pretrain_avhubert_enc_video, use_lora_avhubert, llm_model, hidden_size, intermediate_size, tokenizer,
                 prompt, pad_id, downsample_ratio_audio, downsample_ratio_video, audio_encoder_name, compression_mode,
                 unfrozen_modules, max_dec_tokens, num_beams, PETF_LLM_name=None, peft_config_llm=None,
                 remove_layernorm_from_projector=False, is_matryoshka=False, *, prompt_ids: Optional[Sequence[int]] = None,
                 spec: Optional[OmniSpec] = None, dtype: str = "bf16", device: str = "cuda:0"):
        self.modality = modality
        self._omni_matry_map_quirk = False      # modeling_LlamaAVSR.py's non-matryoshka inference branch has its own projector call
        self.max_dec_tokens, self.num_beams = max_dec_tokens, num_beams
        self.llm_model, self.tokenizer, self.prompt = llm_model, tokenizer, prompt
        self.is_matryoshka, self.is_task_specific = bool(is_matryoshka), False
        self.downsample_ratio_audio, self.downsample_ratio_video = downsample_ratio_audio, downsample_ratio_video
        # prompt token ids: tokenizer(prompt).input_ids[:, start:-1] with start = 0 (Qwen) / 1 (Llama: drop BOS)   (:283-285)
        if prompt_ids is None:
            if tokenizer is None:
                raise ValueError("modeling_llamaavsr.AVSR_LLMs needs a tokenizer or explicit prompt_ids (no hub access offline)")
            ids = tokenizer(prompt, return_tensors="pt").input_ids[0].tolist()
            prompt_ids = ids[(0 if "Qwen" in llm_model else 1):-1]
        self.prompt_ids = [int(t) for t in prompt_ids]
        if spec is None:
            rank = alpha = 0
            if PETF_LLM_name is not None:
                assert PETF_LLM_name == "lora"                                              # modeling_LlamaAVSR.py:193
                rank, alpha = peft_config_llm.RANK, peft_config_llm.ALPHA
            as_list = lambda r: list(r) if isinstance(r, (list, tuple)) else [r]
            spec = spec_from_names(
                llm_model, audio_encoder_name if modality in ("audio", "audiovisual") else None,
                pretrain_avhubert_enc_video if modality in ("video", "audiovisual") else None,
                use_lora_avhubert=bool(use_lora_avhubert), intermediate_size=intermediate_size, compression_mode=compression_mode,
                downsample_ratio_audio=as_list(downsample_ratio_audio) if is_matryoshka else downsample_ratio_audio,
                downsample_ratio_video=as_list(downsample_ratio_video) if is_matryoshka else downsample_ratio_video,
                is_matryoshka=bool(is_matryoshka), is_single_matry_projector=False,
                remove_layernorm_from_projector=bool(remove_layernorm_from_projector), projector_layernorm_plain=True,
                lora_rank=rank, lora_alpha=alpha, is_task_specific=False, shared_lora=False,
                prompt_len={t: len(self.prompt_ids) for t in ("audio", "video", "audiovisual")},
                max_dec_tokens=max_dec_tokens, num_beams=num_beams)
            assert spec.llm.hidden == hidden_size, "hidden_size does not match the LLM"
            if tokenizer is not None:
                v = tokenizer.vocab
                spec.ids.update(audio_sos=v["<audio>"], audio_eos=v["</audio>"], video_sos=v["<video>"], video_eos=v["</video>"])
        self.spec = spec
        self.engine = OmniEngine(spec, dtype=dtype, device=device)

    def load_state_dict(self, state_dict, strict: bool = True):
        # the reference embeds the prompt on every call; embedding it once at load time is the same lookup
        sd = dict(state_dict)
        table = sd["llm.model.embed_tokens.weight"]
        emb = table[torch.tensor(self.prompt_ids, dtype=torch.long, device=table.device)].unsqueeze(0)
        for t in ("audio", "video", "audiovisual"):
            sd.setdefault(f"prompt_{t}", emb)
        self.engine.load_state_dict(sd)
        return self

    def forward(self, inputs, is_trainval=True, test_ratio_matry=None, force_len: bool = False):
        ra = rv = None
        if self.is_matryoshka:
            if self.modality == "audiovisual":
                rv, ra = test_ratio_matry[0], test_ratio_matry[1]                          # modeling_LlamaAVSR.py:334,339
            elif self.modality == "audio":
                ra = test_ratio_matry
            else:
                rv = test_ratio_matry
        return super().forward(inputs, is_trainval=is_trainval, modality=None, test_ratio_matry_audio=ra, test_ratio_matry_video=rv,
                               force_len=force_len)
