"""Python shim that keeps the reference's boundary: `AVSR_LLMs(...)`, `.load_state_dict(ckpt)`, `.modality`,
`.forward(inputs, is_trainval=False, modality=None, test_ratio_matry_audio=None, test_ratio_matry_video=None)`
(Omni_AVSR/modeling_OmniAVSR.py:27-34,263,308-323).  All tensor work happens in libomniavsr_hip.so; torch is only
the owner of device memory and of the HIP stream.

Differences from the reference that are deliberate:
  * batches > 1 are allowed; every utterance is computed exactly as if it were alone (no padded attention), which is
    what the reference's batch-1 evaluation does.  Optional `inputs["video_lengths"]` gives frames per utterance.
  * decoding is deterministic (`do_sample=False`): `num_beams == 1` greedy (the north-star decode) or HF-equivalent beam
    search on device (`num_beams > 1`, the reference's default 15; csrc/beam.hip).
  * `is_trainval=True` (loss computation) is training and out of scope: NotImplementedError.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import lib as L
from .config import (LLM_SPECS, LoRA_config, OmniSpec, QwenLoRA_config, spec_from_names)
from .weights import av_lora_k2, build_blob, lora_k2, pack_state_dict


def _i32arr(xs: Sequence[int]):
    return (C.c_int32 * len(xs))(*[int(x) for x in xs])


def whisper_keep_tokens(max_len: int) -> int:
    """modeling_OmniAVSR.py:537 — `audio_enc[:, :max(int(max_len/16000*50), 25)]` evaluated with Python float semantics; the
    slice of the 1500-frame encoder output simply stops at 1500 for clips longer than 30 s."""
    return min(max(int(max_len / 16000 * 50), 25), 1500)


_CALLER_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}      # (device, i) -> the engine's caller stream i (process lifetime, see OmniEngine.caller_stream)


class OmniEngine:
    """Owns the device blob, the oavsr_model and one workspace; thin typed wrappers over the C ABI."""

    def __init__(self, spec: OmniSpec, dtype: str = "bf16", device: str = "cuda:0"):
        self.spec, self.dtype, self.device = spec, dtype, torch.device(device)
        self.tdtype = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.lib = L.load()
        self._model = C.c_void_p(None)
        self._ws = C.c_void_p(None)
        self._ws_key = None
        self._slots = {}            # extra workspaces (one per concurrent stream): slot -> (ptr, key)
        self._sched = C.c_void_p(None)   # decode scheduler (continuous batching of the greedy decode across calls); created on first submit
        self._sched_key = None
        self._sched_gen = 0              # tickets carry the generation of the pool they belong to (a pool that had to grow is rebuilt)
        # rows per decode step (<= 128) / row slots of the pool / independent step loops sharing the pool.  Every decode-shaped GEMM runs on the
        # tiled LDS-DMA kernels (model.hip: llm_decode_layers): a step is ONE 128-row tile whatever its row count and streams every LLM weight
        # once, so the more rows share it the better -- 128-row steps, three lanes, for every model
        # 1024 slots (round 5; 512 before): calls whose prefill is done wait in the pool for a lane -- with cheaper encoders 512 slots were the
        # queue's limit (headline 337-340 -> 345-346 utt/s; Llama-3.2-1B at 16 s clips: 11.7 GB, Llama-3.1-8B: _fit_slots caps it at a quarter of the device)
        self.sched_rows, self.sched_slots = 128, 1024
        self.sched_lanes = 3
        self.sched_stride = 0                           # steps per admission (0: automatic; oavsr_scheduler_set_stride)
        self._blob = None
        self._names = None

    # ---- construction ------------------------------------------------------------------------------
    def _config(self, packed_names) -> L.OavsrConfig:
        s, l = self.spec, self.spec.llm
        c = L.OavsrConfig()
        c.dtype = L.DT_BF16 if self.dtype == "bf16" else L.DT_F32
        c.llm_family = 0 if l.family == "llama" else 1
        c.llm_hidden, c.llm_layers, c.llm_heads, c.llm_kv_heads = l.hidden, l.layers, l.heads, l.kv_heads
        c.llm_head_dim, c.llm_mlp, c.llm_vocab, c.llm_rms_eps, c.llm_qkv_bias = l.head_dim, l.mlp, l.vocab, l.rms_eps, int(l.qkv_bias)
        c.lora_inner, c.lora_k2, c.lora_task_specific = s.lora_inner(), lora_k2(s), int(s.is_task_specific)
        if s.whisper is not None:
            c.wh_dim, c.wh_layers, c.wh_heads, c.wh_ffn = s.whisper.dim, s.whisper.layers, s.whisper.heads, s.whisper.ffn
        if s.avhubert is not None:
            a = s.avhubert
            c.av_dim, c.av_layers, c.av_heads, c.av_ffn = a.dim, a.layers, a.heads, a.ffn
            c.av_conv_pos, c.av_conv_groups, c.av_lora_k2 = a.conv_pos, a.conv_pos_groups, av_lora_k2(s)
        c.proj_inter = s.intermediate_size
        c.compression_stack = int(s.compression_mode == "stack")
        ra, rv = s.rates("audio"), s.rates("video")
        c.n_rates_audio, c.n_rates_video = len(ra), len(rv)
        for i, r in enumerate(ra):
            c.rates_audio[i] = r
        for i, r in enumerate(rv):
            c.rates_video[i] = r
        c.proj_per_rate = int(s.proj_per_rate())
        c.proj_layernorm_audio = int("pa.0.ln.w" in packed_names)
        c.proj_layernorm_video = int("pv.0.ln.w" in packed_names)
        ids = s.ids
        c.id_bos, c.id_eos, c.id_pad = ids["bos"], ids["eos"], ids["pad"]
        c.id_audio_sos, c.id_audio_eos, c.id_video_sos, c.id_video_eos = ids["audio_sos"], ids["audio_eos"], ids["video_sos"], ids["video_eos"]
        for t, name in enumerate(("audio", "video", "audiovisual")):
            c.prompt_len[t] = s.prompt_len[name] if f"l.prompt.{t}" in packed_names else 0
        return c

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Accepts the reference's checkpoint layout (strict for the modules the spec declares)."""
        if not torch.cuda.is_available():
            raise RuntimeError("omni_avsr_amd needs a ROCm GPU (MI355X); there is no CPU path")
        l = self.spec.llm
        for key, want in (("llm.model.embed_tokens.weight", (l.vocab, l.hidden)), ("llm.lm_head.weight", (l.vocab, l.hidden))):
            if key in sd and tuple(sd[key].shape) != want:
                # what nn.Module.load_state_dict(strict=True) raises in the reference (lightning_OmniAVSR.py:150): the vocabulary must be
                # the RESIZED one (modeling_OmniAVSR.py:214, +5 / +4 rows); build the spec with ingest.infer_spec to take it from the file
                raise RuntimeError(f"size mismatch for {key}: copying a param with shape {tuple(sd[key].shape)} from checkpoint, "
                                   f"the shape in current model is {want}")
        packed = pack_state_dict(sd, self.spec, self.dtype)
        # keep the true prompt lengths of the checkpoint
        for t, name in enumerate(("audio", "video", "audiovisual")):
            if f"l.prompt.{t}" in packed:
                self.spec.prompt_len[name] = int(packed[f"l.prompt.{t}"].shape[0])
        blob, manifest = build_blob(packed, device=self.device)
        self._blob = blob
        self._names = [n.encode() for n, _, _ in manifest]
        arr = (L.OavsrTensor * len(manifest))()
        for i, (n, off, nb) in enumerate(manifest):
            arr[i].name, arr[i].offset, arr[i].nbytes = self._names[i], off, nb
        cfg = self._config({n for n, _, _ in manifest})
        self.close()
        torch.cuda.synchronize(self.device)
        L.check(self.lib.oavsr_model_create(C.byref(cfg), C.c_void_p(blob.data_ptr()), blob.numel(), arr, len(manifest), C.byref(self._model)))
        self.blob_bytes = blob.numel()
        return self

    def ensure_workspace(self, B: int, Tv: int, Ts: int, beams: int = 1, max_new: Optional[int] = None):
        max_new = max_new or self.spec.max_dec_tokens
        key = self._ws_key
        # beam search indexes the tail cache / ancestry table with the workspace's max_new stride: needs an exact match
        if key is not None and key[0] >= B and key[1] >= Tv and key[2] >= Ts and key[3] >= beams and (key[4] == max_new or (beams == 1 and key[4] >= max_new)):
            return
        if self._ws:
            torch.cuda.synchronize(self.device)
            self.lib.oavsr_workspace_destroy(self._ws)
            self._ws = C.c_void_p(None)
        L.check(self.lib.oavsr_workspace_create(self._model, B, max(Tv, 1), max(Ts, 1), beams, max_new, C.byref(self._ws)))
        self._ws_key = (B, Tv, Ts, beams, max_new)

    def launch_objective(self, alone: bool):
        """What the GEMM dispatcher optimises (process-wide; include/omni_avsr.h: oavsr_set_launch_objective).  False (default): calls share
        the chip -- coarse tiles on as many CUs as they have tiles, least CU-time.  True: one call at a time on an idle chip (the reference's
        evaluation protocol) -- finer tiles that finish sooner.  Same bits either way.  Returns the PREVIOUS objective: the setting is global
        to the process, so a caller that changes it for a phase restores what it found (`prev = launch_objective(True) ... launch_objective(prev)`)."""
        prev = bool(self.lib.oavsr_get_launch_objective())
        L.check(self.lib.oavsr_set_launch_objective(int(bool(alone))))
        return prev

    def serialize_encoders(self, on: bool = True):
        """Several generate() calls in flight (slots): run their encoder + prefill phases back to back, overlap only decode."""
        L.check(self.lib.oavsr_model_serialize_encoders(self._model, int(on)))
        return self

    def workspace_bytes(self) -> int:
        return int(self.lib.oavsr_workspace_bytes(self._ws)) if self._ws else 0

    def _slot_workspace(self, slot: int, B: int, Tv: int, Ts: int, beams: int, max_new: int):
        """slot 0 is the default workspace; slots >= 1 are independent workspaces for pipelining batches on other streams
        (one oavsr_workspace per stream, include/omni_avsr.h)."""
        if slot == 0:
            self.ensure_workspace(B, Tv, Ts, beams, max_new)
            return self._ws
        ptr, key = self._slots.get(slot, (None, None))
        if key is not None and key[0] >= B and key[1] >= Tv and key[2] >= Ts and key[3] >= beams and (key[4] == max_new or (beams == 1 and key[4] >= max_new)):
            return ptr
        if ptr:
            torch.cuda.synchronize(self.device)
            self.lib.oavsr_workspace_destroy(ptr)
        ptr = C.c_void_p(None)
        L.check(self.lib.oavsr_workspace_create(self._model, B, max(Tv, 1), max(Ts, 1), beams, max_new, C.byref(ptr)))
        self._slots[slot] = (ptr, (B, Tv, Ts, beams, max_new))
        return ptr

    def reserve(self, slots: int, B: int, Tv: int, Ts: int, beams: int = 1, max_new: Optional[int] = None):
        """Size the workspaces of slots 0..slots-1 once (e.g. for the longest clip of a dataset): later calls that fit reuse them, so no
        workspace is re-created while other slots have work in flight."""
        for s_ in range(slots):
            self._slot_workspace(s_, B, Tv, Ts, beams, max_new or self.spec.max_dec_tokens)
        return self

    def close(self):
        if self._sched:
            self.lib.oavsr_scheduler_destroy(self._sched)
            self._sched = C.c_void_p(None)
            self._sched_key = None
        for ptr, _ in self._slots.values():
            if ptr:
                self.lib.oavsr_workspace_destroy(ptr)
        self._slots = {}
        if self._ws:
            self.lib.oavsr_workspace_destroy(self._ws)
            self._ws = C.c_void_p(None)
            self._ws_key = None
        if self._model:
            self.lib.oavsr_model_destroy(self._model)
            self._model = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _device_video(self, ws, video: torch.Tensor, Tv: Sequence[int], video_is_flat: bool) -> torch.Tensor:
        """The video operand of a call on workspace `ws`.  float frames (after the dataset's test-time transform): f32 [sum T, 88, 88].
        uint8 frames (DECODED media, `video_is_flat` only: [sum T, C, H, W], C = 1 or 3): uploaded as they are, the transform runs inside
        the stem kernel (oavsr_workspace_set_video_format)."""
        if video.dtype == torch.uint8:
            if not video_is_flat or video.dim() != 4:
                raise ValueError("uint8 video must be the flat [sum T, C, H, W] frames of the batch (harness.FlatCollator: 'video_u8_flat')")
            flat = video if video.device == self.device and video.is_contiguous() else video.to(self.device, non_blocking=video.is_pinned()).contiguous()
            L.check(self.lib.oavsr_workspace_set_video_format(ws, int(flat.shape[1]), int(flat.shape[2]), int(flat.shape[3])))
            return flat
        flat = video if video_is_flat else self._flat_video(video, Tv)
        if flat.device != self.device or flat.dtype != torch.float32 or not flat.is_contiguous():
            flat = flat.to(self.device, torch.float32, non_blocking=flat.is_pinned()).contiguous()     # pinned staging buffers copy asynchronously
        L.check(self.lib.oavsr_workspace_set_video_format(ws, 0, 88, 88))
        return flat

    # ---- stage wrappers (parity tests call these) ----------------------------------------------------
    def _flat_video(self, video: torch.Tensor, Tv: Sequence[int]) -> torch.Tensor:
        """[B,T,1,88,88] (padded) -> contiguous f32 [sum(Tv),88,88]"""
        v = video.to(self.device, torch.float32)
        return torch.cat([v[b, : Tv[b], 0] for b in range(v.shape[0])], 0).contiguous()

    def encode_video(self, video: torch.Tensor, Tv: Sequence[int], rate: int, return_enc: bool = False):
        B = len(Tv)
        self.ensure_workspace(max(B, self._ws_key[0] if self._ws_key else 0), max(max(Tv), self._ws_key[1] if self._ws_key else 0), self._ws_key[2] if self._ws_key else 16000)
        flat = self._device_video(self._ws, video, Tv, video.dtype == torch.uint8)
        D = self.spec.avhubert.dim
        n_out = [t // rate for t in Tv]
        Dout = D * rate if self.spec.compression_mode == "stack" else D
        out = torch.empty(sum(n_out), Dout, dtype=self.tdtype, device=self.device)
        enc = torch.empty(sum(Tv), D, dtype=self.tdtype, device=self.device) if return_enc else None
        L.check(self.lib.oavsr_encode_video(self._model, self._ws, C.c_void_p(flat.data_ptr()), _i32arr(Tv), B, rate, C.c_void_p(out.data_ptr()),
                                            C.c_void_p(enc.data_ptr()) if enc is not None else None, self._stream()))
        return (out, enc, n_out) if return_enc else (out, n_out)

    def encode_audio(self, audio: torch.Tensor, lengths: Sequence[int], rate: int, return_enc: bool = False, keep: Optional[Sequence[int]] = None):
        """audio [B,T_s] or [B,T_s,1] f32; lengths = valid samples per utterance."""
        a = audio.to(self.device, torch.float32)
        if a.dim() == 3:
            a = a[..., 0]
        a = a.contiguous()
        B = a.shape[0]
        keep = list(keep) if keep is not None else [whisper_keep_tokens(int(n)) for n in lengths]
        self.ensure_workspace(max(B, self._ws_key[0] if self._ws_key else 0), self._ws_key[1] if self._ws_key else 1, max(int(max(lengths)), self._ws_key[2] if self._ws_key else 0))
        D = self.spec.whisper.dim
        n_out = [k // rate for k in keep]
        Dout = D * rate if self.spec.compression_mode == "stack" else D
        out = torch.empty(sum(n_out), Dout, dtype=self.tdtype, device=self.device)
        enc = torch.empty(B, 1500, D, dtype=self.tdtype, device=self.device) if return_enc else None
        mel = torch.empty(B, 3000, 80, dtype=self.tdtype, device=self.device) if return_enc else None
        L.check(self.lib.oavsr_encode_audio(self._model, self._ws, C.c_void_p(a.data_ptr()), a.stride(0), _i32arr(lengths), _i32arr(keep), B, rate,
                                            C.c_void_p(out.data_ptr()), C.c_void_p(enc.data_ptr()) if enc is not None else None,
                                            C.c_void_p(mel.data_ptr()) if mel is not None else None, self._stream()))
        return (out, enc, mel, n_out) if return_enc else (out, n_out)

    def project(self, which: str, rate: int, feats: torch.Tensor) -> torch.Tensor:
        out = torch.empty(feats.shape[0], self.spec.llm.hidden, dtype=self.tdtype, device=self.device)
        L.check(self.lib.oavsr_project(self._model, self._ws, 0 if which == "audio" else 1, rate, C.c_void_p(feats.data_ptr()), feats.shape[0],
                                       C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def build_prompt(self, task: str, B: int, na, atok, nv, vtok):
        P = self.spec.prompt_len[task]
        bos = 1 if self.spec.llm.family == "llama" else 0
        S = [bos + (na[b] + 2 if na is not None else 0) + (nv[b] + 2 if nv is not None else 0) + P for b in range(B)]
        emb = torch.empty(sum(S), self.spec.llm.hidden, dtype=self.tdtype, device=self.device)
        S_out = (C.c_int32 * B)()
        L.check(self.lib.oavsr_build_prompt(self._model, self._ws, L.TASK_ID[task], B, _i32arr(na) if na is not None else None,
                                            C.c_void_p(atok.data_ptr()) if atok is not None else None, _i32arr(nv) if nv is not None else None,
                                            C.c_void_p(vtok.data_ptr()) if vtok is not None else None, C.c_void_p(emb.data_ptr()), S_out, self._stream()))
        assert list(S_out) == S
        return emb, S

    def prefill(self, task: str, embeds: torch.Tensor, S: Sequence[int]) -> torch.Tensor:
        B = len(S)
        logits = torch.empty(B, self.spec.llm.vocab, dtype=torch.float32, device=self.device)
        L.check(self.lib.oavsr_prefill(self._model, self._ws, L.TASK_ID[task], B, C.c_void_p(embeds.data_ptr()), _i32arr(S), C.c_void_p(logits.data_ptr()), self._stream()))
        return logits

    def decode(self, task: str, B: int, beams: int, max_new: int, force_len: bool = False):
        ids = torch.empty(B, max_new, dtype=torch.int32, device=self.device)
        lens = torch.empty(B, dtype=torch.int32, device=self.device)
        L.check(self.lib.oavsr_decode(self._model, self._ws, L.TASK_ID[task], B, beams, max_new, int(force_len), C.c_void_p(ids.data_ptr()),
                                      C.c_void_p(lens.data_ptr()), self._stream()))
        return ids, lens

    def generate(self, task: str, video: Optional[torch.Tensor], Tv: Optional[Sequence[int]], audio: Optional[torch.Tensor],
                 lengths: Optional[Sequence[int]], rate_a: int, rate_v: int, beams: int = 1, max_new: Optional[int] = None,
                 force_len: bool = False, video_is_flat: bool = False, slot: int = 0, chunk: Optional[int] = None,
                 digests: Optional[torch.Tensor] = None):
        """End-to-end `forward(is_trainval=False)`: returns (ids int32 [B,max_new] pad-filled, lens int32 [B]) on device.
        No host synchronisation happens inside when the current stream is a stream of the caller's; on the legacy default stream the call is complete on return.  `chunk`: run encoders / prefill in chunks of that many utterances and decode all B
        rows together (oavsr_generate_chunked: same results, the LLM weights are streamed once per step for the whole group).
        `digests`: optional int64 [B, 2] device tensor the call fills with {inputs_embeds, first-step logits} digests per utterance
        (oavsr_workspace_set_digests: input-sensitivity checks of bench.py / tests)."""
        max_new = max_new or self.spec.max_dec_tokens
        use_a, use_v = task in ("audio", "audiovisual"), task in ("video", "audiovisual")
        B = len(Tv) if use_v else len(lengths)
        flat = a = None
        keep = None
        ws = self._slot_workspace(slot, B, max(Tv) if use_v else 1, int(max(lengths)) if use_a else 16000, beams, max_new)
        if use_v:
            flat = self._device_video(ws, video, Tv, video_is_flat)
        if use_a:
            a = audio if audio.dim() == 2 else audio[..., 0]
            if a.device != self.device or a.dtype != torch.float32 or not a.is_contiguous():
                a = a.to(self.device, torch.float32, non_blocking=a.is_pinned()).contiguous()
            keep = [whisper_keep_tokens(int(n)) for n in lengths]
        ids = torch.empty(B, max_new, dtype=torch.int32, device=self.device)
        lens = torch.empty(B, dtype=torch.int32, device=self.device)
        self._set_digests(ws, digests, B)
        # A call issued on the legacy default stream (what a caller gets who never sets a stream: the reference's protocol) runs on an engine-owned
        # stream instead, ordered behind the default stream's work and completed before generate() returns: the default stream cannot be captured
        # (no hipGraph replay of the decode steps) and work queued on or behind it finishes ~4 ms later per call (B = 1, 16 s clip: 52.3 vs 48.2 ms).
        cur = torch.cuda.current_stream(self.device)
        side = None
        if cur.cuda_stream == 0:
            side = self.__dict__.get("_graph_stream")
            if side is None:
                side = self.__dict__["_graph_stream"] = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
        run_on = side if side is not None else cur
        with torch.cuda.stream(run_on):
            L.check(self.lib.oavsr_generate_chunked(
                self._model, ws, L.TASK_ID[task],
                C.c_void_p(flat.data_ptr()) if use_v else None, _i32arr(Tv) if use_v else None,
                C.c_void_p(a.data_ptr()) if use_a else None, a.stride(0) if use_a else 0,
                _i32arr(lengths) if use_a else None, _i32arr(keep) if use_a else None,
                B, int(chunk or B), int(rate_a or 1), int(rate_v or 1), beams, max_new, int(force_len),
                C.c_void_p(ids.data_ptr()), C.c_void_p(lens.data_ptr()), self._stream()), after=lambda: self._set_digests(ws, None, B))
        if side is not None:
            for t in (ids, lens, flat, a, digests):
                if t is not None and t.is_cuda:
                    t.record_stream(side)
            # ... by waiting for the call on the HOST: an event wait queued on the legacy default stream keeps the device busy ~4 ms longer per call
            # (measured: 52.3 vs 48.2 ms, tools/probe/latency_split.py); the synchronous forward() of the reference protocol blocks here anyway
            side.synchronize()
        self._keepalive = getattr(self, "_keepalive", {})
        self._keepalive[slot] = (flat, a)
        return ids, lens

    def _set_digests(self, ws, digests: Optional[torch.Tensor], B: int):
        if digests is None:
            L.check(self.lib.oavsr_workspace_set_digests(ws, None))
            return
        if digests.dtype != torch.int64 or tuple(digests.shape) != (B, 2) or digests.device != self.device or not digests.is_contiguous():
            raise ValueError("digests must be a contiguous int64 [B, 2] tensor on the engine's device")
        L.check(self.lib.oavsr_workspace_set_digests(ws, C.c_void_p(digests.data_ptr())))

    # ---- decode scheduler: calls of any size keep arriving one at a time, their decode rows share the weight passes ----------
    def prompt_rows(self, task: str, Tv, lengths, rate_a: int, rate_v: int) -> int:
        """Longest prompt (rows) of a call: [BOS] <audio> A </audio> <video> V </video> prompt (modeling_OmniAVSR.py:434-437,452-455)."""
        use_a, use_v = task in ("audio", "audiovisual"), task in ("video", "audiovisual")
        n = len(Tv) if use_v else len(lengths)
        bos = 1 if self.spec.llm.family == "llama" else 0
        P = self.spec.prompt_len[task]
        return max(bos + (whisper_keep_tokens(int(lengths[b])) // rate_a + 2 if use_a else 0) + (int(Tv[b]) // rate_v + 2 if use_v else 0) + P for b in range(n))

    def ensure_scheduler(self, max_prefix: int, max_new: int):
        key = self._sched_key
        if key is not None and key[0] >= max_prefix and key[1] >= max_new and key[2:] == (self.sched_rows, self.sched_slots, self.sched_lanes):
            return
        if self._sched:                                  # grow: finish what is in flight, then rebuild the pool
            L.check(self.lib.oavsr_scheduler_flush(self._sched))
            torch.cuda.synchronize(self.device)
            self.lib.oavsr_scheduler_destroy(self._sched)
            self._sched = C.c_void_p(None)
            max_prefix, max_new = max(max_prefix, key[0]), max(max_new, key[1])
        max_prefix = (max_prefix + 63) // 64 * 64
        slots = self._fit_slots(max_prefix, max_new)
        L.check(self.lib.oavsr_scheduler_create(self._model, slots, self.sched_rows, self.sched_lanes, max_prefix, max_new, C.byref(self._sched)))
        self._sched_key = (max_prefix, max_new, self.sched_rows, self.sched_slots, self.sched_lanes)
        self._sched_gen += 1
        self._stride_set = 0

    def _fit_slots(self, max_prefix: int, max_new: int) -> int:
        """Row slots the pool is built with: `sched_slots`, cut down (never below one step's rows) so that the pool -- per slot the prefix and tail
        KV rows of every layer plus one fp32 logits row -- stays under a quarter of the device's memory (Llama-3.1-8B, 16 s clips: ~42 MB per
        slot, 1024 slots = 43 GB: fine on 288 GB, not on a small board or with several engines per process)."""
        l = self.spec.llm
        esz = 2 if self.dtype == "bf16" else 4
        per_slot = 2 * l.layers * l.kv_heads * l.head_dim * (max_prefix + max_new) * esz + l.vocab * 4
        cap = torch.cuda.get_device_properties(self.device).total_memory // 4
        return int(max(self.sched_rows, min(self.sched_slots, cap // max(1, per_slot))))

    def scheduler_bytes(self) -> int:
        """Device bytes of the decode scheduler's pool (row slots' KV + logits, the lanes' step scratch); 0 before the first submit."""
        return int(self.lib.oavsr_scheduler_bytes(self._sched)) if self._sched else 0

    def reserve_scheduler(self, task: str, tv_max: int, ts_max: int, rate_a: Optional[int] = None, rate_v: Optional[int] = None,
                          max_new: Optional[int] = None):
        """Size the decode scheduler's pool once for the longest clip of a run (rates default to the smallest configured ones = the longest
        prompt): a pool that has to grow in the middle of an evaluation drains every lane, synchronises the device and re-captures the step graphs."""
        use_a, use_v = task in ("audio", "audiovisual"), task in ("video", "audiovisual")
        ra = int(rate_a or (min(self.spec.rates("audio")) if use_a else 1))
        rv = int(rate_v or (min(self.spec.rates("video")) if use_v else 1))
        self.ensure_scheduler(self.prompt_rows(task, [int(tv_max)], [int(ts_max)], ra, rv), max_new or self.spec.max_dec_tokens)
        return self

    def caller_stream(self, i: int) -> "torch.cuda.Stream":
        """Caller stream i (0..7) of the decode scheduler, created by the ENGINE before its lanes' streams (oavsr_scheduler_caller_stream):
        submit calls on these and the loop's throughput no longer depends on when the caller happened to create its streams (ROCm binds a
        stream to a hardware queue at creation).  Needs the scheduler (reserve_scheduler / a first submit); the
        streams belong to the process (one per GPU): they are never destroyed, every engine of the process hands out the same eight."""
        if not self._sched:
            raise RuntimeError("caller_stream: the decode scheduler does not exist yet (call reserve_scheduler first)")
        cache = _CALLER_STREAMS
        key = (str(self.device), int(i))
        if key not in cache:
            ptr = self.lib.oavsr_scheduler_caller_stream(self._sched, int(i))
            if not ptr:
                raise IndexError(L.last_error())
            cache[key] = torch.cuda.ExternalStream(ptr, device=self.device)
        return cache[key]

    def submit(self, task: str, video, Tv, audio, lengths, rate_a: int, rate_v: int, max_new: Optional[int] = None, force_len: bool = False,
               video_is_flat: bool = False, slot: int = 0, digests: Optional[torch.Tensor] = None):
        """`generate(beams=1)` through the decode scheduler: encoders + prefill are queued on the current stream (workspace `slot`), the
        rows then join the engine's shared decode loop.  Returns (ids, lens, ticket); ids / lens are valid after `wait(ticket)`."""
        max_new = max_new or self.spec.max_dec_tokens
        use_a, use_v = task in ("audio", "audiovisual"), task in ("video", "audiovisual")
        B = len(Tv) if use_v else len(lengths)
        flat = a = keep = None
        ws = self._slot_workspace(slot, B, max(Tv) if use_v else 1, int(max(lengths)) if use_a else 16000, 1, max_new)
        if use_v:
            flat = self._device_video(ws, video, Tv, video_is_flat)
        if use_a:
            a = audio if audio.dim() == 2 else audio[..., 0]
            if a.device != self.device or a.dtype != torch.float32 or not a.is_contiguous():
                a = a.to(self.device, torch.float32, non_blocking=a.is_pinned()).contiguous()
            keep = [whisper_keep_tokens(int(n)) for n in lengths]
        self.ensure_scheduler(self.prompt_rows(task, Tv, lengths, int(rate_a or 1), int(rate_v or 1)), max_new)
        if self.sched_stride != self._stride_set:
            L.check(self.lib.oavsr_scheduler_set_stride(self._sched, int(self.sched_stride)))
            self._stride_set = self.sched_stride
        ids = torch.empty(B, max_new, dtype=torch.int32, device=self.device)
        lens = torch.empty(B, dtype=torch.int32, device=self.device)
        host = torch.empty(B * (max_new + 1), dtype=torch.int32, pin_memory=True)      # [ids | lens], filled by the scheduler's stream when the call leaves
        ticket = C.c_int64(-1)
        self._set_digests(ws, digests, B)
        L.check(self.lib.oavsr_generate_submit(
            self._model, self._sched, ws, L.TASK_ID[task],
            C.c_void_p(flat.data_ptr()) if use_v else None, _i32arr(Tv) if use_v else None,
            C.c_void_p(a.data_ptr()) if use_a else None, a.stride(0) if use_a else 0,
            _i32arr(lengths) if use_a else None, _i32arr(keep) if use_a else None,
            B, int(rate_a or 1), int(rate_v or 1), max_new, int(force_len),
            C.c_void_p(ids.data_ptr()), C.c_void_p(lens.data_ptr()), C.c_void_p(host.data_ptr()), C.c_void_p(host.data_ptr() + 4 * B * max_new),
            self._stream(), C.byref(ticket)), after=lambda: self._set_digests(ws, None, B))
        self._keepalive = getattr(self, "_keepalive", {})
        self._keepalive[("s", slot)] = (flat, a, digests)
        self._inflight = getattr(self, "_inflight", {})
        tk = (self._sched_gen, ticket.value)
        self._reap()
        self._inflight[tk] = [ids, lens, host, False]    # the scheduler writes them when the call's last step has run; [3]: wait() was called
        return ids, lens, tk

    def wait(self, ticket):
        """Issue the steps the call still needs and make the CURRENT stream wait for its results (no host synchronisation).  The call's
        buffers -- the pinned host copy in particular, which the scheduler's stream fills when the call leaves its lane -- stay referenced
        until the call's `done` event has completed (`_reap`): PyTorch's pinned-memory allocator knows nothing about the library's lane
        stream, so a block handed back earlier could be given to the next submit() while the old call's copy into it is still queued."""
        gen, t = ticket
        if gen == self._sched_gen and self._sched:       # (a pool of an earlier generation was flushed and synchronised before it was rebuilt)
            L.check(self.lib.oavsr_generate_wait(self._sched, int(t), self._stream()))
        ent = getattr(self, "_inflight", {}).get(ticket)
        if ent is not None:
            ent[3] = True                                # waited for: released as soon as the device has finished the call
        self._reap()

    def release(self, ticket):
        """The caller has synchronised with the call's results (event / stream synchronisation after `wait`): its buffers may go."""
        getattr(self, "_inflight", {}).pop(ticket, None)

    def _reap(self):
        """Drop the buffers of calls that were waited for AND whose last step, result copy included, has run on the device."""
        infl = getattr(self, "_inflight", None)
        if not infl:
            return
        for tk in [tk for tk, ent in infl.items() if ent[3]]:
            if self.poll(tk):
                del infl[tk]

    def host_result(self, ticket):
        """The pinned host copy [B, max_new + 1] view material of a submitted call (ids then lens), valid once `poll(ticket)` is True."""
        ent = getattr(self, "_inflight", {}).get(ticket)
        return ent[2] if ent is not None and len(ent) > 2 else None

    def result_stream(self) -> "torch.cuda.Stream":
        """Side stream on which results of scheduled calls are awaited and copied to the host (PendingIds)."""
        if getattr(self, "_result_stream", None) is None:
            self._result_stream = torch.cuda.Stream(device=self.device)
        return self._result_stream

    def poll(self, ticket) -> bool:
        """True once the call's results are final; never issues steps, never blocks (oavsr_generate_poll)."""
        gen, t = ticket
        if gen != self._sched_gen or not self._sched:
            return True
        rc = self.lib.oavsr_generate_poll(self._sched, int(t))
        if rc < 0:
            L.check(rc)
        return rc == 1

    def flush(self):
        if self._sched:
            L.check(self.lib.oavsr_scheduler_flush(self._sched))

    def scheduler_trace(self, on: bool):
        """Bracket every call's encoder + prefill phase and every burst of decode steps with timing events (oavsr_scheduler_trace)."""
        if self._sched:
            L.check(self.lib.oavsr_scheduler_trace(self._sched, int(on)))

    def scheduler_trace_read(self, max_recs: int = 65536) -> List[dict]:
        if not self._sched:
            return []
        arr = (L.OavsrTraceRec * max_recs)()
        n = self.lib.oavsr_scheduler_trace_read(self._sched, arr, max_recs)
        if n < 0:
            L.check(n)
        return [dict(kind=int(arr[i].kind), id=int(arr[i].id), steps=int(arr[i].steps), rows=int(arr[i].rows), t0=float(arr[i].t0_ms), t1=float(arr[i].t1_ms))
                for i in range(n)]

    def scheduler_stats(self) -> dict:
        out = (C.c_uint64 * 4)()
        if not self._sched:
            return dict(steps=0, row_steps=0, calls=0, graphs=0)
        L.check(self.lib.oavsr_scheduler_stats(self._sched, out))
        return dict(steps=int(out[0]), row_steps=int(out[1]), calls=int(out[2]), graphs=int(out[3]), rows_per_step=round(out[1] / max(1, out[0]), 2))

    # ---- profiling (bench.py roofline) ------------------------------------------------------------------
    def profile(self, on: bool):
        self.lib.oavsr_profile_enable(int(on))
        self.lib.oavsr_profile_reset()

    def profile_read(self) -> List[dict]:
        arr = (L.OavsrProfEntry * 256)()
        n = self.lib.oavsr_profile_read(arr, 256)
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), ms=float(arr[i].ms), flops=float(arr[i].flops),
                     bytes=float(arr[i].bytes), cu_ms=float(arr[i].cu_ms)) for i in range(max(n, 0))]


class PendingIds:
    """Result of `AVSR_LLMs.forward(..., sync=False)`: the call has been queued on the caller's current HIP stream; `result()` waits for it
    and returns what the synchronous forward returns (ids [B, n] int64 on the device, n = longest hypothesis)."""

    def __init__(self, ids: torch.Tensor, lens: torch.Tensor, stream: "torch.cuda.Stream", engine: "OmniEngine" = None, ticket=None):
        self._ids, self._lens, self._stream = ids, lens, stream
        self._engine, self._ticket = engine, ticket          # ticket: the call's decode rows are in the engine's shared decode loop
        self._event = None
        if ticket is None:
            self._event = torch.cuda.Event()
            self._event.record(stream)

    def _settle(self):
        if self._event is None:                              # first query: have the remaining steps issued, then mark a stream
            # NOT the call's own stream: later calls of the same slot have queued their encoder phases there, and the device-to-host
            # copy of the result would wait behind them (the host would fall back into lock step with the GPU).  The engine's result
            # stream carries nothing but these waits and copies.
            self._stream = self._engine.result_stream()
            with torch.cuda.stream(self._stream):
                self._engine.wait(self._ticket)
            self._ids.record_stream(self._stream)
            self._lens.record_stream(self._stream)
            self._event = torch.cuda.Event()
            self._event.record(self._stream)

    def done(self) -> bool:
        self._settle()
        return self._event.query()

    def ready(self) -> bool:
        """Like done(), but never makes the engine issue decode steps on this call's behalf: True only once the steps that later
        submissions (or waits) issued have completed it.  Collecting only what is ready keeps the scheduler's steps full."""
        if self._event is None and self._ticket is not None:
            return self._engine.poll(self._ticket)           # complete incl. the pinned host copy: result(host=True) touches no stream
        return self.done()

    def result(self, host: bool = False) -> torch.Tensor:
        """ids [B, n] int64, n = longest hypothesis -- on the device like the reference's `generated_ids` (default), or as a host tensor
        (`host=True`: what `tokenizer.batch_decode` needs anyway).  Both go through ONE small copy into pinned memory: a device-to-host copy
        into pageable memory takes ~10 ms on this platform while other streams are busy copying and computing (measured: two `.cpu()` calls
        per batch were 21 of the caller loop's 36 ms)."""
        B, W = self._ids.shape
        if host and self._ticket is not None and self._event is None and self._engine.poll(self._ticket):
            # the scheduler's stream has already put ids and lengths into the call's pinned host buffer: no stream is touched at all
            flat = self._engine.host_result(self._ticket)
            if flat is not None:
                n = int(flat[B * W:].max())
                out = flat[: B * W].view(B, W)[:, :n].to(torch.long)
                self._engine.release(self._ticket)      # poll() said 1: the lane's copy into the pinned buffer is complete
                return out
        self._settle()
        self._event.synchronize()
        if self._ticket is not None:
            self._engine.release(self._ticket)           # the event sits behind the call's `done` on the result stream: the device is through with it
        buf = torch.empty(B, W + 1, dtype=torch.int32, pin_memory=True)
        with torch.cuda.stream(self._stream):                # the tensors belong to that stream's allocator pool
            buf[:, :W].copy_(self._ids, non_blocking=True)
            buf[:, W].copy_(self._lens, non_blocking=True)
        self._stream.synchronize()
        n = int(buf[:, W].max())
        if host:
            return buf[:, :n].to(torch.long)
        with torch.cuda.stream(self._stream):
            out = self._ids[:, :n].to(torch.long)
        self._stream.synchronize()
        out.record_stream(torch.cuda.current_stream(out.device))
        return out


class AVSR_LLMs:
    """Drop-in for `Omni_AVSR.modeling_OmniAVSR.AVSR_LLMs` (inference branch).  Same ctor arguments (:28-34); because
    no hub / tokenizer is reachable offline, `tokenizer` may be None (ids then come from the model family's published
    vocabulary) and the pretrained encoders are not downloaded: weights arrive through `load_state_dict`, exactly like
    `lightning_OmniAVSR.py:148-150`."""

    def __init__(self, modality, pretrain_avhubert_enc_video, use_lora_avhubert, llm_model, hidden_size, intermediate_size, tokenizer,
                 prompt_audio, prompt_video, prompt_audiovisual, pad_id, downsample_ratio_audio, downsample_ratio_video,
                 audio_encoder_name, compression_mode, unfrozen_modules, max_dec_tokens, num_beams, PETF_LLM_name=None,
                 peft_config_llm=None, remove_layernorm_from_projector=False, matry_weights=None, is_task_specific=None,
                 is_matryoshka=False, is_single_matry_projector=False, *, spec: Optional[OmniSpec] = None, dtype: str = "bf16",
                 device: str = "cuda:0"):
        self.modality = modality
        self.max_dec_tokens, self.num_beams = max_dec_tokens, num_beams
        self.llm_model, self.tokenizer = llm_model, tokenizer
        self.is_matryoshka, self.is_task_specific = is_matryoshka, bool(is_task_specific)
        self.downsample_ratio_audio, self.downsample_ratio_video = downsample_ratio_audio, downsample_ratio_video
        if spec is None:
            rank = alpha = 0
            shared = False
            if PETF_LLM_name is not None:
                assert PETF_LLM_name == "lora"                                              # modeling_OmniAVSR.py:202
                rank, alpha = peft_config_llm.RANK, peft_config_llm.ALPHA
                shared = bool(peft_config_llm.SHARED_LORA)
                if shared and not peft_config_llm.IS_TASK_SPECIFIC:
                    raise ValueError("SHARED_LORA without IS_TASK_SPECIFIC is invalid in the reference (Llama_LoRA.py:138-142,254-256)")
            spec = spec_from_names(
                llm_model, audio_encoder_name if modality in ("audio", "audiovisual") else None,
                pretrain_avhubert_enc_video if modality in ("video", "audiovisual") else None,
                use_lora_avhubert=bool(use_lora_avhubert), intermediate_size=intermediate_size, compression_mode=compression_mode,
                downsample_ratio_audio=downsample_ratio_audio, downsample_ratio_video=downsample_ratio_video,
                is_matryoshka=bool(is_matryoshka), is_single_matry_projector=bool(is_single_matry_projector),
                remove_layernorm_from_projector=bool(remove_layernorm_from_projector), lora_rank=rank, lora_alpha=alpha,
                is_task_specific=bool(is_task_specific), shared_lora=shared, max_dec_tokens=max_dec_tokens, num_beams=num_beams)
            assert spec.llm.hidden == hidden_size, "hidden_size does not match the LLM (lightning_OmniAVSR.py:28-37)"
            if tokenizer is not None:
                # every id the inference branch looks up in tokenizer.vocab (:312-322,422-445), and the embedding rows after
                # `resize_token_embeddings(len(tokenizer))` (:214)
                from .harness import special_ids_from_tokenizer
                spec.ids.update(special_ids_from_tokenizer(tokenizer, llm_model))
                import dataclasses
                spec.llm = dataclasses.replace(spec.llm, vocab=len(tokenizer))
        self.spec = spec
        self.engine = OmniEngine(spec, dtype=dtype, device=device)

    # nn.Module-ish surface used by the reference's callers
    def load_state_dict(self, state_dict, strict: bool = True):
        self.engine.load_state_dict(state_dict)
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def requires_grad_(self, *a, **k):
        return self

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def forward(self, inputs, is_trainval=True, modality=None, test_ratio_matry_audio=None, test_ratio_matry_video=None,
                force_len: bool = False, slot: int = 0, sync: bool = True):
        """`slot` / `sync` are not reference arguments: `sync=False` queues the call on the caller's current HIP stream with workspace
        `slot` and returns a PendingIds (harness.evaluate(in_flight=k) keeps k batches in flight that way, as bench.py does)."""
        if is_trainval:
            raise NotImplementedError("training branch (modeling_OmniAVSR.py:265-306) is out of scope of the inference hot path")
        task = self.modality
        if self.is_task_specific and modality is None:
            raise KeyError(None)      # reference: lora_up_Q[None] (Llama_LoRA.py:250)
        if self.is_matryoshka:
            ra, rv = test_ratio_matry_audio, test_ratio_matry_video
        else:
            ra, rv = self.downsample_ratio_audio, self.downsample_ratio_video
        lora_task = modality if self.is_task_specific else task
        if lora_task != task:
            raise NotImplementedError("LoRA task different from the input modality is not supported")
        use_a, use_v = task in ("audio", "audiovisual"), task in ("video", "audiovisual")
        if getattr(self, "_omni_matry_map_quirk", True) and not self.is_matryoshka and not getattr(self.spec, "is_single_matry_projector", False):
            # the reference's inference branch indexes `self.matry_map_*`, which its constructor only creates in the matryoshka branches
            # (modeling_OmniAVSR.py:429,447 vs :66-70,92-94): without --is-matryoshka it needs --is-single-matry-projector
            raise AttributeError("'AVSR_LLMs' object has no attribute 'matry_map_" + ("video" if use_v else "audio") + "'")
        if use_a and ra not in self.spec.rates("audio"):
            raise KeyError(ra)        # matry_map_audio[...] modeling_OmniAVSR.py:447,556
        if use_v and rv not in self.spec.rates("video"):
            raise KeyError(rv)
        # "video_flat" (not a reference key; harness.collate_flat): the valid frames of all utterances back to back, [sum T, 88, 88] f32 --
        # what the engine consumes, without the padded [B, T, 1, 88, 88] detour; needs "video_lengths"
        # "video_u8_flat" (harness.FlatCollator on raw samples): the DECODED uint8 frames [sum T, C, H, W]; the test-time transform then runs
        # inside the engine's stem kernel instead of on the host
        video_is_flat = use_v and ("video_flat" in inputs or "video_u8_flat" in inputs)
        video = (inputs["video_u8_flat"] if "video_u8_flat" in inputs else inputs["video_flat"] if video_is_flat else inputs.get("video")) if use_v else None
        audio = inputs.get("audio") if use_a else None
        Tv = lengths = None
        if use_v:
            Tv = [int(x) for x in inputs["video_lengths"]] if "video_lengths" in inputs else [video.shape[1]] * video.shape[0]
            if not self.is_matryoshka and self.spec.compression_mode == "stack" and rv != 1 and any(t % rv for t in Tv):
                # the non-matryoshka video stack branch has no remainder handling: torch.stack raises on the ragged last group
                # (modeling_OmniAVSR.py:517-520; the audio branch and both matryoshka branches drop the remainder)
                raise RuntimeError("stack expects each tensor to be equal size")
        if use_a:
            B = audio.shape[0]
            if B == 1:
                lengths = [int(max(inputs["lengths"]))]                                   # modeling_OmniAVSR.py:399
            else:
                lengths = [int(x) for x in inputs["lengths"]]
        nrows = len(Tv) if use_v else len(lengths)
        if (not sync and self.num_beams == 1 and getattr(self, "decode_scheduler", True) and nrows <= self.engine.sched_rows
                and getattr(self, "encoder_chunk", None) is None):
            # asynchronous greedy calls share the engine's decode loop (continuous batching inside the engine: the rows of every call in
            # flight ride the same weight passes; ids identical to a call on its own).  `model.decode_scheduler = False` opts out.
            ids, lens, ticket = self.engine.submit(task, video, Tv, audio, lengths, ra, rv, self.max_dec_tokens, force_len,
                                                   video_is_flat=video_is_flat, slot=slot)
            return PendingIds(ids, lens, torch.cuda.current_stream(self.engine.device), self.engine, ticket)
        # `model.encoder_chunk = 8` (not a reference attribute): large evaluation batches run their encoders / prefill in chunks of
        # that many utterances and share one decode loop (oavsr_generate_chunked; identical ids)
        ids, lens = self.engine.generate(task, video, Tv, audio, lengths, ra, rv, self.num_beams, self.max_dec_tokens, force_len,
                                         video_is_flat=video_is_flat, slot=slot, chunk=getattr(self, "encoder_chunk", None))
        if not sync:
            return PendingIds(ids, lens, torch.cuda.current_stream(self.engine.device))
        lens_h = torch.empty(lens.shape, dtype=lens.dtype, pin_memory=True)      # (pageable device-to-host copies are slow on this platform)
        lens_h.copy_(lens, non_blocking=True)
        torch.cuda.current_stream(self.engine.device).synchronize()
        n = int(lens_h.max())
        return ids[:, :n].to(torch.long)
