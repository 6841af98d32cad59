/* omni_avsr.h — C ABI of libomniavsr_hip.so (MI355X / gfx950).
 *
 * This library is the drop-in for the tensor work behind the reference's Python boundary
 *   AVSR_LLMs.forward(inputs, is_trainval=False, modality, test_ratio_matry_audio, test_ratio_matry_video)
 *   (Omni_AVSR/modeling_OmniAVSR.py:263,308-323)  ->  llm.generate(...)  (:313-322).
 * The reference has no FFI layer of its own; the Python shim `omni_avsr_amd.modeling.AVSR_LLMs`
 * binds these entry points with ctypes (see INTEGRATION.md for the stub a maintainer would add).
 *
 * Conventions: every pointer named *_dev is device memory owned by the caller; *_host is host memory read
 * before the call returns.  All calls are asynchronous on `stream` (a hipStream_t passed as void*), allocate
 * nothing, never throw; return 0 on success, <0 on error (message: oavsr_last_error()).  One oavsr_model is
 * immutable and shareable; one oavsr_workspace per stream.  One process per GPU.
 */
#ifndef OMNI_AVSR_H_
#define OMNI_AVSR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): oavsr_op_stem_pool takes the pixel format (three ints after `video`), oavsr_gemm_desc gained `grid`, the decode scheduler entry
 * points (oavsr_scheduler_*, oavsr_generate_submit / _wait / _poll) and oavsr_trace_rec were added.
 * 3 (round 5): oavsr_gemm_desc follows the single decode regime (weight-streaming decode kernels removed: no `skinny` / norm_gain / rn_* /
 * shared_chip; `kernel`, decode_tiled, sync, row_scale, ng_* added); oavsr_generate / _submit take optional per-utterance digest buffers.
 * 4 (round 6): oavsr_gemm_desc gained rowsq_strips / row_sumsq_form (the <= 32-row strip kernels' layout of the deferred RMSNorm's sums of
 * squares); oavsr_get_launch_objective added.
 * A binding built against an older version must not load. */
#define OAVSR_ABI_VERSION 4
#define OAVSR_DT_F32 0  /* parity mode: exact-f32 MFMA */
#define OAVSR_DT_BF16 1 /* throughput mode: bf16 storage, f32 accumulate (reference: bf16-true, eval_OmniAVSR.py:19) */
#define OAVSR_MAX_RATES 8
#define OAVSR_TASK_AUDIO 0
#define OAVSR_TASK_VIDEO 1
#define OAVSR_TASK_AUDIOVISUAL 2

typedef struct oavsr_model oavsr_model;
typedef struct oavsr_workspace oavsr_workspace;
typedef struct oavsr_scheduler oavsr_scheduler;
typedef void* oavsr_stream; /* hipStream_t */

/* Mirrors the ctor arguments of AVSR_LLMs (modeling_OmniAVSR.py:28-34) + the HF configs it pulls in. */
typedef struct oavsr_config {
  int32_t dtype;
  /* LLM (Llama_LoRA.py / Qwen_LoRA.py) */
  int32_t llm_family; /* 0 llama, 1 qwen2 (no BOS, qkv bias) */
  int32_t llm_hidden, llm_layers, llm_heads, llm_kv_heads, llm_head_dim, llm_mlp, llm_vocab;
  float llm_rms_eps;
  int32_t llm_qkv_bias;
  int32_t lora_inner;         /* round(hidden / RANK), 0 = no LoRA (Llama_LoRA.py:125) */
  int32_t lora_k2;            /* padded K of the fused LoRA segment (multiple of 64) */
  int32_t lora_task_specific; /* IS_TASK_SPECIFIC */
  /* Whisper encoder */
  int32_t wh_dim, wh_layers, wh_heads, wh_ffn;
  /* AV-HuBERT */
  int32_t av_dim, av_layers, av_heads, av_ffn, av_conv_pos, av_conv_groups;
  int32_t av_lora_k2; /* 0 = no LoRA on the video encoder (modeling_OmniAVSR.py:125-142) */
  /* compressor + projector (modeling_OmniAVSR.py:86-111,171-196) */
  int32_t proj_inter;
  int32_t compression_stack; /* 0 avg-pooling, 1 stack */
  int32_t n_rates_audio, rates_audio[OAVSR_MAX_RATES];
  int32_t n_rates_video, rates_video[OAVSR_MAX_RATES];
  int32_t proj_per_rate; /* 1: one projector per rate (ModuleList), 0: single projector */
  int32_t proj_layernorm_audio, proj_layernorm_video; /* LayerNorm really present (index 3 of the Sequential) */
  /* token ids (lightning_OmniAVSR.py:83-88, modeling_OmniAVSR.py:312-322) */
  int32_t id_bos, id_eos, id_pad, id_audio_sos, id_audio_eos, id_video_sos, id_video_eos;
  int32_t prompt_len[3]; /* tokens in prompt_audio / prompt_video / prompt_audiovisual buffers */
} oavsr_config;

typedef struct oavsr_tensor {
  const char* name; /* packed-weight name, see omni_avsr_amd/weights.py */
  uint64_t offset;  /* byte offset inside the device blob (256-byte aligned) */
  uint64_t nbytes;
} oavsr_tensor;

int oavsr_abi_version(void);
const char* oavsr_last_error(void);
/* Every OAVSR_* variable of the process environment, ';'-separated "NAME=value" ("NAME=value?": a name the library does not read) into buf;
 * returns how many there are.  The library's own switches (csrc/model.hip: kKnobs) only pick between kernel forms that produce the same
 * bits or turn diagnostics on -- no environment variable changes a result; bench.py records this string in `config.env`. */
int oavsr_env_report(char* buf, int32_t nbytes);
/* What the GEMM dispatcher optimises, process-wide (default 0).  0: the calls in flight share the chip (oavsr_generate_submit pipelines,
 * several streams) -- launches of 64 .. 255 coarse tiles stay on as many CUs as they have tiles, because the job is bound by CU-time and the
 * other CUs are not idle.  1: ONE call at a time on an otherwise idle chip (the reference's evaluation protocol, data_module.py:238-242) --
 * such launches take finer tiles that cover more CUs and finish sooner.  Never a bit: every tile shape accumulates in the same order. */
int oavsr_set_launch_objective(int32_t alone);
/* the current objective (0 / 1): it is GLOBAL to the process -- a caller that changes it for a phase reads it first and restores it after */
int oavsr_get_launch_objective(void);

/* weights: one device blob + manifest (replaces AVSR_LLMs.load_state_dict, lightning_OmniAVSR.py:148-150) */
int oavsr_model_create(const oavsr_config* cfg, const void* blob_dev, uint64_t blob_bytes, const oavsr_tensor* tensors,
                       int32_t n_tensors, oavsr_model** out);
void oavsr_model_destroy(oavsr_model* m);
/* Scheduling hint for callers that keep several oavsr_generate() calls in flight (one workspace + stream each): when on,
 * the encoder + prefill phase of a call starts only after the prefill of the previously issued call has finished, so the
 * MFMA-bound phases run back to back and only the decode loops overlap them.  Results are unaffected. */
int oavsr_model_serialize_encoders(oavsr_model* m, int32_t on);

/* scratch + KV caches for batches of up to max_B utterances, max_Tv video frames / max_Ts samples each */
int oavsr_workspace_create(const oavsr_model* m, int32_t max_B, int32_t max_Tv, int32_t max_Ts, int32_t max_beams,
                           int32_t max_new, oavsr_workspace** out);
void oavsr_workspace_destroy(oavsr_workspace* ws);
uint64_t oavsr_workspace_bytes(const oavsr_workspace* ws);
/* What `video_dev` holds in the calls that follow on this workspace.  channels = 0 (default): f32 [sum(Tv)][88][88], the frames after the
 * dataset's test-time transform (datamodule/transforms.py:93-99), as the reference's forward() receives them.  channels = 1 / 3: the
 * DECODED frames, uint8 [sum(Tv)][channels][height][width] (av_dataset.py: load_video(...).permute(0, 3, 1, 2)); x / 255, CenterCrop(88),
 * Grayscale and Normalize(0.421, 0.165) then run inside the fused stem kernel's window gather with the host code's fp32 operations in
 * the host code's order (same bits), so a 16 s clip is uploaded as 3.7 MB (gray) / 11 MB (RGB 96 x 96) instead of 12.4 MB of floats and
 * the host does no per-pixel work. */
int oavsr_workspace_set_video_format(oavsr_workspace* ws, int32_t channels, int32_t height, int32_t width);
/* Input-sensitivity digests (tests / benchmarks: does a call under a pipelined schedule see ITS inputs and ITS cache rows?).  digests_dev:
 * device array of max_B x 2 uint64, or NULL (default: off).  Every later oavsr_prefill / oavsr_generate / oavsr_generate_chunked /
 * oavsr_generate_submit on this workspace writes, on the call's stream, for utterance b of the call: [2b] = digest of its inputs_embeds
 * rows (the spliced prompt, modeling_OmniAVSR.py:397-458, as stored in the compute dtype), [2b + 1] = digest of its first-step logits row
 * (fp32, the prefill's last position).  Digest of 32-bit words w_0..w_{n-1}: sum_i splitmix64_finalizer(w_i | (i + 1) << 32) mod 2^64. */
int oavsr_workspace_set_digests(oavsr_workspace* ws, uint64_t* digests_dev);

/* ---- stage entry points (each replaces the reference function cited) ------------------------------------ */

/* encode_video (modeling_OmniAVSR.py:461-483) = AVHubertModel.extract_finetune (hubert.py:695-755) + compressor.
 * video_dev: f32 [sum(Tv)][88][88] (frames of all utterances back to back; or uint8 frames, oavsr_workspace_set_video_format); out_dev: [sum(Tv/rate)][D or D*rate];
 * enc_dev (optional): uncompressed encoder output [sum(Tv)][D]. */
int oavsr_encode_video(const oavsr_model* m, oavsr_workspace* ws, const void* video_dev, const int32_t* Tv_host, int32_t B,
                       int32_t rate, void* out_dev, void* enc_dev, oavsr_stream stream);

/* encode_audio (modeling_OmniAVSR.py:528-558) = log-mel + WhisperEncoder + truncation (:537) + compressor.
 * wav_dev: f32 [B][ld]; Ts_host: valid samples; keep_host: tokens kept per utterance (:537, computed by the shim);
 * out_dev: [sum(keep/rate)][D or D*rate]; enc_dev (optional): [B][1500][D]; mel_dev (optional): [B][3000][80]. */
int oavsr_encode_audio(const oavsr_model* m, oavsr_workspace* ws, const float* wav_dev, int64_t ld, const int32_t* Ts_host,
                       const int32_t* keep_host, int32_t B, int32_t rate, void* out_dev, void* enc_dev, void* mel_dev,
                       oavsr_stream stream);

/* audio_proj / video_proj (modeling_OmniAVSR.py:429,447): which = 0 audio, 1 video */
int oavsr_project(const oavsr_model* m, oavsr_workspace* ws, int32_t which, int32_t rate, const void* feats_dev, int32_t rows,
                  void* out_dev, oavsr_stream stream);

/* prepare_inputs, inference branch (modeling_OmniAVSR.py:406-458): splice BOS / <audio> A </audio> / <video> V </video> /
 * prompt.  na_host / nv_host: projected tokens per utterance (NULL when the modality is absent).
 * embeds_dev: [sum(S)][hidden]; S_host_out: prompt length per utterance. */
int oavsr_build_prompt(const oavsr_model* m, oavsr_workspace* ws, int32_t task, int32_t B, const int32_t* na_host,
                       const void* audio_tok_dev, const int32_t* nv_host, const void* video_tok_dev, void* embeds_dev,
                       int32_t* S_host_out, oavsr_stream stream);

/* LLM prompt pass (LlamaForCausalLM_lora.forward, Llama_LoRA.py:328-398; once per utterance, last-position logits
 * only).  Fills the prefix KV cache of the workspace.  logits_dev (optional): f32 [B][vocab]. */
int oavsr_prefill(const oavsr_model* m, oavsr_workspace* ws, int32_t task, int32_t B, const void* embeds_dev,
                  const int32_t* S_host, float* logits_dev, oavsr_stream stream);

/* generate (modeling_OmniAVSR.py:313-322) after oavsr_prefill: beams == 1 greedy, > 1 HF beam search.
 * out_ids_dev: int32 [B][max_new] (pad-filled); out_len_dev: int32 [B].  force_len: suppress EOS (benchmarks). */
int oavsr_decode(const oavsr_model* m, oavsr_workspace* ws, int32_t task, int32_t B, int32_t beams, int32_t max_new,
                 int32_t force_len, int32_t* out_ids_dev, int32_t* out_len_dev, oavsr_stream stream);

/* AVSR_LLMs.forward(inputs, is_trainval=False) end to end (modeling_OmniAVSR.py:308-323). */
int oavsr_generate(const oavsr_model* m, oavsr_workspace* ws, int32_t task, const void* video_dev, const int32_t* Tv_host,
                   const float* wav_dev, int64_t wav_ld, const int32_t* Ts_host, const int32_t* keep_host, int32_t B,
                   int32_t rate_audio, int32_t rate_video, int32_t beams, int32_t max_new, int32_t force_len,
                   int32_t* out_ids_dev, int32_t* out_len_dev, oavsr_stream stream);
/* Same result as oavsr_generate on the B utterances, scheduled for throughput: encoders / projector / prompt / prefill run in
 * chunks of chunk_B utterances (the launch shapes of a batch-of-chunk_B request), every chunk's prefix KV and last-token logits
 * land in their own slots, and ONE decode loop then serves all B rows, so the LLM weights are streamed once per step for the
 * whole group (continuous batching of the decode phase).  chunk_B == B is oavsr_generate. */
int oavsr_generate_chunked(const oavsr_model* m, oavsr_workspace* ws, int32_t task, const void* video_dev, const int32_t* Tv_host,
                           const float* wav_dev, int64_t wav_ld, const int32_t* Ts_host, const int32_t* keep_host, int32_t B, int32_t chunk_B,
                           int32_t rate_audio, int32_t rate_video, int32_t beams, int32_t max_new, int32_t force_len, int32_t* out_ids_dev,
                           int32_t* out_len_dev, oavsr_stream stream);

/* ---- decode scheduler: continuous batching of the greedy decode phase across independent calls ------------------------------
 * The reference evaluates one utterance per forward() (eval_OmniAVSR.py:18-24, datamodule/data_module.py:238-242): nothing to mirror,
 * this is the engine's own serving layer under the same per-utterance results.  Calls keep arriving one batch at a time on the caller's
 * streams (encoders + prefill of a call run there, as in oavsr_generate); the decode rows of every submitted call live in row slots of
 * ONE pool per model, and one step loop on the scheduler's own stream serves all admitted rows (up to max_rows <= 128 per step, each row
 * at its own step), so the LLM weights are streamed once per step for all of them.  Rows join after their call's prefill and leave after
 * their own max_new tokens.  The schedule is fixed by the host at submission time in units of steps -- no host thread, no host
 * synchronisation, no device-side queue -- and the ids of an utterance are bit-identical to oavsr_generate on that utterance alone.
 * Greedy decode only (beam search keeps its own per-call loop: oavsr_generate). */
/* n_slots: row slots of the pool (>= max_rows; max_rows x lanes fills every lane, extra blocks let the prefill of later calls run ahead of the step loops);
 * lanes: independent step loops (own stream each) sharing the pool -- a step is a chain of ~100 dependent launches, so one loop serves a
 * bounded number of utterances per second however wide its steps are; calls go to the lane with the fewest rows;
 * max_prefix: longest prompt (rows) a slot can hold; max_new: longest hypothesis. */
int oavsr_scheduler_create(const oavsr_model* m, int32_t n_slots, int32_t max_rows, int32_t lanes, int32_t max_prefix, int32_t max_new,
                           oavsr_scheduler** out);
void oavsr_scheduler_destroy(oavsr_scheduler* s);
uint64_t oavsr_scheduler_bytes(const oavsr_scheduler* s);
/* Caller stream i (0 .. 7) of the scheduler: a HIP stream the ENGINE created, before its lanes' streams, for submitting calls on.  ROCm
 * binds a stream to one of its few hardware queues at creation, and which caller stream shares a queue with which lane decides how encoder
 * phases and decode bursts take turns: with streams of the caller's own making the same loop ran at 226-229 or at 255-260 utt/s depending
 * on when they were created (round 3).  Streams handed out here are in the fast order by construction; they belong to the PROCESS and to
 * the DEVICE the scheduler lives on (eight per device: created with the first scheduler on that device, shared by every scheduler created
 * there later, never destroyed -- a host framework that has recorded them, e.g. PyTorch's caching allocator, never holds a dangling stream).
 * Any other stream still works (speed only, never results).  NULL: index out of range. */
oavsr_stream oavsr_scheduler_caller_stream(oavsr_scheduler* s, int32_t i);
/* oavsr_generate(beams = 1) split in two: submit queues encoders + prefill on `stream` (workspace ws, as oavsr_generate) and the
 * admission of the B rows on the scheduler's stream; it returns a ticket.  out_ids_dev / out_len_dev ([B][max_new] / [B] int32) are
 * written by the scheduler's stream when the call's last step has run and must stay valid until oavsr_generate_wait.
 * out_ids_host / out_len_host (optional, PINNED host memory of the same shapes): filled on the scheduler's stream right after the device
 * buffers, before the call counts as done -- once oavsr_generate_poll returns 1 the host can read them without touching any stream (a caller
 * loop that fetches every result through a stream of its own pays for it: on ROCm the enqueue can block the host behind another stream that
 * shares the hardware queue). */
int oavsr_generate_submit(const oavsr_model* m, oavsr_scheduler* s, oavsr_workspace* ws, int32_t task, const void* video_dev,
                          const int32_t* Tv_host, const float* wav_dev, int64_t wav_ld, const int32_t* Ts_host, const int32_t* keep_host,
                          int32_t B, int32_t rate_audio, int32_t rate_video, int32_t max_new, int32_t force_len, int32_t* out_ids_dev,
                          int32_t* out_len_dev, int32_t* out_ids_host, int32_t* out_len_host, oavsr_stream stream, int64_t* ticket_out);
/* Issues whatever steps the call still needs (other admitted rows ride along) and makes `stream` wait for its results: work queued on
 * `stream` afterwards sees out_ids_dev / out_len_dev.  No host synchronisation. */
int oavsr_generate_wait(oavsr_scheduler* s, int64_t ticket, oavsr_stream stream);
/* 1: the call's results are final (its last step has been issued by later submissions / waits AND has run); 0: not yet.  Never issues
 * steps and never blocks: a caller that only collects what is ready lets the scheduler keep its steps full.  < 0: error. */
int oavsr_generate_poll(oavsr_scheduler* s, int64_t ticket);
/* Issues the remaining steps of every admitted call. */
int oavsr_scheduler_flush(oavsr_scheduler* s);
/* Pacing: steps the loop advances per admission before it waits for the new call's prefill.  0 (default) = (max_new - 1) / k with
 * k = max_rows / B calls sharing a step in the steady state; 1 = lowest latency (rows join as early as possible, fewer rows per weight
 * pass); larger = fewer, fuller steps.  Scheduling only: results never depend on it. */
int oavsr_scheduler_set_stride(oavsr_scheduler* s, int32_t steps);
/* Schedule trace: while on, every submitted call's encoder + prefill phase (on the caller's stream) and every burst of steps a lane issues
 * are bracketed by timing events; _read waits for the device and returns the records in issue order, times in ms relative to the first
 * record's start (at most 65 536 records per session; later ones are dropped).  kind 0: a call's phase (id = ticket, rows = B); kind 1: a burst (id = lane, steps, rows = live rows).  Diagnostics
 * (bench.py's `schedule` object): a kernel trace cannot show this overlap because rocprofv3 takes the hardware queues in turn. */
typedef struct oavsr_trace_rec {
  int32_t kind, id, steps, rows;
  float t0_ms, t1_ms;
} oavsr_trace_rec;
int oavsr_scheduler_trace(oavsr_scheduler* s, int32_t on);
int oavsr_scheduler_trace_read(oavsr_scheduler* s, oavsr_trace_rec* out, int32_t max_recs);
/* out4 = {steps issued, sum over steps of live rows, calls submitted, step graphs instantiated} */
int oavsr_scheduler_stats(const oavsr_scheduler* s, uint64_t* out4);

/* ---- profiling hook (bench.py roofline): per-kernel-family HIP-event timing of every launch ---------------- */
typedef struct oavsr_prof_entry {
  char name[32];
  uint64_t launches;
  double ms;    /* summed HIP-event duration */
  double flops; /* algorithmic flops (2*M*N*K) */
  double bytes; /* algorithmic bytes */
  double cu_ms; /* sum of duration x the fraction of the chip's CUs the launch can occupy (GEMMs: by the dispatcher's tile; others: 1): its CU-time */
} oavsr_prof_entry;
int oavsr_profile_enable(int32_t on); /* on=1: bracket launches with events (serialises the stream); 0: off */
int oavsr_profile_reset(void);
int oavsr_profile_read(oavsr_prof_entry* out, int32_t max_entries); /* returns number of entries, syncs events */

/* ---- single-kernel entry points (parity tests of each HIP kernel through the C ABI) ------------------------- */
typedef struct oavsr_gemm_desc {
  int32_t dtype, M, N, K, K2;
  const void *A, *W, *A2, *W2;
  int64_t lda, ldw, lda2, ldw2;
  void* C;
  int64_t ldc;
  int32_t out_f32;
  const float *col_scale, *col_bias, *prelu;
  int32_t act; /* 0 none, 1 gelu, 2 relu, 3 prelu, 4 silu */
  const void* residual;
  int64_t ldr;
  int32_t res_after_act;
  int32_t groups, a_goff;
  int32_t amode; /* 0 dense, 1 conv2d NHWC, 2 conv1d token-major, 3 video stem */
  int32_t H, W_, Cin, OH, OW, stride, pad, ksize;
  const int32_t* rowinfo; /* int4 per output row */
  int32_t taps, Kvalid;
  const int32_t* frameinfo; /* int2 per frame */
  int32_t kernel;           /* kernel selector: 0 dispatcher's choice; parity tests only: 2 ping-pong 256 x 256, 3 persistent ping-pong,
                               5 16-wave 256 x 256, 7 256 x 128, 8 persistent ping-pong with 192-row tiles */
  int32_t swiglu_pairs;     /* fused SwiGLU epilogue (gate / up weight rows interleaved by 16) */
  int32_t decode_tiled;     /* 1: decode-shaped launch (M <= 128): the 128-row LDS-DMA tiles at any row count */
  /* decode-shaped launches (M <= 128): split-K with fp32 partial slabs [ksplit][M][N], summed in slab order -- by the tile's last-arriving
     workgroup INSIDE the launch (`sync`: OAVSR_GEMM_SYNC_INTS zeroed ints, the arrival counters; zero again when the launch ends) or, with
     sync == NULL, by a reduce launch of its own that runs the same code (same bits).  <= 16 rows (the strips) need neither; given BOTH, a launch
     of few long strips (<= half a workgroup per CU, >= 128 KiB of weights per strip: down_proj of a 2048-wide model) runs TWO workgroups per
     strip, the second continuing the first's chain of additions from M x N fp32 sums in `partial` (same bits; counters zero again at the end) */
  int32_t ksplit;
  float* partial;
  int32_t* sync;
  /* the decode path's deferred RMSNorm (Llama_LoRA.py:587-655 computes norm(x) W^T): A holds T(x * gain); the accumulator is multiplied,
     before col_scale / col_bias, by 1 / rms(x[m]) = rsqrt(sum_{t < row_sumsq_tiles} row_sumsq[t * M + m] / row_dim + row_eps).  NULL: none */
  const float* row_sumsq;
  int32_t row_sumsq_tiles, row_dim;
  float row_eps;
  /* split-K only (optional): "next norm" outputs for the consumer of the residual stream C = x (plain epilogue, C in `dtype`, N a multiple
     of 64): ng_out[m][n] = T(x[m][n] * ng_gain[n]) (row stride ng_ld), ng_rowsq[t * M + m] = sum over column tile t (64 columns) of
     x[m][n]^2 -- the consumer's row_sumsq with ceil(N / 64) tiles */
  void* ng_out;
  int64_t ng_ld;
  const float* ng_gain;
  float* ng_rowsq;
  int32_t grid;        /* persistent kernel only: workgroups (multiple of 8); 0 = the launcher's rule */
  /* <= 16 rows (csrc/decode_strip.hip): rowsq_strips = 1: ng_rowsq is written per 16-column STRIP, [N / 16][M] (needs M <= 16); row_sumsq_form = 1:
     row_sumsq is read in that layout -- row_sumsq_tiles tiles = 4 x as many entries, a tile's sum formed as (s0 + s1) + (s2 + s3), the tiles added
     in order: the bits of the tiled layout */
  int32_t rowsq_strips, row_sumsq_form;
} oavsr_gemm_desc;
#define OAVSR_GEMM_SYNC_INTS 4096
int oavsr_op_gemm(const oavsr_gemm_desc* d, oavsr_stream stream);
/* n (2 .. 6) decode-shaped launches of <= 16 rows (same M, same dtype; each one oavsr_op_gemm would give to the strips of
   csrc/decode_strip.hip), launch i + 1 consuming what launch i produced, as the PHASES OF ONE LAUNCH: the workgroups of a phase request their
   first weights, wait until every workgroup of the phase before has released its outputs, then read them.  The bits of n oavsr_op_gemm calls
   (what a decode step issues between two attention launches: o_proj, gate / up, down_proj, the next layer's LoRA-down and QKV; the reference
   runs them as separate modules, Llama_LoRA.py:246-259,314,587-655).  counters: n + 1 ints, zero before the first launch; the launch leaves
   them zero (counters[n] != 0 afterwards: a wait was given up -- never on a correct schedule). */
int oavsr_op_gemm_chain(const oavsr_gemm_desc* d, int32_t n, int32_t* counters, oavsr_stream stream);

typedef struct oavsr_attn_desc {
  int32_t dtype, head_dim, n_heads, n_kv_heads, n_seqs, max_q, causal;
  const void *Q, *K, *V;
  void* O;
  int64_t ldq, ldk, ldv, ldo;
  const int32_t *cu_q, *cu_k;
  float scale;
  int32_t form;             /* bf16, head_dim 64: 0 = the dispatcher's kernels (16x16x32 MFMAs), 32 = the 32x32x16-MFMA kernel (another fp32
                               summation order; the engine uses it for the Whisper encoder, whose sequences always have 1500 frames) */
} oavsr_attn_desc;
int oavsr_op_attention(const oavsr_attn_desc* d, oavsr_stream stream);

typedef struct oavsr_decode_attn_desc {
  int32_t dtype, head_dim, n_heads, n_kv_heads, rows, beams, max_prefix, max_tail;
  const void *Q, *Kp, *Vp, *Kt, *Vt;
  void* O;
  int64_t ldq, ldo;
  const int32_t *prefix_len, *step, *anc;
  float scale;
  /* fused RoPE + cache append when inv_freq != NULL: Q = fused [q|k|v] rows, rotated with pos[row], k/v appended at cpos[row] */
  const float* inv_freq;
  const int32_t *pos, *cpos;
} oavsr_decode_attn_desc;
int oavsr_op_decode_attention(const oavsr_decode_attn_desc* d, oavsr_stream stream);

/* mode: 0 LayerNorm, 1 RMSNorm, 2 LayerNorm over [zeros(D), x] */
int oavsr_op_norm(int32_t mode, int32_t dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, const float* b,
                  int32_t rows, int32_t D, float eps, oavsr_stream stream);
/* Deferred RMSNorm of the decode path (Llama_LoRA.py:587-655 input_layernorm / post_attention_layernorm / model.norm): y = T(x * w),
 * sumsq[row] = sum_c x[row][c]^2; the consumer GEMM takes y as A and sumsq as oavsr_gemm_desc.row_sumsq (one tile, row_dim = D). */
int oavsr_op_rms_prep(int32_t dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, float* sumsq, int32_t rows, int32_t D,
                      oavsr_stream stream);
/* Fused visual stem: Conv3d(1->64, k (5,7,7), s (1,2,2), p (2,3,3)) + folded BN + PReLU + MaxPool3d((1,3,3), s (1,2,2), p (0,1,1))
 * (reference resnet.py:137-141).  video: f32 [F][88][88] (channels = 0) or uint8 [F][channels][height][width] with the test-time
 * transform fused (oavsr_workspace_set_video_format); frameinfo int32 pairs {t, len} per frame; w [64][256] in the model
 * dtype with k = dt*49 + ky*7 + kx (245 valid); wp_scratch: 64*288 elements of the model dtype; out [F][22][22][64]. */
int oavsr_op_stem_pool(int32_t dtype, const void* video, int32_t channels, int32_t height, int32_t width, const int32_t* frameinfo_dev,
                       int32_t F, const void* w, void* wp_scratch, const float* scale, const float* shift, const float* slope, void* out,
                       oavsr_stream stream);
/* Image-resident 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, NHWC bf16, H*W <= 512 (ResNet trunk layer1,
 * resnet.py:61-74): out = prelu(scale * conv(x, w) + shift + residual); w [64][3*3*64] tap-major; slope NULL = no activation;
 * residual may alias out. */
int oavsr_op_conv3x3_c64(const void* x, int32_t imgs, int32_t H, int32_t W, const void* w, const float* scale, const float* shift,
                         const float* slope, const void* residual, void* out, oavsr_stream stream);
/* the 128-channel twin (trunk layer2, 11 x 11 maps): groups of images resident in LDS, weights streamed tap by tap; bit-identical to the implicit GEMM */
int oavsr_op_conv3x3_c128(const void* x, int32_t imgs, int32_t H, int32_t W, const void* w, const float* scale, const float* shift,
                         const float* slope, const void* residual, void* out, oavsr_stream stream);
/* Row-wise candidates of one HF beam-search step (`_beam_search`: log_softmax + beam score, top 2K per row, ties by lower
 * token id): cand_val / cand_tok [rows][2K], sorted.  need_exact: rows int32 of scratch for the certified fast path
 * (NULL = exact radix-select kernel only). */
int oavsr_op_beam_topk(const float* logits, int64_t ld, int32_t rows, int32_t V, int32_t K, const float* beam_scores, float* cand_val,
                       int32_t* cand_tok, int32_t* need_exact, oavsr_stream stream);
int oavsr_op_maxpool3x3s2(int32_t dtype, const void* x, void* y, int32_t imgs, int32_t H, int32_t W, int32_t C, oavsr_stream stream);
int oavsr_op_avgpool_hw(int32_t dtype, const void* x, void* y, int32_t imgs, int32_t HW, int32_t C, oavsr_stream stream);
int oavsr_op_time_compress(int32_t dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const int32_t* seg_dev, int32_t n_seqs,
                           int32_t max_out, int32_t rate, int32_t C, int32_t stack, oavsr_stream stream);
int oavsr_op_copy_rows(int32_t dtype, const void* src, int64_t lds, const int32_t* map_dev, void* y, int64_t ldy, int32_t n,
                       int32_t D, oavsr_stream stream);
int oavsr_op_rope_kv(int32_t dtype, int32_t rows, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, void* qkv, int64_t ld,
                     const float* inv_freq, const int32_t* pos, const int32_t* slot, const int32_t* cpos, void* kcache,
                     void* vcache, int32_t max_ctx, oavsr_stream stream);
/* interleaved: 0 = [gate | up], 1 = 16-column blocks g0 u0 g1 u1 ... (the packed gate/up weight layout) */
int oavsr_op_swiglu(int32_t dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t I, int32_t interleaved,
                    oavsr_stream stream);
/* scratch: NULL (one workgroup per row) or rows*64*8 bytes (two-stage, fills the chip for 128k-entry vocabularies) */
int oavsr_op_argmax(const float* logits, int64_t ld, int32_t rows, int32_t V, int32_t banned, int32_t* out, void* scratch,
                    oavsr_stream stream);
int oavsr_op_build_rowinfo(const int32_t* cu_out, const int32_t* cu_in, int32_t n_seqs, int32_t stride, int32_t* rowinfo,
                           int32_t rows, oavsr_stream stream);
/* twiddle / mel filterbank come from the model (built at oavsr_model_create) */
int oavsr_op_logmel(const oavsr_model* m, int32_t dtype, const float* wav, int64_t ld, const int32_t* lens_dev, int32_t B,
                    float* mel_ws, float* max_ws, void* out, int64_t ldo, oavsr_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNI_AVSR_H_ */
