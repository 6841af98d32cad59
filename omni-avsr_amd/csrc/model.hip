// Host orchestration of the Omni-AVSR inference hot path + the C ABI (include/omni_avsr.h).
//
// Stage map (reference file:line each stage replaces is quoted in the header):
//   encode_video  : stem conv3d (implicit GEMM) -> maxpool -> ResNet-18 trunk (implicit-GEMM convs, BN/PReLU/residual
//                   fused) -> avgpool -> proj -> LN([0,x]) -> post_extract_proj -> grouped pos-conv (+GELU+res) ->
//                   24 pre-LN layers (fused QKV + LoRA K-segment, flash attention, fused FFN epilogues) -> LN -> compressor
//   encode_audio  : log-mel -> conv1/conv2 (implicit GEMM, GELU, +pos-emb) -> N pre-LN layers -> LN -> truncate -> compressor
//   project       : Linear+ReLU, Linear(+LN)
//   build_prompt  : row gathers into inputs_embeds
//   prefill       : Llama/Qwen2 layers on all prompt rows of all utterances (var-len causal attention), KV prefix cache,
//                   last-position logits only
//   decode        : greedy / beam loop, one step of decode-shaped GEMMs (tiled LDS-DMA kernels, split-K reduced inside the launch) per token,
//                   no host sync inside the loop
#include <stdarg.h>
#include <stdlib.h>
#include <math.h>
#include <string>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <atomic>
#include <mutex>

#include "../../include/omni_avsr.h"
#include "kernels.h"

namespace oavsr {

static thread_local char g_err[1024] = "";
void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------------------------------------
// developer switches: the one table, the one getenv site (common.h: knob)
// ------------------------------------------------------------------------------------------------
struct KnobDef { const char* name; int dflt; const char* what; };
static const KnobDef kKnobs[] = {
    {"OAVSR_PROF_SHAPES", 0, "profiling hook: one family per GEMM shape instead of per kernel"},
    {"OAVSR_DECODE_GRAPH", 1, "0: decode steps are launched eagerly instead of replayed from hipGraphs (same kernels, same bits)"},
    {"OAVSR_DECODE_CHAIN", 0, "1: the strip GEMMs between two attention launches of a decode step (o_proj, gate / up, down_proj, next LoRA-down, QKV) as the phases of ONE launch (decode_strip.hip: decode_chain_kernel; same bits).  Measured slower than the launches it replaces (1.56 vs 1.15 ms per step at 1 row): off"},
    {"OAVSR_DECODE_CHAIN_PER_CU", 1, "workgroups of a chained strip launch per CU: 1 = the full ring depth, 2 = half the depth each (speed only)"},
    {"OAVSR_STRIP_PAIRS", 1, "0: a strip launch of few long strips (down_proj of a 2048-wide model at <= 16 rows) keeps one workgroup per strip instead of two that share its k-slices (same bits)"},
    {"OAVSR_DECODE_STRIPS", 1, "0: decode-shaped GEMMs of <= 16 rows run on the 128-row LDS-DMA tiles like larger launches instead of the register-streaming strips (same bits)"},
    {"OAVSR_CONV_C128", 1, "0: the 128-channel 3x3 convolutions of the trunk as implicit GEMMs instead of the image-resident kernel (same bits)"},
    {"OAVSR_CONV_TILE256", 1, "0: implicit-GEMM convolutions with >= 256 output channels on the 256 x 128 tile instead of the 16-wave 256 x 256 tile (same bits)"},
    {"OAVSR_ENC_SIDE_BY_SIDE", 1, "0: an audio-visual call made alone (the latency objective) runs its two encoders one after the other instead of side by side on two streams (same launches, same bits)"},
    {"OAVSR_FUSED_REDUCE_ROWS", 32, "decode-shaped GEMMs with at most this many rows reduce their split-K slabs inside the launch, others in a second launch (same code, same bits)"},
    {"OAVSR_GEMM_TILE_ROWS", 0, "persistent GEMM: 256 / 192 = that tile height wherever its form exists, 0 = the launcher's rule (256 while calls share the chip; alone: whichever finishes sooner); same bits"},
};
int knob(const char* name) {
  for (const KnobDef& k : kKnobs)
    if (strcmp(k.name, name) == 0) {
      const char* e = getenv(name);
      return e ? atoi(e) : k.dflt;
    }
  fprintf(stderr, "libomniavsr_hip: unregistered developer switch %s\n", name);
  abort();
}

// ------------------------------------------------------------------------------------------------
// profiler: optional HIP-event bracket around every launch, aggregated per kernel family
// ------------------------------------------------------------------------------------------------
struct Profiler {
  bool on = false;
  struct Rec { int fam; hipEvent_t e0, e1; double flops, bytes, cu_share; };
  std::vector<Rec> recs;
  std::vector<Rec> empty;                      // back-to-back event pairs: what a bracket measures with nothing inside
  std::vector<std::string> names;
  std::vector<hipEvent_t> pool;
  std::mutex mu;                               // streams of several host threads may profile at once
  int family(const char* n) {
    for (size_t i = 0; i < names.size(); ++i) if (names[i] == n) return (int)i;
    names.push_back(n);
    return (int)names.size() - 1;
  }
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
  }
  void reset() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& r : recs) { pool.push_back(r.e0); pool.push_back(r.e1); }
    for (auto& r : empty) { pool.push_back(r.e0); pool.push_back(r.e1); }
    recs.clear(); empty.clear();
  }
};
static Profiler g_prof;

struct ProfScope {
  hipStream_t st; int idx = -1;
  ProfScope(const char* name, double flops, double bytes, hipStream_t s, double cu_share = 1.0) : st(s) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.empty.size() < 16) {            // calibration: an empty bracket on the same stream, subtracted from every record on read
      Profiler::Rec c; c.fam = -1; c.e0 = g_prof.get(); c.e1 = g_prof.get(); c.flops = c.bytes = 0; c.cu_share = 1.0;
      hipEventRecord(c.e0, st); hipEventRecord(c.e1, st);
      g_prof.empty.push_back(c);
    }
    Profiler::Rec r; r.fam = g_prof.family(name); r.e0 = g_prof.get(); r.e1 = g_prof.get(); r.flops = flops; r.bytes = bytes; r.cu_share = cu_share;
    hipEventRecord(r.e0, st);
    g_prof.recs.push_back(r);
    idx = (int)g_prof.recs.size() - 1;
  }
  ~ProfScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    hipEventRecord(g_prof.recs[idx].e1, st);
  }
};

// Profiling families are named after the kernel instantiation that will actually run, so that the HIP-event averages
// bench.py reports line up 1:1 with the rows of a `rocprofv3 --kernel-trace --stats` summary.
static const char* gemm_family(const GemmArgs& a) {
  static thread_local char buf[64];
  const char* dt = a.dtype == DT_F32 ? "f32" : "bf16";
  const char* mode = a.amode == A_DENSE ? "gemm" : a.amode == A_CONV2D ? "conv2d_gemm" : a.amode == A_CONV1D ? "conv1d_gemm" : "stem_gemm";
  const char* tile = gemm_strip_selected(a) ? (a.M <= 16 ? "strip16" : "strip32") : gemm_dma_eligible(a) ? gemm_dma_variant(a) : (a.N <= 64 ? "128x64" : "128x128");
  static const bool shapes = knob("OAVSR_PROF_SHAPES") != 0;      // developer aid: one family per GEMM shape
  if (shapes) { snprintf(buf, sizeof(buf), "%s%c%s %dx%dx%d", a.decode_tiled ? "decode_" : "", mode[0], tile, a.M, a.N, a.K + a.K2); buf[31] = 0; return buf; }
  // decode-shaped GEMMs: a family of their own ("decode_..."), so that bench.py keeps the decode side apart from the encoder / prefill
  // GEMMs that run on the same kernels
  snprintf(buf, sizeof(buf), "%s%s_%s_%s", a.decode_tiled ? "decode_" : "", mode, tile, dt);
  return buf;
}

// Split-K factors are STATIC per call site (functions of the layer's N and K, never of the row count), so that an utterance gets the same
// bits whether it runs alone or inside a batch.  The only split-K call sites are the decode-shaped GEMMs (decode_gemm below); the narrow
// encoder projections and the prefill's LoRA down projections run un-split on the small-GEMM tile (gemm2.hip).
static std::atomic<int> g_alone{0};            // oavsr_set_launch_objective
int launch_alone() { return g_alone.load(std::memory_order_relaxed); }
static int run_gemm(const GemmArgs& a, hipStream_t st) {
  const double esz = a.dtype == DT_F32 ? 4 : 2;
  const double kk = (double)a.K + a.K2;
  const double flops = 2.0 * a.M * a.N * kk * a.groups;
  const double bytes = esz * ((double)a.M * kk + (double)a.N * kk * a.groups) + (a.out_f32 ? 4 : esz) * (double)a.M * a.N * a.groups;
  ProfScope ps(gemm_family(a), flops, bytes, st, g_prof.on ? gemm_cu_share(a) : 1.0);
  return launch_gemm(a, st);
}

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
struct EncLayer {
  const float *ln1w, *ln1b, *qkvb, *ob, *ln2w, *ln2b, *fc1b, *fc2b;
  const void *qkvw, *lora_dn, *lora_up, *ow, *fc1w, *fc2w;
  const void* qkvw_cat = nullptr;      // [3D][D + k2] = [qkv.w | lora_up.w] (model_create: enc_cat), see run_encoder_layers
};
struct ConvBN { const void* w; const float *scale, *shift, *slope; };
struct ResBlock { ConvBN c1, c2, down; bool has_down; int cin, cout, stride; };
struct LlmLayer {
  const float *ln1w, *ln2w, *qkvb;
  const void *qkvw, *ow, *guw, *downw;
  const void* lora_dn[3]; const void* lora_up[3];
};
struct Projector { const void *fc1w, *fc2w; const float *fc1b, *fc2b, *lnw, *lnb; };

}  // namespace oavsr

using namespace oavsr;

// caller streams handed out by oavsr_scheduler_caller_stream: per DEVICE, created with the first scheduler on that device, before that
// scheduler's lanes (hardware-queue binding order); process lifetime (a host framework that has seen them never holds a dangling stream)
static constexpr int kCallerStreams = 8;
static std::unordered_map<int, std::vector<hipStream_t>> g_caller_streams;      // device id -> its eight streams
static std::mutex g_caller_mu;

struct oavsr_model {
  oavsr_config cfg;
  const char* blob = nullptr;
  uint64_t blob_bytes = 0;
  std::unordered_map<std::string, std::pair<uint64_t, uint64_t>> t;
  int dtype = DT_BF16; size_t esz = 2;
  // video
  bool has_video = false, has_audio = false;
  ConvBN stem; ResBlock blocks[8];
  const void *vproj_w, *vpost_w, *vpos_w; const float *vproj_b, *vln_w, *vln_b, *vpost_b, *vpos_b, *vlnf_w, *vlnf_b;
  std::vector<EncLayer> vlayers;
  void* enc_cat = nullptr;               // the video layers' concatenated QKV + LoRA-up weights (one allocation)
  // audio
  const void *aconv1_w, *aconv2_w, *apos; const float *aconv1_b, *aconv2_b, *alnf_w, *alnf_b;
  std::vector<EncLayer> alayers;
  std::vector<Projector> proj_a, proj_v;
  // llm
  const void *embed, *lm_head; const float *norm_w, *inv_freq;
  std::vector<LlmLayer> llayers;
  const void* prompt[3];
  // constants built at create
  double* twiddle = nullptr; float* melfb = nullptr;
  void* stem_wp = nullptr;                     // stem weights in the (dt, ky, kx8) order of stem.hip
  // cross-stream scheduling of concurrent generate() calls (oavsr_model_serialize_encoders)
  mutable bool serialize_enc = false;
  mutable hipEvent_t enc_done = nullptr;       // recorded after the prefill of the most recent generate()
  mutable hipStream_t enc_last = nullptr;
  mutable bool enc_recorded = false;

  const void* ptr(const std::string& n, bool required = true) const {
    auto it = t.find(n);
    if (it == t.end()) { if (required) set_last_error("missing packed tensor '%s'", n.c_str()); return nullptr; }
    return blob + it->second.first;
  }
};

static int decode_form() { return knob("OAVSR_DECODE_CHAIN") | knob("OAVSR_DECODE_STRIPS") << 1 | knob("OAVSR_DECODE_CHAIN_PER_CU") << 2; }
struct DecodeGraphKey {
  int B, task, max_new, banned;
  int form = decode_form();      // the developer switches a captured step depends on: a graph captured under other settings is not replayed
  bool operator==(const DecodeGraphKey& o) const { return B == o.B && task == o.task && max_new == o.max_new && banned == o.banned && form == o.form; }
};

// Activations of one encoder's transformer stack + the small index tables its launches read.
struct EncScratch {
  void *x = nullptr, *h = nullptr, *qkv = nullptr, *attn = nullptr, *ffn = nullptr, *lora = nullptr, *hcat = nullptr;
  int4* rowinfo = nullptr; int* cu = nullptr; int4* seg = nullptr;
};

struct oavsr_workspace {
  const oavsr_model* m = nullptr;
  std::vector<std::pair<DecodeGraphKey, hipGraphExec_t>> graphs;   // captured greedy decode steps
  int max_B = 0, max_Tv = 0, max_Ts = 0, max_beams = 1, max_new = 0;
  char* arena = nullptr; size_t arena_bytes = 0, used = 0;
  char* host_base = nullptr; char* host = nullptr; size_t host_bytes = 0, host_used = 0;   // pinned staging (2 halves)
  hipEvent_t host_ev[2] = {nullptr, nullptr}; unsigned call_idx = 0;
  // encoder scratch: one set per encoder, so that the two encoders of an audio-visual call may run side by side (encode_prefill)
  EncScratch esa, esv;           // Whisper's / AV-HuBERT's (esv.hcat [rows][D + k2]: LayerNorm output next to the LoRA-down output, run_encoder_layers)
  void *enc_out_v, *enc_out_a;
  // the video encoder's stream when the two run side by side: created by the FIRST such call, not with the workspace -- a stream that exists takes a share of
  // the process's 8 hardware queues even when idle (one per workspace of a 4-stream pipeline: beam-15 throughput 157 -> 108 utt/s)
  hipStream_t enc2 = nullptr; hipEvent_t enc2_fork = nullptr, enc2_join = nullptr; bool two_sets = false;
  // video front
  void *pool_out, *trunk[2], *feat512, *featproj, *ln2d;
  int2* frameinfo; int4* rowinfo2; int* cu_b;
  // audio front
  void *mel, *conv1_out, *pos_rep; float *mel_ws, *max_ws; int* lens_dev;
  // projector / prompt
  void *comp_a, *comp_v, *proj_h, *tok_a, *tok_v, *embeds; int* maps;
  // llm
  void *lx, *lh, *lqkv, *lattn, *lgu, *lact, *llora, *lastx;
  float* logits;
  void *kpre, *vpre, *ktail, *vtail; int max_prefix = 0; size_t kpre_layer = 0, ktail_layer = 0;
  int *pos, *slot, *cpos, *cu_s, *prefix_len, *last_rows;
  int *amax; void* amax_scratch; GreedyState gs;
  // decode-shaped GEMMs: arrival counters of the split-K reduce (GemmArgs::sync), the deferred RMSNorm's 1 / rms per row and its per-tile sums
  int* gsync = nullptr; float* rowsq = nullptr;
  // strip launches of a decode step that feed each other, collected between two attention launches and issued as the phases of one launch
  // (decode_gemm / chain_flush); chain_ctr: the phases' done counters of one step (zero between launches) + the error word at kChainErr
  struct { bool open = false; int n = 0, used = 0; GemmArgs g[kStripChainMax]; } chain;
  int* chain_ctr = nullptr;
  // beam search state
  float *beam_scores, *cand_val, *partial; int *cand_tok, *topk_flag, *hyp_n, *hyp_len, *hyp_tok, *seq[2], *anc[2]; double* hyp_score; size_t partial_bytes = 0;
  int max_tok_enc = 0, max_S_total = 0;
  int S_host[1024]; int B_cur = 0;
  // where the prefill in flight puts its prefix KV rows / last-token logits / prefix lengths: this workspace's own caches (default) or
  // the row slots of a decode scheduler's pool (set by prefill_impl for the duration of the call)
  void *pf_k = nullptr, *pf_v = nullptr; size_t pf_layer = 0; int pf_max_prefix = 0;
  const int* rowmap = nullptr;                 // decode scheduler only: activation row -> row slot (DecodeAttnArgs::rowmap)
  StemSrc vsrc;                                // what `video_dev` of the calls on this workspace holds (oavsr_workspace_set_video_format)
  hipStream_t own_stream = nullptr; hipEvent_t own_in = nullptr, own_out = nullptr;      // calls made on the legacy default stream run here (generate_impl)
  uint64_t* digests = nullptr;                 // optional [utterance][2]: {inputs_embeds, first-step logits} digests of the calls on this workspace (oavsr_workspace_set_digests)

  void* take(size_t bytes) {
    size_t off = align_up(used, 256);
    if (off + bytes > arena_bytes) return nullptr;
    used = off + bytes;
    return arena + off;
  }
  template <typename T> T* stage(size_t n) {   // pinned host staging (bump; reset per call)
    size_t off = align_up(host_used, 64);
    if (off + n * sizeof(T) > host_bytes) return nullptr;
    host_used = off + n * sizeof(T);
    return reinterpret_cast<T*>(host + off);
  }
};

// ------------------------------------------------------------------------------------------------
// Decode scheduler: continuous batching of the greedy decode phase across independent generate calls (include/omni_avsr.h).
//
// The reference decodes one utterance at a time (eval_OmniAVSR.py:18-24, datamodule/data_module.py:238-242); a decode step is bound by
// streaming the LLM weights once, whatever the number of rows (<= 128) it serves.  Calls therefore keep arriving on their own streams
// (encoders + prefill per call, unchanged), but their decode rows live in ROW SLOTS of one pool per model -- prefix KV written there by
// the call's prefill, tail KV, position / step / hypothesis per slot -- and ONE step loop on the scheduler's own stream serves every
// admitted row: rows join after their prefill (event) and leave after their own max_new tokens, at step granularity.
//
// The schedule is decided by the host at submission time, in units of steps (no device-side queues, no host thread, no host sync):
//   submit(call i):  [stream of the call]  wait(slots free) -> encoders -> prefill into the slots -> record E_i
//                    [scheduler stream]    n steps for the rows already admitted -> wait(E_i) -> admit(i)
// with n = (max_new - 1) / k, k = row capacity / rows of the call (plus whatever steps make room for the call's rows): the loop advances by 1/k of a hypothesis per admission, so in the
// steady state k calls share every weight pass and the n steps run beside the next call's MFMA-bound phase.  Steps not yet issued when
// the caller asks for a result are issued then (oavsr_generate_wait / oavsr_scheduler_flush).
// Results do not depend on the schedule: the decode-shaped GEMMs give a row the same bits wherever it sits and whatever shares the
// step, at every row count (tests/test_invariance_gpu.py, tests/test_scheduler_gpu.py).
// ------------------------------------------------------------------------------------------------
struct SchedCall {
  int64_t ticket = -1;
  int task = 0, B = 0, slot0 = 0, limit = 0, remaining = 0, banned = -1, lane = 0;
  int32_t *out_ids = nullptr, *out_len = nullptr;
  int32_t *host_ids = nullptr, *host_len = nullptr;     // optional pinned host copies, filled on the lane's stream right after the device ones
  hipEvent_t prefilled = nullptr, done = nullptr;
  bool retired = true;
};

// One step loop: its own stream, activation scratch, row map and step graphs.  Several lanes (each up to max_rows rows per step) share the
// pool: a decode step is a chain of ~100 dependent launches, and beside the MFMA-bound phases of other calls a chain advances at ~2.7 ms
// per step whatever its width, i.e. one chain serves ~380 utterances/s (31/32 of a step per utterance).  A second chain doubles that
// without costing weight passes per utterance -- every lane still fills its steps -- at the price of twice the calls in flight.
struct SchedLane {
  hipStream_t st = nullptr;
  oavsr_workspace* dws = nullptr;              // activation scratch of a step (lx ... logits) + the pool's pointers (shared by all lanes)
  int* rowmap = nullptr;                       // device [128]
  void* partials = nullptr;                    // [rows][64] float2 arg-max partials
  std::vector<int> active;                     // ring indices in admission order
  bool map_dirty = true;                       // membership changed since the last step: rowmap + input rows must be rebuilt
  int cur_R = 0;
  uint64_t steps = 0;                          // steps issued on this lane
  std::vector<std::pair<DecodeGraphKey, hipGraphExec_t>> graphs;
};

struct oavsr_scheduler {
  const oavsr_model* m = nullptr;
  std::vector<SchedLane> lanes;
  oavsr_workspace* pool = nullptr;             // kpre / vpre / ktail / vtail, prefix_len: indexed by row slot
  char* arena = nullptr; size_t arena_bytes = 0;
  int n_slots = 0, max_rows = 0, max_prefix = 0, max_new = 0;
  int device = 0;                              // the device the pool and the lanes' streams live on
  SchedState ss;
  float* logits_pre = nullptr;                 // [n_slots][vocab] last-prompt-position logits, written by the calls' prefills
  std::vector<int64_t> slot_owner;             // ticket that last owned the slot (-1: never used)
  std::vector<char> slot_used;
  std::vector<uint64_t> slot_freed_at;         // allocation prefers the block freed longest ago (its owner's last step is furthest back in the queue)
  uint64_t free_seq = 0;
  static constexpr int kRing = 1024;
  std::vector<SchedCall> ring;
  int64_t next_ticket = 0;
  uint64_t steps_issued = 0, row_steps = 0, calls = 0;     // statistics (oavsr_scheduler_stats)
  int stride = 0;                              // steps issued per admission; 0 = (max_new - 1) / (max_rows / B) (oavsr_scheduler_set_stride)
  // schedule trace (oavsr_scheduler_trace): timing events around every call's encoder + prefill phase (on the caller's stream) and around every
  // burst of steps a lane issues -- the only way to SEE the overlap (rocprofv3's kernel trace takes the queues in turn)
  struct TraceRec { int kind, id, steps, rows; hipEvent_t e0, e1; };
  bool trace = false;
  std::vector<TraceRec> trace_recs;
  std::vector<hipEvent_t> trace_pool;
  hipEvent_t trace_event() {
    if (!trace_pool.empty()) { hipEvent_t e = trace_pool.back(); trace_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; hipEventCreate(&e); return e;
  }
  static constexpr size_t kTraceMax = 1 << 16;           // records kept per trace session (two events each); later ones are dropped
  bool tracing() const { return trace && trace_recs.size() < kTraceMax; }
  mutable std::mutex mu;
};

namespace oavsr {

// Every API call stages its host metadata in one half of the pinned buffer; the half is reused two calls later, after
// the event recorded at the end of the call that last used it (keeps back-to-back calls asynchronous).
struct CallScope {
  oavsr_workspace* ws; hipStream_t st; int half;
  CallScope(oavsr_workspace* w, hipStream_t s) : ws(w), st(s) {
    half = (int)(ws->call_idx++ & 1u);
    hipEventSynchronize(ws->host_ev[half]);
    ws->host = ws->host_base + (size_t)half * ws->host_bytes;
    ws->host_used = 0;
  }
  ~CallScope() { hipEventRecord(ws->host_ev[half], st); }
};

static int h2d(void* dst, const void* src_pinned, size_t bytes, hipStream_t st) {
  OAVSR_CHECK_HIP(hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, st));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// slaney mel filterbank (librosa / transformers mel_filter_bank(norm="slaney", mel_scale="slaney")), [201][80]
// ------------------------------------------------------------------------------------------------
static void build_melfb(std::vector<float>& fb) {
  const int nf = 201, nm = 80;
  auto hz2mel = [](double f) { return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0; };
  auto mel2hz = [](double m) { return m >= 15.0 ? 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0; };
  std::vector<double> ff(nm + 2);
  double m0 = hz2mel(0.0), m1 = hz2mel(8000.0);
  for (int i = 0; i < nm + 2; ++i) ff[i] = mel2hz(m0 + (m1 - m0) * i / (nm + 1));
  fb.assign((size_t)nf * nm, 0.f);
  for (int f = 0; f < nf; ++f) {
    double hz = 8000.0 * f / (nf - 1);
    for (int m = 0; m < nm; ++m) {
      double down = -(ff[m] - hz) / (ff[m + 1] - ff[m]);
      double up = (ff[m + 2] - hz) / (ff[m + 2] - ff[m + 1]);
      double v = std::max(0.0, std::min(down, up));
      v *= 2.0 / (ff[m + 2] - ff[m]);
      fb[(size_t)f * nm + m] = (float)v;
    }
  }
}

static int resolve_enc_layer(const oavsr_model* m, const std::string& p, bool lora, EncLayer& L) {
#define GETP(field, name, req) L.field = (decltype(L.field))m->ptr(p + name, req); if (req && !L.field) return -1;
  GETP(ln1w, "ln1.w", true) GETP(ln1b, "ln1.b", true) GETP(qkvw, "qkv.w", true) GETP(qkvb, "qkv.b", true)
  GETP(ow, "o.w", true) GETP(ob, "o.b", true) GETP(ln2w, "ln2.w", true) GETP(ln2b, "ln2.b", true)
  GETP(fc1w, "fc1.w", true) GETP(fc1b, "fc1.b", true) GETP(fc2w, "fc2.w", true) GETP(fc2b, "fc2.b", true)
  GETP(lora_dn, "lora_dn.w", lora) GETP(lora_up, "lora_up.w", lora)
#undef GETP
  return 0;
}

static int resolve_convbn(const oavsr_model* m, const std::string& p, bool slope, ConvBN& c) {
  c.w = m->ptr(p + ".w"); c.scale = (const float*)m->ptr(p + ".scale"); c.shift = (const float*)m->ptr(p + ".shift");
  c.slope = slope ? (const float*)m->ptr(p + ".slope") : nullptr;
  return (c.w && c.scale && c.shift && (!slope || c.slope)) ? 0 : -1;
}

// Every packed tensor must have exactly the byte size the config implies: the kernels take their extents (vocabulary rows of
// lm_head / embed, projector widths, LoRA ranks ...) from the config, so a checkpoint of another shape would otherwise be read
// past its end or with the wrong stride.  The reference's strict load_state_dict raises in that case (lightning_OmniAVSR.py:150).
static int validate_sizes(const oavsr_model* m) {
  const oavsr_config& c = m->cfg;
  const uint64_t e = m->esz;
  int bad = 0;
  auto chk = [&](const std::string& n, uint64_t expect) {
    if (bad) return;
    auto it = m->t.find(n);
    if (it == m->t.end()) return;                      // presence is checked by the resolver
    if (it->second.second != expect) {
      set_last_error("model_create: size mismatch for packed tensor '%s': the checkpoint holds %llu bytes, the config implies %llu",
                     n.c_str(), (unsigned long long)it->second.second, (unsigned long long)expect);
      bad = 1;
    }
  };
  auto enc = [&](const std::string& p, uint64_t D, uint64_t F, uint64_t k2) {
    chk(p + "ln1.w", D * 4); chk(p + "ln1.b", D * 4); chk(p + "ln2.w", D * 4); chk(p + "ln2.b", D * 4);
    chk(p + "qkv.w", 3 * D * D * e); chk(p + "qkv.b", 3 * D * 4); chk(p + "o.w", D * D * e); chk(p + "o.b", D * 4);
    chk(p + "fc1.w", F * D * e); chk(p + "fc1.b", F * 4); chk(p + "fc2.w", D * F * e); chk(p + "fc2.b", D * 4);
    if (k2) { chk(p + "lora_dn.w", k2 * D * e); chk(p + "lora_up.w", 3 * D * k2 * e); }
  };
  if (m->has_video) {
    const uint64_t D = c.av_dim;
    chk("v.stem.w", 64 * 256 * e); chk("v.stem.scale", 64 * 4); chk("v.stem.shift", 64 * 4); chk("v.stem.slope", 64 * 4);
    for (int i = 0; i < 8; ++i) {
      const ResBlock& rb = m->blocks[i];
      const std::string p = "v.l" + std::to_string(i / 2 + 1) + "." + std::to_string(i % 2);
      chk(p + ".conv1.w", (uint64_t)rb.cout * 9 * rb.cin * e); chk(p + ".conv2.w", (uint64_t)rb.cout * 9 * rb.cout * e);
      for (const char* cn : {".conv1", ".conv2"})
        for (const char* f : {".scale", ".shift", ".slope"}) chk(p + cn + f, (uint64_t)rb.cout * 4);
      if (rb.has_down) { chk(p + ".down.w", (uint64_t)rb.cout * rb.cin * e); chk(p + ".down.scale", (uint64_t)rb.cout * 4); chk(p + ".down.shift", (uint64_t)rb.cout * 4); }
    }
    chk("v.proj.w", D * 512 * e); chk("v.proj.b", D * 4); chk("v.ln.w", 2 * D * 4); chk("v.ln.b", 2 * D * 4);
    chk("v.post.w", D * 2 * D * e); chk("v.post.b", D * 4);
    chk("v.pos.w", D * (uint64_t)c.av_conv_pos * (D / std::max(1, c.av_conv_groups)) * e); chk("v.pos.b", D * 4);
    chk("v.lnf.w", D * 4); chk("v.lnf.b", D * 4);
    for (int i = 0; i < c.av_layers; ++i) enc("v.L" + std::to_string(i) + ".", D, c.av_ffn, c.av_lora_k2);
  }
  if (m->has_audio) {
    const uint64_t D = c.wh_dim;
    chk("a.conv1.w", D * 256 * e); chk("a.conv1.b", D * 4); chk("a.conv2.w", D * 3 * D * e); chk("a.conv2.b", D * 4);
    chk("a.pos", 1500 * D * e); chk("a.lnf.w", D * 4); chk("a.lnf.b", D * 4);
    for (int i = 0; i < c.wh_layers; ++i) enc("a.L" + std::to_string(i) + ".", D, c.wh_ffn, 0);
  }
  const uint64_t d = c.llm_hidden, I = c.llm_mlp, V = c.llm_vocab;
  for (int which = 0; which < 2; ++which) {
    const auto& vec = which == 0 ? m->proj_a : m->proj_v;
    const uint64_t Denc = which == 0 ? c.wh_dim : c.av_dim;
    const int* rates = which == 0 ? c.rates_audio : c.rates_video;
    for (size_t i = 0; i < vec.size(); ++i) {
      const std::string p = std::string(which == 0 ? "pa." : "pv.") + std::to_string(i) + ".";
      const uint64_t Din = c.compression_stack ? Denc * rates[i] : Denc;      // stack mode: one projector per rate (or a single rate)
      chk(p + "fc1.w", (uint64_t)c.proj_inter * Din * e); chk(p + "fc1.b", (uint64_t)c.proj_inter * 4);
      chk(p + "fc2.w", d * c.proj_inter * e); chk(p + "fc2.b", d * 4); chk(p + "ln.w", d * 4); chk(p + "ln.b", d * 4);
    }
  }
  const uint64_t nq = (uint64_t)c.llm_heads * c.llm_head_dim, nqkv = (uint64_t)(c.llm_heads + 2 * c.llm_kv_heads) * c.llm_head_dim;
  chk("l.embed", V * d * e); chk("l.lm_head", V * d * e); chk("l.norm.w", d * 4); chk("l.inv_freq", (uint64_t)c.llm_head_dim / 2 * 4);
  for (int i = 0; i < c.llm_layers; ++i) {
    const std::string p = "l.L" + std::to_string(i) + ".";
    chk(p + "ln1.w", d * 4); chk(p + "ln2.w", d * 4); chk(p + "qkv.w", nqkv * d * e); chk(p + "qkv.b", nqkv * 4);
    chk(p + "o.w", d * nq * e); chk(p + "gu.w", 2 * I * d * e); chk(p + "down.w", d * I * e);
    for (const char* tn : {"0", "1", "2", "x"}) { chk(p + "lora_dn." + tn + ".w", (uint64_t)c.lora_k2 * d * e); chk(p + "lora_up." + tn + ".w", nqkv * c.lora_k2 * e); }
  }
  for (int t = 0; t < 3; ++t) if (c.prompt_len[t] > 0) chk("l.prompt." + std::to_string(t), (uint64_t)c.prompt_len[t] * d * e);
  return bad ? -1 : 0;
}

static int resolve_model(oavsr_model* m) {
  const oavsr_config& c = m->cfg;
  m->has_video = m->t.count("v.stem.w") > 0;
  m->has_audio = m->t.count("a.conv1.w") > 0;
  if (m->has_video) {
    OAVSR_TRY(resolve_convbn(m, "v.stem", true, m->stem));
    const int chans[5] = {64, 64, 128, 256, 512};
    for (int l = 1; l <= 4; ++l)
      for (int b = 0; b < 2; ++b) {
        ResBlock& rb = m->blocks[(l - 1) * 2 + b];
        std::string p = "v.l" + std::to_string(l) + "." + std::to_string(b);
        rb.cin = b == 0 ? chans[l - 1] : chans[l]; rb.cout = chans[l];
        rb.stride = (b == 0 && l > 1) ? 2 : 1;
        rb.has_down = (b == 0 && l > 1);
        OAVSR_TRY(resolve_convbn(m, p + ".conv1", true, rb.c1));
        OAVSR_TRY(resolve_convbn(m, p + ".conv2", true, rb.c2));
        if (rb.has_down) OAVSR_TRY(resolve_convbn(m, p + ".down", false, rb.down));
      }
#define GETM(field, name) m->field = (decltype(m->field))m->ptr(name); if (!m->field) return -1;
    GETM(vproj_w, "v.proj.w") GETM(vproj_b, "v.proj.b") GETM(vln_w, "v.ln.w") GETM(vln_b, "v.ln.b")
    GETM(vpost_w, "v.post.w") GETM(vpost_b, "v.post.b") GETM(vpos_w, "v.pos.w") GETM(vpos_b, "v.pos.b")
    GETM(vlnf_w, "v.lnf.w") GETM(vlnf_b, "v.lnf.b")
    m->vlayers.resize(c.av_layers);
    for (int i = 0; i < c.av_layers; ++i) OAVSR_TRY(resolve_enc_layer(m, "v.L" + std::to_string(i) + ".", c.av_lora_k2 > 0, m->vlayers[i]));
  }
  if (m->has_audio) {
    GETM(aconv1_w, "a.conv1.w") GETM(aconv1_b, "a.conv1.b") GETM(aconv2_w, "a.conv2.w") GETM(aconv2_b, "a.conv2.b")
    GETM(apos, "a.pos") GETM(alnf_w, "a.lnf.w") GETM(alnf_b, "a.lnf.b")
    m->alayers.resize(c.wh_layers);
    for (int i = 0; i < c.wh_layers; ++i) OAVSR_TRY(resolve_enc_layer(m, "a.L" + std::to_string(i) + ".", false, m->alayers[i]));
  }
  for (int which = 0; which < 2; ++which) {
    bool have = which == 0 ? m->has_audio : m->has_video;
    if (!have) continue;
    int n = c.proj_per_rate ? (which == 0 ? c.n_rates_audio : c.n_rates_video) : 1;
    auto& vec = which == 0 ? m->proj_a : m->proj_v;
    bool ln = which == 0 ? c.proj_layernorm_audio : c.proj_layernorm_video;
    vec.resize(n);
    for (int i = 0; i < n; ++i) {
      std::string p = std::string(which == 0 ? "pa." : "pv.") + std::to_string(i) + ".";
      Projector& P = vec[i];
      P.fc1w = m->ptr(p + "fc1.w"); P.fc1b = (const float*)m->ptr(p + "fc1.b");
      P.fc2w = m->ptr(p + "fc2.w"); P.fc2b = (const float*)m->ptr(p + "fc2.b");
      if (!P.fc1w || !P.fc1b || !P.fc2w || !P.fc2b) return -1;
      P.lnw = ln ? (const float*)m->ptr(p + "ln.w") : nullptr; P.lnb = ln ? (const float*)m->ptr(p + "ln.b") : nullptr;
      if (ln && (!P.lnw || !P.lnb)) return -1;
    }
  }
  GETM(embed, "l.embed") GETM(lm_head, "l.lm_head") GETM(norm_w, "l.norm.w") GETM(inv_freq, "l.inv_freq")
  m->llayers.resize(c.llm_layers);
  for (int i = 0; i < c.llm_layers; ++i) {
    LlmLayer& L = m->llayers[i];
    std::string p = "l.L" + std::to_string(i) + ".";
    L.ln1w = (const float*)m->ptr(p + "ln1.w"); L.ln2w = (const float*)m->ptr(p + "ln2.w");
    L.qkvw = m->ptr(p + "qkv.w"); L.ow = m->ptr(p + "o.w"); L.guw = m->ptr(p + "gu.w"); L.downw = m->ptr(p + "down.w");
    if (!L.ln1w || !L.ln2w || !L.qkvw || !L.ow || !L.guw || !L.downw) return -1;
    L.qkvb = c.llm_qkv_bias ? (const float*)m->ptr(p + "qkv.b") : nullptr;
    if (c.llm_qkv_bias && !L.qkvb) return -1;
    for (int t = 0; t < 3; ++t) {
      L.lora_dn[t] = L.lora_up[t] = nullptr;
      if (c.lora_k2 > 0) {
        std::string tn = c.lora_task_specific ? std::to_string(t) : "x";
        L.lora_dn[t] = m->ptr(p + "lora_dn." + tn + ".w"); L.lora_up[t] = m->ptr(p + "lora_up." + tn + ".w");
        if (!L.lora_dn[t] || !L.lora_up[t]) return -1;
      }
    }
  }
  for (int t = 0; t < 3; ++t) m->prompt[t] = c.prompt_len[t] > 0 ? m->ptr("l.prompt." + std::to_string(t), false) : nullptr;
#undef GETM
  return validate_sizes(m);
}

// ------------------------------------------------------------------------------------------------
// shared transformer-encoder layer stack (AV-HuBERT wav2vec2.py:977-1006 / WhisperEncoderLayer), pre-LN
// ------------------------------------------------------------------------------------------------
// The LoRA down projections and AV-HuBERT's out-projection / fc2 are never split (a static property of the call site: an utterance's bits
// must not depend on its batch) -- they run on the 128 x 64 small-GEMM tile (gemm2.hip: gemm_small_tile; the TILE may follow M, it never
// changes a bit).

// The LoRA term of the encoders' QKV projection, (h A^T)(s B)^T, as columns D .. D + k2 of ONE K range: LayerNorm writes h with a row
// stride of D + k2, the LoRA-down GEMM writes t = h A^T next to it, and QKV = [h | t] . [W | s B]^T is a plain single-segment GEMM -- which
// the persistent 256 x 256 kernel takes (it has no second K segment; AV-HuBERT's QKV ran on the 16-wave kernel: 40 -> 30 us).  Whole k-tiles
// of 128 bytes on both sides, so the k-tile sequence -- hence every bit -- is that of the two-segment launch.
static bool enc_cat_ok(int D, int k2, size_t esz) { return k2 > 0 && (D * esz) % 128 == 0 && (k2 * esz) % 128 == 0; }

static int run_encoder_layers(const oavsr_model* m, const EncScratch& es, const std::vector<EncLayer>& layers, int rows, int D,
                              int heads, int ffn, int lora_k2, const int* cu_dev, int n_seqs, int max_len, hipStream_t st, int attn_form = 0) {
  const int dt = m->dtype;
  const int hd = D / heads;
  for (const EncLayer& L : layers) {
    const bool cat = lora_k2 > 0 && L.qkvw_cat && es.hcat;
    const int ldh = cat ? D + lora_k2 : D;
    void* h = cat ? es.hcat : es.h;
    { ProfScope ps("layernorm", 0, 2.0 * rows * D * m->esz, st);
      OAVSR_TRY(launch_layernorm(dt, es.x, D, h, ldh, L.ln1w, L.ln1b, rows, D, 1e-5f, st)); }
    GemmArgs g;
    if (lora_k2 > 0) {   // t = h . [A_q; A_v]^T
      g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = lora_k2; g.K = D; g.A = h; g.lda = ldh; g.W = L.lora_dn; g.ldw = D;
      g.C = cat ? (void*)((char*)h + (size_t)D * m->esz) : es.lora; g.ldc = cat ? ldh : lora_k2;
      OAVSR_TRY(run_gemm(g, st));
    }
    g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = 3 * D; g.A = h; g.lda = ldh;
    if (cat) { g.K = D + lora_k2; g.W = L.qkvw_cat; g.ldw = D + lora_k2; }
    else {
      g.K = D; g.W = L.qkvw; g.ldw = D;
      if (lora_k2 > 0) { g.A2 = es.lora; g.lda2 = lora_k2; g.W2 = L.lora_up; g.ldw2 = lora_k2; g.K2 = lora_k2; }
    }
    g.col_bias = L.qkvb; g.C = es.qkv; g.ldc = 3 * D;
    OAVSR_TRY(run_gemm(g, st));
    AttnArgs a; a.dtype = dt; a.head_dim = hd; a.n_heads = heads; a.n_kv_heads = heads; a.n_seqs = n_seqs;
    a.Q = es.qkv; a.K = (const char*)es.qkv + (size_t)D * m->esz; a.V = (const char*)es.qkv + (size_t)2 * D * m->esz;
    a.ldq = a.ldk = a.ldv = 3 * D; a.O = es.attn; a.ldo = D; a.cu_q = cu_dev; a.cu_k = cu_dev; a.max_q = max_len;
    a.scale = 1.0f / sqrtf((float)hd); a.causal = 0; a.form = attn_form;
    { double fl = 0; /* 4*T*T*D per sequence, upper bound with max_len */ fl = 4.0 * (double)rows * max_len * D;
      ProfScope ps(dt == DT_F32 ? "attention_f32" : "attention_bf16", fl, 4.0 * rows * D * m->esz, st);
      OAVSR_TRY(launch_attention(a, st)); }
    g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = D; g.K = D; g.A = es.attn; g.lda = D; g.W = L.ow; g.ldw = D; g.col_bias = L.ob;
    g.residual = es.x; g.ldr = D; g.C = es.x; g.ldc = D;
    OAVSR_TRY(run_gemm(g, st));
    { ProfScope ps("layernorm", 0, 2.0 * rows * D * m->esz, st);
      OAVSR_TRY(launch_layernorm(dt, es.x, D, es.h, D, L.ln2w, L.ln2b, rows, D, 1e-5f, st)); }
    g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = ffn; g.K = D; g.A = es.h; g.lda = D; g.W = L.fc1w; g.ldw = D; g.col_bias = L.fc1b;
    g.act = ACT_GELU; g.C = es.ffn; g.ldc = ffn;
    OAVSR_TRY(run_gemm(g, st));
    g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = D; g.K = ffn; g.A = es.ffn; g.lda = ffn; g.W = L.fc2w; g.ldw = ffn; g.col_bias = L.fc2b;
    g.residual = es.x; g.ldr = D; g.C = es.x; g.ldc = D;
    OAVSR_TRY(run_gemm(g, st));
  }
  return 0;
}

static int conv2d_gemm(const oavsr_model* m, const void* x, int imgs, int H, int W, int Cin, int Cout, int ks, int stride, const ConvBN& c,
                       const void* residual, bool prelu, void* out, hipStream_t st) {
  if (conv3x3_c64_eligible(m->dtype, H, W, Cin, Cout, ks, stride)) {
    ProfScope ps("conv3x3_c64_bf16", 2.0 * imgs * H * W * 9.0 * Cin * Cout, 2.0 * imgs * H * W * (Cin + Cout * (residual ? 2 : 1)), st);
    return launch_conv3x3_c64(x, imgs, H, W, c.w, c.scale, c.shift, prelu ? c.slope : nullptr, residual, out, st);
  }
  if (conv3x3_c128_eligible(m->dtype, H, W, Cin, Cout, ks, stride)) {
    ProfScope ps("conv3x3_c128_bf16", 2.0 * imgs * H * W * 9.0 * Cin * Cout, 2.0 * imgs * H * W * (Cin + Cout * (residual ? 2 : 1)), st);
    return launch_conv3x3_c128(x, imgs, H, W, c.w, c.scale, c.shift, prelu ? c.slope : nullptr, residual, out, st);
  }
  GemmArgs g; g.dtype = m->dtype; g.amode = A_CONV2D; g.H = H; g.Wd = W; g.Cin = Cin; g.ksize = ks; g.stride = stride; g.pad = ks / 2;
  g.OH = (H + 2 * g.pad - ks) / stride + 1; g.OW = (W + 2 * g.pad - ks) / stride + 1;
  g.M = imgs * g.OH * g.OW; g.N = Cout; g.K = ks * ks * Cin; g.A = x; g.W = c.w; g.ldw = g.K;
  g.col_scale = c.scale; g.col_bias = c.shift; g.residual = residual; g.ldr = Cout;
  if (prelu) { g.act = ACT_PRELU; g.prelu = c.slope; }
  g.C = out; g.ldc = Cout;
  return run_gemm(g, st);
}

static int time_compress(const oavsr_model* m, oavsr_workspace* ws, const EncScratch& es, const void* x, int D, const std::vector<int>& in_off,
                         const std::vector<int>& n_in, int rate, void* out, int* total_out, hipStream_t st) {
  const int B = (int)in_off.size();
  int4* seg_h = ws->stage<int4>(B);
  OAVSR_REQUIRE(seg_h, "host staging exhausted");
  int off = 0, mx = 0;
  for (int b = 0; b < B; ++b) { int n = n_in[b] / rate; seg_h[b] = make_int4(in_off[b], off, n, 0); off += n; mx = std::max(mx, n); }
  *total_out = off;
  OAVSR_TRY(h2d(es.seg, seg_h, sizeof(int4) * B, st));
  const int stack = m->cfg.compression_stack;
  ProfScope ps("time_compress", 0, (double)off * D * (rate + 1) * m->esz, st);
  if (rate == 1 && !stack) {
    // identity compress still goes through the kernel (rate 1 mean == copy)
  }
  return launch_time_compress(m->dtype, x, D, out, stack ? (int64_t)D * rate : D, es.seg, B, mx, rate, D, stack, st);
}

// ------------------------------------------------------------------------------------------------
// encode_video
// ------------------------------------------------------------------------------------------------
static int encode_video_impl(const oavsr_model* m, oavsr_workspace* ws, const void* video, const int32_t* Tv, int B, int rate,
                             void* out_dev, void* enc_dev, int* n_out_host, hipStream_t st) {
  OAVSR_REQUIRE(m->has_video, "model has no video encoder");
  OAVSR_REQUIRE(B >= 1 && B <= ws->max_B, "encode_video: B=%d exceeds workspace (%d)", B, ws->max_B);
  const oavsr_config& c = m->cfg;
  const int dt = m->dtype, D = c.av_dim;
  const EncScratch& es = ws->esv;
  int F = 0, maxT = 0;
  for (int b = 0; b < B; ++b) { OAVSR_REQUIRE(Tv[b] >= 1 && Tv[b] <= ws->max_Tv, "encode_video: Tv[%d]=%d out of range", b, Tv[b]); F += Tv[b]; maxT = std::max(maxT, Tv[b]); }
  int2* fi = ws->stage<int2>(F); int* cu = ws->stage<int>(B + 1);
  OAVSR_REQUIRE(fi && cu, "host staging exhausted");
  { int f = 0; cu[0] = 0;
    for (int b = 0; b < B; ++b) { for (int t = 0; t < Tv[b]; ++t) fi[f++] = make_int2(t, Tv[b]); cu[b + 1] = f; } }
  OAVSR_TRY(h2d(ws->frameinfo, fi, sizeof(int2) * F, st));
  OAVSR_TRY(h2d(es.cu, cu, sizeof(int) * (B + 1), st));
  // K1: stem conv3d + BN + PReLU + MaxPool3d (resnet.py:137-141), one fused kernel (stem.hip)
  { const double in_bytes = ws->vsrc.C ? (double)ws->vsrc.C * ws->vsrc.H * ws->vsrc.W : 7744.0 * 4;
    ProfScope ps(dt == DT_F32 ? "stem_conv_pool_f32" : "stem_conv_pool_bf16", 2.0 * F * 1936 * 245 * 64, (double)F * (in_bytes + 484 * 64 * m->esz), st);
    OAVSR_TRY(launch_stem_pool(dt, video, ws->frameinfo, F, m->stem_wp, m->stem.scale, m->stem.shift, m->stem.slope, ws->pool_out, st, ws->vsrc)); }
  GemmArgs g;
  // K2: ResNet-18 trunk (resnet.py:61-74, 121-126)
  // x lives in pool_out and is updated in place: conv2's epilogue reads residual[m][n] and writes C[m][n] from the
  // same thread, and conv1 / downsample (the only readers of x as a GEMM operand) precede it in stream order.
  void* x = ws->pool_out;
  int H = 22;
  for (int i = 0; i < 8; ++i) {
    const ResBlock& rb = m->blocks[i];
    void* tmp = ws->trunk[0]; void* res = ws->trunk[1];
    const int OHo = (H - 1) / rb.stride + 1;
    OAVSR_TRY(conv2d_gemm(m, x, F, H, H, rb.cin, rb.cout, 3, rb.stride, rb.c1, nullptr, true, tmp, st));
    const void* residual = x;
    if (rb.has_down) { OAVSR_TRY(conv2d_gemm(m, x, F, H, H, rb.cin, rb.cout, 1, rb.stride, rb.down, nullptr, false, res, st)); residual = res; }
    OAVSR_TRY(conv2d_gemm(m, tmp, F, OHo, OHo, rb.cout, rb.cout, 3, 1, rb.c2, residual, true, x, st));
    H = OHo;
  }
  // K3: avgpool -> Linear 512->D -> LN([0,x]) -> Linear 2D->D   (resnet.py:127-128, hubert.py:327,709-728)
  { ProfScope ps("avgpool", 0, (double)F * (H * H + 1) * 512 * m->esz, st);
    OAVSR_TRY(launch_avgpool_hw(dt, x, ws->feat512, F, H * H, 512, st)); }
  g = GemmArgs(); g.dtype = dt; g.M = F; g.N = D; g.K = 512; g.A = ws->feat512; g.lda = 512; g.W = m->vproj_w; g.ldw = 512; g.col_bias = m->vproj_b;
  g.C = ws->featproj; g.ldc = D;
  OAVSR_TRY(run_gemm(g, st));
  { ProfScope ps("layernorm", 0, 3.0 * F * D * m->esz, st);
    OAVSR_TRY(launch_layernorm_zero_prefix(dt, ws->featproj, D, ws->ln2d, 2 * D, m->vln_w, m->vln_b, F, D, 1e-5f, st)); }
  g = GemmArgs(); g.dtype = dt; g.M = F; g.N = D; g.K = 2 * D; g.A = ws->ln2d; g.lda = 2 * D; g.W = m->vpost_w; g.ldw = 2 * D; g.col_bias = m->vpost_b;
  g.C = es.h; g.ldc = D;   // x0 in enc_h
  OAVSR_TRY(run_gemm(g, st));
  // K4: x = x0 + GELU(pos_conv(x0))  (wav2vec2.py:875-877)
  OAVSR_TRY(launch_build_rowinfo(es.cu, es.cu, B, 1, es.rowinfo, F, st));
  const int G = c.av_conv_groups, Cg = D / G, taps = c.av_conv_pos;
  g = GemmArgs(); g.dtype = dt; g.amode = A_CONV1D; g.M = F; g.N = Cg; g.K = taps * Cg; g.Kvalid = taps * Cg; g.taps = taps; g.pad = taps / 2;
  g.Cin = Cg; g.A = es.h; g.lda = D; g.groups = G; g.a_goff = Cg; g.rowinfo = es.rowinfo; g.W = m->vpos_w; g.ldw = g.K;
  g.col_bias = m->vpos_b; g.act = ACT_GELU; g.residual = es.h; g.ldr = D; g.res_after_act = 1; g.C = es.x; g.ldc = D;
  OAVSR_TRY(run_gemm(g, st));
  OAVSR_TRY(run_encoder_layers(m, es, m->vlayers, F, D, c.av_heads, c.av_ffn, c.av_lora_k2, es.cu, B, maxT, st, 32));      // attention: the 32x32x16-MFMA kernel, like the Whisper encoder (VSR B = 16: 0.90 -> 0.72 ms per call)
  void* enc = ws->enc_out_v;
  { ProfScope ps("layernorm", 0, 2.0 * F * D * m->esz, st);
    OAVSR_TRY(launch_layernorm(dt, es.x, D, enc, D, m->vlnf_w, m->vlnf_b, F, D, 1e-5f, st)); }
  if (enc_dev) OAVSR_CHECK_HIP(hipMemcpyAsync(enc_dev, enc, (size_t)F * D * m->esz, hipMemcpyDeviceToDevice, st));
  std::vector<int> in_off(B), n_in(B);
  for (int b = 0; b < B; ++b) { in_off[b] = cu[b]; n_in[b] = Tv[b]; }
  int total = 0;
  OAVSR_TRY(time_compress(m, ws, es, enc, D, in_off, n_in, rate, ws->comp_v, &total, st));
  const int Dout = c.compression_stack ? D * rate : D;
  if (out_dev) OAVSR_CHECK_HIP(hipMemcpyAsync(out_dev, ws->comp_v, (size_t)total * Dout * m->esz, hipMemcpyDeviceToDevice, st));
  if (n_out_host) for (int b = 0; b < B; ++b) n_out_host[b] = Tv[b] / rate;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// encode_audio
// ------------------------------------------------------------------------------------------------
static int encode_audio_impl(const oavsr_model* m, oavsr_workspace* ws, const float* wav, int64_t ld, const int32_t* Ts, const int32_t* keep,
                             int B, int rate, void* out_dev, void* enc_dev, void* mel_dev, int* n_out_host, hipStream_t st) {
  OAVSR_REQUIRE(m->has_audio, "model has no audio encoder");
  OAVSR_REQUIRE(B >= 1 && B <= ws->max_B, "encode_audio: B=%d exceeds workspace (%d)", B, ws->max_B);
  const oavsr_config& c = m->cfg;
  const int dt = m->dtype, D = c.wh_dim;
  const EncScratch& es = ws->esa;
  int* lens = ws->stage<int>(B); int* cu3000 = ws->stage<int>(B + 1); int* cu1500 = ws->stage<int>(B + 1);
  OAVSR_REQUIRE(lens && cu3000 && cu1500, "host staging exhausted");
  for (int b = 0; b < B; ++b) {
    OAVSR_REQUIRE(Ts[b] >= 0 && Ts[b] <= ld, "encode_audio: Ts[%d]=%d exceeds row stride", b, Ts[b]);
    OAVSR_REQUIRE(keep[b] >= 1 && keep[b] <= 1500, "encode_audio: keep[%d]=%d out of range", b, keep[b]);
    lens[b] = Ts[b]; cu3000[b] = b * 3000; cu1500[b] = b * 1500;
  }
  cu3000[B] = B * 3000; cu1500[B] = B * 1500;
  OAVSR_TRY(h2d(ws->lens_dev, lens, sizeof(int) * B, st));
  OAVSR_TRY(h2d(es.cu, cu3000, sizeof(int) * (B + 1), st));
  OAVSR_TRY(h2d(ws->cu_b, cu1500, sizeof(int) * (B + 1), st));
  { ProfScope ps("logmel", 2.0 * B * 3000.0 * 201 * 400 * 2, (double)B * 480000 * 4, st);
    OAVSR_TRY(launch_logmel(dt, wav, ld, ws->lens_dev, B, m->twiddle, m->melfb, ws->mel_ws, ws->max_ws, ws->mel, 80, st)); }
  if (mel_dev) OAVSR_CHECK_HIP(hipMemcpyAsync(mel_dev, ws->mel, (size_t)B * 3000 * 80 * m->esz, hipMemcpyDeviceToDevice, st));
  // K9: conv1 (k3,p1)+GELU ; conv2 (k3,s2,p1)+GELU ; + embed_positions   (transformers WhisperEncoder.forward)
  OAVSR_TRY(launch_build_rowinfo(es.cu, es.cu, B, 1, es.rowinfo, B * 3000, st));
  GemmArgs g; g.dtype = dt; g.amode = A_CONV1D; g.M = B * 3000; g.N = D; g.K = 256; g.Kvalid = 240; g.taps = 3; g.pad = 1; g.Cin = 80;
  g.A = ws->mel; g.lda = 80; g.rowinfo = es.rowinfo; g.W = m->aconv1_w; g.ldw = 256; g.col_bias = m->aconv1_b; g.act = ACT_GELU;
  g.C = ws->conv1_out; g.ldc = D;
  OAVSR_TRY(run_gemm(g, st));
  OAVSR_TRY(launch_build_rowinfo(ws->cu_b, es.cu, B, 2, ws->rowinfo2, B * 1500, st));
  g = GemmArgs(); g.dtype = dt; g.amode = A_CONV1D; g.M = B * 1500; g.N = D; g.K = 3 * D; g.Kvalid = 3 * D; g.taps = 3; g.pad = 1; g.Cin = D;
  g.A = ws->conv1_out; g.lda = D; g.rowinfo = ws->rowinfo2; g.W = m->aconv2_w; g.ldw = 3 * D; g.col_bias = m->aconv2_b; g.act = ACT_GELU;
  g.residual = ws->pos_rep; g.ldr = D; g.res_after_act = 1; g.C = es.x; g.ldc = D;
  OAVSR_TRY(run_gemm(g, st));
  // attention on the 32x32x16-MFMA kernel: chosen by the call site (the encoders), so the choice never depends on a batch
  OAVSR_TRY(run_encoder_layers(m, es, m->alayers, B * 1500, D, c.wh_heads, c.wh_ffn, 0, ws->cu_b, B, 1500, st, 32));
  void* enc = ws->enc_out_a;
  { ProfScope ps("layernorm", 0, 2.0 * B * 1500 * D * m->esz, st);
    OAVSR_TRY(launch_layernorm(dt, es.x, D, enc, D, m->alnf_w, m->alnf_b, B * 1500, D, 1e-5f, st)); }
  if (enc_dev) OAVSR_CHECK_HIP(hipMemcpyAsync(enc_dev, enc, (size_t)B * 1500 * D * m->esz, hipMemcpyDeviceToDevice, st));
  std::vector<int> in_off(B), n_in(B);
  for (int b = 0; b < B; ++b) { in_off[b] = b * 1500; n_in[b] = keep[b]; }   // truncation modeling_OmniAVSR.py:537
  int total = 0;
  OAVSR_TRY(time_compress(m, ws, es, enc, D, in_off, n_in, rate, ws->comp_a, &total, st));
  const int Dout = c.compression_stack ? D * rate : D;
  if (out_dev) OAVSR_CHECK_HIP(hipMemcpyAsync(out_dev, ws->comp_a, (size_t)total * Dout * m->esz, hipMemcpyDeviceToDevice, st));
  if (n_out_host) for (int b = 0; b < B; ++b) n_out_host[b] = keep[b] / rate;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// projector
// ------------------------------------------------------------------------------------------------
static int project_impl(const oavsr_model* m, oavsr_workspace* ws, int which, int rate, const void* feats, int rows, void* out, hipStream_t st) {
  const oavsr_config& c = m->cfg;
  const auto& vec = which == 0 ? m->proj_a : m->proj_v;
  OAVSR_REQUIRE(!vec.empty(), "project: modality %d has no projector", which);
  int idx = 0;
  const int n_rates = which == 0 ? c.n_rates_audio : c.n_rates_video;
  const int* rates = which == 0 ? c.rates_audio : c.rates_video;
  int found = -1;
  for (int i = 0; i < n_rates; ++i) if (rates[i] == rate) found = i;
  // the reference looks the rate up in matry_map_* (modeling_OmniAVSR.py:429,447) -> KeyError for an unknown rate
  OAVSR_REQUIRE(found >= 0, "KeyError: compression rate %d is not in the configured list", rate);
  if (c.proj_per_rate) idx = found;
  const Projector& P = vec[idx];
  const int Denc = which == 0 ? c.wh_dim : c.av_dim;
  const int Din = c.compression_stack ? Denc * rate : Denc;
  const int I = c.proj_inter, d = c.llm_hidden, dt = m->dtype;
  GemmArgs g; g.dtype = dt; g.M = rows; g.N = I; g.K = Din; g.A = feats; g.lda = Din; g.W = P.fc1w; g.ldw = Din; g.col_bias = P.fc1b; g.act = ACT_RELU;
  g.C = ws->proj_h; g.ldc = I;
  OAVSR_TRY(run_gemm(g, st));
  g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = d; g.K = I; g.A = ws->proj_h; g.lda = I; g.W = P.fc2w; g.ldw = I; g.col_bias = P.fc2b;
  g.C = out; g.ldc = d;
  OAVSR_TRY(run_gemm(g, st));
  if (P.lnw) { ProfScope ps("layernorm", 0, 2.0 * rows * d * m->esz, st);
    OAVSR_TRY(launch_layernorm(dt, out, d, out, d, P.lnw, P.lnb, rows, d, 1e-5f, st)); }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// prompt splice (modeling_OmniAVSR.py:406-458)
// ------------------------------------------------------------------------------------------------
static int build_prompt_impl(const oavsr_model* m, oavsr_workspace* ws, int task, int B, const int32_t* na, const void* atok,
                             const int32_t* nv, const void* vtok, void* embeds, int32_t* S_out, hipStream_t st) {
  const oavsr_config& c = m->cfg;
  OAVSR_REQUIRE(task >= 0 && task < 3, "bad task %d", task);
  const int P = c.prompt_len[task], d = c.llm_hidden;
  OAVSR_REQUIRE(P == 0 || m->prompt[task], "prompt buffer for task %d missing", task);
  int total = 0;
  for (int b = 0; b < B; ++b) {
    int S = (c.llm_family == 0 ? 1 : 0) + (na ? na[b] + 2 : 0) + (nv ? nv[b] + 2 : 0) + P;
    S_out[b] = S; total += S;
  }
  OAVSR_REQUIRE(total <= ws->max_S_total, "build_prompt: %d prompt rows exceed workspace (%d)", total, ws->max_S_total);
  int* maps = ws->stage<int>((size_t)4 * total);
  OAVSR_REQUIRE(maps, "host staging exhausted");
  int* m_emb = maps; int* m_a = maps + total; int* m_v = maps + 2 * total; int* m_p = maps + 3 * total;
  for (int i = 0; i < 4 * total; ++i) maps[i] = -1;
  int r = 0, ao = 0, vo = 0;
  for (int b = 0; b < B; ++b) {
    // final order: [BOS] <audio> A </audio> <video> V </video> prompt   (Qwen: no BOS)  — :434-437,452-455
    if (c.llm_family == 0) m_emb[r++] = c.id_bos;
    if (na) { m_emb[r++] = c.id_audio_sos; for (int i = 0; i < na[b]; ++i) m_a[r++] = ao++; m_emb[r++] = c.id_audio_eos; }
    if (nv) { m_emb[r++] = c.id_video_sos; for (int i = 0; i < nv[b]; ++i) m_v[r++] = vo++; m_emb[r++] = c.id_video_eos; }
    for (int i = 0; i < P; ++i) m_p[r++] = i;
  }
  OAVSR_TRY(h2d(ws->maps, maps, sizeof(int) * 4 * total, st));
  const int dt = m->dtype;
  ProfScope ps("prompt_splice", 0, 2.0 * total * d * m->esz, st);
  OAVSR_TRY(launch_copy_rows(dt, m->embed, d, ws->maps, embeds, d, total, d, st));
  if (na) OAVSR_TRY(launch_copy_rows(dt, atok, d, ws->maps + total, embeds, d, total, d, st));
  if (nv) OAVSR_TRY(launch_copy_rows(dt, vtok, d, ws->maps + 2 * total, embeds, d, total, d, st));
  if (P) OAVSR_TRY(launch_copy_rows(dt, m->prompt[task], d, ws->maps + 3 * total, embeds, d, total, d, st));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// LLM layer stack; rows = prompt rows (prefill: the MFMA-bound GEMM families) or decode-shaped rows
// ------------------------------------------------------------------------------------------------
// DECODE-SHAPED GEMMs (one new token per row: greedy and beam-search steps, the decode scheduler's steps, and the pruned tail of the
// prefill's last layer + the first lm_head) all run ONE regime, whatever the row count: the tiled LDS-DMA kernels (128 x 64 tiles up to
// 128 rows, 256 x 128 above), K split over workgroups with fp32 slabs, the slabs summed in slab order by the tile's last-arriving
// workgroup inside the launch (gemm2.hip).  Rows are independent and the split factor is a function of N and K ONLY, so a row's bits do
// not depend on how many rows share the launch: an utterance decodes to the same ids alone, in a batch, as one of K beams x B
// utterances or inside a 128-row scheduler step, in f32 and in bf16.
// RMSNorm is DEFERRED (Llama_LoRA.py:587-655 computes norm(x) W^T): the GEMM that produces the residual stream x (o_proj, down_proj) also
// stores bf16(x * gain_next) and 1 / rms(x) per row (GemmArgs::ng_*); the consumer takes the former as its A operand and multiplies its
// fp32 accumulator by the latter (row_scale) -- six launches per layer (LoRA-down, QKV, attention, o_proj, gate/up + SwiGLU, down_proj).
// Split factor of a decode-shaped GEMM: a function of N and K ONLY (part of the regime's summation order).  As many k-slices as give <= 256
// workgroups (one per CU) with at least EIGHT k-tiles each: below that the fp32 slabs -- written through to memory and read back by the tile's
// last workgroup -- cost more than the extra workgroups stream (tools/decode_split_scan.py, 97 rows: o_proj 2048 x 2048 17.8 us at 4 slices,
// 22.9 at 8, 33.8 at 16; QKV 3072 x 2304 20.0 / 32.3 / 56.3; 8 rows: within 1 us of the best choice everywhere).
constexpr int kDecodeSplitTarget = 256, kDecodeMinKtiles = 8;
constexpr int kChainInts = 1024, kChainErr = kChainInts - 1;      // oavsr_workspace::chain_ctr
static int decode_split(int N, int Ktot, int esz, int min_ks = 1) {
  const int ktiles = Ktot / (128 / esz);
  int ks = kDecodeSplitTarget / std::max(1, ceil_div(N, 64));
  ks = std::min(ks, std::max(1, ktiles / kDecodeMinKtiles));
  ks = std::max(1, std::min(ks, 16));
  return std::max(ks, std::min(min_ks, std::max(1, ktiles)));
}

// ws->rowsq: per 128-row chunk of the decode rows, [column tile of the producer][row in chunk] partial sums of x^2
static size_t rowsq_chunk(int d) { return (size_t)(ceil_div(d, 64) + 1) * 128; }

// One decode-shaped GEMM over g.M rows, 128 rows (one M-tile of the 128-row tiles) per launch -- every kernel on this path then has the
// row-factor epilogue and the in-launch reduce, whatever the row count (beam search with B x K > 128 rows: two launches).
// sq_tiles > 0: A is T(x * gain) of a deferred RMSNorm; the accumulator is multiplied by 1 / rms(x) from the `sq_tiles` partial sums per row
//   that the producer of A left in ws->rowsq (rms_prep: 1; the reduce of o_proj / down_proj: d / 64).
// next_gain: the GEMM's output is the residual stream; also write T(x * next_gain) to `ng_out` and the rows' partial sums to ws->rowsq.
constexpr int kSqPrep = -1;      // decode_gemm's sq_tiles: ONE sum of squares per row, left by rms_prep (not by a reduce)

// The collected strip launches go out: one launch with a phase per GEMM (a single one: the plain launch).
static int chain_flush(oavsr_workspace* ws, hipStream_t st) {
  auto& ch = ws->chain;
  const int n = ch.n;
  ch.n = 0;
  if (n == 0) return 0;
  if (n == 1) return run_gemm(ch.g[0], st);
  OAVSR_REQUIRE(ch.used + n <= kChainErr, "decode chain: more phases in a step than counters");
  int* ctr = ws->chain_ctr + ch.used;
  ch.used += n;
  return launch_gemm_strip_chain(ch.g, n, ctr, ws->chain_ctr + kChainErr, st);
}
struct ChainScope {      // a decode step's collection window: closed on every way out
  oavsr_workspace* ws;
  ~ChainScope() { ws->chain.open = false; ws->chain.n = 0; }
};
static int decode_gemm(const oavsr_model* m, oavsr_workspace* ws, const GemmArgs& g0, hipStream_t st, int sq_tiles = 0, const float* next_gain = nullptr,
                       void* ng_out = nullptr) {
  const size_t e = m->esz;
  const int d = m->cfg.llm_hidden;
  const int ks = g0.swiglu_pairs ? 1 : decode_split(g0.N, g0.K + g0.K2, (int)e, next_gain ? 2 : 1);
  OAVSR_REQUIRE(!next_gain || (ks > 1 && g0.N == d), "decode GEMM %d x %d x %d: the next-norm outputs need a split-K launch over the hidden width", g0.M, g0.N, g0.K + g0.K2);
  for (int r0 = 0; r0 < g0.M; r0 += 128) {
    GemmArgs g = g0;
    g.M = std::min(128, g0.M - r0);
    g.decode_tiled = 1;
    auto adv = [&](const void* p, int64_t ld, size_t es) -> const void* { return p ? (const char*)p + (size_t)r0 * ld * es : nullptr; };
    g.A = adv(g0.A, g0.lda, e); g.A2 = adv(g0.A2, g0.lda2, e); g.residual = adv(g0.residual, g0.ldr, e);
    g.C = (void*)adv(g0.C, g0.ldc, g0.out_f32 ? 4 : e);
    float* sq = ws->rowsq + (size_t)(r0 / 128) * rowsq_chunk(d);
    // <= kStripRows rows of a chunk: the register-streaming strips (decode_strip.hip) -- every GEMM of the step sees the same chunk sizes, so
    // producer and consumer of a chunk's sums of squares agree on their layout (per 16-column strip there, per 64-column tile on the tiled path)
    const bool strips = gemm_strip_rows(g.M);
    if (sq_tiles != 0) {
      g.row_sumsq = sq; g.row_sumsq_tiles = sq_tiles == kSqPrep ? 1 : sq_tiles; g.row_dim = d; g.row_eps = m->cfg.llm_rms_eps;
      g.row_sumsq_form = (strips && sq_tiles != kSqPrep) ? 1 : 0;
    }
    if (ks > 1) {
      // up to kFusedRows rows the tile's last-arriving workgroup reduces the slabs inside the launch (one launch instead of two: what
      // counts when a launch is mostly fixed cost); above, that workgroup would be a serial tail -- the same reduce runs as a launch of
      // its own over ~4 x the workgroups.  Same code either way: never a different bit (tests compare them).  (The strips need neither.)
      static const int kFusedRows = knob("OAVSR_FUSED_REDUCE_ROWS");
      g.ksplit = ks;
      if (!strips) {
        OAVSR_REQUIRE((size_t)ks * g.M * g.N * 4 <= ws->partial_bytes, "decode GEMM: split-K slabs (%d x %d x %d) exceed the workspace", ks, g.M, g.N);
        g.partial = ws->partial; g.sync = g.M <= kFusedRows ? ws->gsync : nullptr;
      } else if ((size_t)g.M * g.N * 4 <= ws->partial_bytes && ceil_div(g.N, 16) <= kSyncInts) {
        g.partial = ws->partial; g.sync = ws->gsync;      // the strips' pairs (decode_strip.hip): M x 16 fp32 sums and a counter per strip
      }
      if (next_gain) { g.ng_out = (char*)ng_out + (size_t)r0 * g.N * e; g.ng_ld = g.N; g.ng_gain = next_gain; g.ng_rowsq = sq; g.rowsq_strips = strips ? 1 : 0; }
    }
    if (!strips) OAVSR_REQUIRE(gemm_dma_eligible(g), "decode GEMM %d x %d x %d: no LDS-DMA tile (N >= 64 needed)", g.M, g.N, g.K + g.K2);
    if (ws->chain.open && gemm_strip_selected(g)) {      // one of a chain of strip launches: collected (chain_flush)
      if (ws->chain.n == kStripChainMax) OAVSR_TRY(chain_flush(ws, st));
      ws->chain.g[ws->chain.n++] = g;
      continue;
    }
    if (ws->chain.n) OAVSR_TRY(chain_flush(ws, st));
    OAVSR_TRY(run_gemm(g, st));
  }
  return 0;
}

// the deferred RMSNorm of rows no GEMM produced (a decode step's input embeddings): xg = T(x * gain), one partial sum of squares per row
static int decode_rms_prep(const oavsr_model* m, oavsr_workspace* ws, const void* x, void* xg, const float* gain, int rows, hipStream_t st) {
  const int d = m->cfg.llm_hidden;
  ProfScope ps("decode_rms_prep", 0, 2.0 * rows * d * m->esz, st);
  for (int r0 = 0; r0 < rows; r0 += 128)
    OAVSR_TRY(launch_rms_prep(m->dtype, (const char*)x + (size_t)r0 * d * m->esz, d, (char*)xg + (size_t)r0 * d * m->esz, d, gain,
                              ws->rowsq + (size_t)(r0 / 128) * rowsq_chunk(d), std::min(128, rows - r0), d, st));
  return 0;
}

// Everything of a decoder layer behind the attention, on decode-shaped rows: o_proj (+ residual; prepares ln2), gate / up + SwiGLU,
// down_proj (+ residual; prepares `next_gain`: the next layer's ln1 or the final norm).  xres: residual stream rows (in place); xg: where the
// two norms' operands T(x * gain) go (and come from); the rows' sums of squares travel in ws->rowsq.
static int decode_layer_tail(const oavsr_model* m, oavsr_workspace* ws, const LlmLayer& L, int rows, void* xres, const void* attn_r, void* act_r,
                             void* xg, const float* next_gain, hipStream_t st) {
  const oavsr_config& c = m->cfg;
  const int dt = m->dtype, d = c.llm_hidden, nq = c.llm_heads * c.llm_head_dim, I = c.llm_mlp;
  const int dtiles = ceil_div(d, 64);
  GemmArgs g; g.dtype = dt; g.M = rows; g.N = d; g.K = nq; g.A = attn_r; g.lda = nq; g.W = L.ow; g.ldw = nq;
  g.residual = xres; g.ldr = d; g.C = xres; g.ldc = d;
  OAVSR_TRY(decode_gemm(m, ws, g, st, 0, L.ln2w, xg));
  g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = 2 * I; g.K = d; g.A = xg; g.lda = d; g.W = L.guw; g.ldw = d;
  GemmArgs gf = g; gf.swiglu_pairs = 1; gf.C = act_r; gf.ldc = I; gf.decode_tiled = 1;
  // SwiGLU in the GEMM epilogue when the 128 x 64 tiles fill the chip without a split (every real model: 2 I / 64 >= 256 column tiles)
  GemmArgs gp = gf; gp.M = std::min(rows, 128);
  if (decode_split(g.N, g.K, (int)m->esz) == 1 && gemm_dma_swiglu_ok(gp)) {
    OAVSR_TRY(decode_gemm(m, ws, gf, st, dtiles));
  } else {
    g.C = ws->lgu; g.ldc = 2 * I;
    OAVSR_TRY(decode_gemm(m, ws, g, st, dtiles));
    ProfScope ps("decode_swiglu", 0, 3.0 * rows * I * m->esz, st);
    OAVSR_TRY(launch_swiglu(dt, ws->lgu, 2 * I, act_r, I, rows, I, 1, st));
  }
  g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = d; g.K = I; g.A = act_r; g.lda = I; g.W = L.downw; g.ldw = I;
  g.residual = xres; g.ldr = d; g.C = xres; g.ldc = d;
  return decode_gemm(m, ws, g, st, 0, next_gain, xg);
}

// One decode step through every layer.  In: ws->lx = the rows' input embeddings.  Out: ws->lx = final hidden rows, ws->lh = T(x * norm.w),
// ws->rowsq = the rows' sums of squares: lm_head's operands (lm_head_rows).
static int llm_decode_layers(const oavsr_model* m, oavsr_workspace* ws, int task, int rows, int beams, int max_S, const int* pos, const int* cpos,
                             const int* anc, const int* step, hipStream_t st, const float2* rope_cs = nullptr) {
  const oavsr_config& c = m->cfg;
  const int dt = m->dtype, d = c.llm_hidden, H = c.llm_heads, KV = c.llm_kv_heads, hd = c.llm_head_dim;
  const int nq = H * hd, nqkv = (H + 2 * KV) * hd;
  const int tl = c.lora_task_specific ? task : 0;
  // <= 16 rows (the strips): the GEMMs between two attention launches go out as the phases of one launch
  ChainScope scope{ws};
  if (gemm_strip_rows(rows) && !g_prof.on && knob("OAVSR_DECODE_CHAIN")) {
    ws->chain.open = true; ws->chain.n = 0; ws->chain.used = 0;      // (the counters are zero: workspace creation, then every launch cleans up behind itself)
  }
  OAVSR_TRY(decode_rms_prep(m, ws, ws->lx, ws->lh, m->llayers[0].ln1w, rows, st));
  int sq_tiles = kSqPrep;                      // partial sums per row in ws->rowsq: one behind rms_prep, d / 64 tiles behind a reduce
  for (int li = 0; li < c.llm_layers; ++li) {
    const LlmLayer& L = m->llayers[li];
    GemmArgs g;
    if (c.lora_k2 > 0) {      // t = (x * g) A^T, rounded to T WITHOUT 1 / rms: the QKV launch scales both of its K segments by it
      g.dtype = dt; g.M = rows; g.N = c.lora_k2; g.K = d; g.A = ws->lh; g.lda = d; g.W = L.lora_dn[tl]; g.ldw = d; g.C = ws->llora; g.ldc = c.lora_k2;
      OAVSR_TRY(decode_gemm(m, ws, g, st));
    }
    g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = nqkv; g.K = d; g.A = ws->lh; g.lda = d; g.W = L.qkvw; g.ldw = d; g.col_bias = L.qkvb;
    if (c.lora_k2 > 0) { g.A2 = ws->llora; g.lda2 = c.lora_k2; g.W2 = L.lora_up[tl]; g.ldw2 = c.lora_k2; g.K2 = c.lora_k2; }
    g.C = ws->lqkv; g.ldc = nqkv;
    OAVSR_TRY(decode_gemm(m, ws, g, st, sq_tiles));
    // RoPE + cache append are fused into the decode-attention kernel
    DecodeAttnArgs a; a.dtype = dt; a.head_dim = hd; a.n_heads = H; a.n_kv_heads = KV; a.rows = rows; a.beams = beams;
    a.Q = ws->lqkv; a.ldq = nqkv; a.O = ws->lattn; a.ldo = nq;
    a.Kp = (char*)ws->kpre + li * ws->kpre_layer; a.Vp = (char*)ws->vpre + li * ws->kpre_layer; a.max_prefix = ws->max_prefix;
    a.Kt = (char*)ws->ktail + li * ws->ktail_layer; a.Vt = (char*)ws->vtail + li * ws->ktail_layer; a.max_tail = ws->max_new;
    a.prefix_len = ws->prefix_len; a.step = step; a.anc = anc; a.scale = 1.0f / sqrtf((float)hd);
    a.inv_freq = m->inv_freq; a.pos = pos; a.cpos = cpos; a.rope_cs = rope_cs; a.rowmap = ws->rowmap;
    OAVSR_TRY(chain_flush(ws, st));
    { ProfScope ps("decode_attention", 0, 2.0 * rows * max_S * KV * hd * m->esz, st);
      OAVSR_TRY(launch_decode_attention(a, st)); }
    const float* next_gain = li + 1 < c.llm_layers ? m->llayers[li + 1].ln1w : m->norm_w;
    OAVSR_TRY(decode_layer_tail(m, ws, L, rows, ws->lx, ws->lattn, ws->lact, ws->lh, next_gain, st));
    sq_tiles = ceil_div(d, 64);
  }
  return chain_flush(ws, st);
}

// Prefill: all prompt rows of all utterances through the MFMA-bound GEMM families (var-len causal attention, prefix KV written by the RoPE
// kernel).  last_rows / cu_last: row index of every utterance's final prompt token and the device array 0..B.  Only those rows feed lm_head
// (the reference materialises all positions, Llama_LoRA.py:372-373), so in the LAST layer everything after the KV append -- attention
// queries, o_proj, the MLP -- runs on the B gathered rows (compact residual stream ws->lastx) through the decode-shaped path above, 128
// rows at a time: an utterance's first-token logits do not depend on its batch.  On return ws->lastx holds the final hidden rows, ws->lh /
// ws->rowsq the final norm's operands for lm_head_rows.
static int llm_prefill_layers(const oavsr_model* m, oavsr_workspace* ws, int task, int rows, int B, int max_S, const int* pos, const int* cpos,
                              const int* last_rows, const int* cu_last, hipStream_t st) {
  const oavsr_config& c = m->cfg;
  const int dt = m->dtype, d = c.llm_hidden, H = c.llm_heads, KV = c.llm_kv_heads, hd = c.llm_head_dim, I = c.llm_mlp;
  const int nq = H * hd, nqkv = (H + 2 * KV) * hd;
  const int tl = c.lora_task_specific ? task : 0;
  for (int li = 0; li < c.llm_layers; ++li) {
    const LlmLayer& L = m->llayers[li];
    const bool prune = li == c.llm_layers - 1;
    { ProfScope ps("rmsnorm", 0, 2.0 * rows * d * m->esz, st);
      OAVSR_TRY(launch_rmsnorm(dt, ws->lx, d, ws->lh, d, L.ln1w, rows, d, c.llm_rms_eps, st)); }
    GemmArgs g;
    if (c.lora_k2 > 0) {
      g.dtype = dt; g.M = rows; g.N = c.lora_k2; g.K = d; g.A = ws->lh; g.lda = d; g.W = L.lora_dn[tl]; g.ldw = d; g.C = ws->llora; g.ldc = c.lora_k2;
      OAVSR_TRY(run_gemm(g, st));
    }
    g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = nqkv; g.K = d; g.A = ws->lh; g.lda = d; g.W = L.qkvw; g.ldw = d; g.col_bias = L.qkvb;
    if (c.lora_k2 > 0) { g.A2 = ws->llora; g.lda2 = c.lora_k2; g.W2 = L.lora_up[tl]; g.ldw2 = c.lora_k2; g.K2 = c.lora_k2; }
    g.C = ws->lqkv; g.ldc = nqkv;
    OAVSR_TRY(run_gemm(g, st));
    RopeArgs r; r.dtype = dt; r.rows = rows; r.n_heads = H; r.n_kv_heads = KV; r.head_dim = hd; r.qkv = ws->lqkv; r.ld = nqkv;
    r.inv_freq = m->inv_freq; r.pos = pos; r.slot = ws->slot; r.cpos = cpos;
    r.kcache = (char*)ws->pf_k + li * ws->pf_layer; r.vcache = (char*)ws->pf_v + li * ws->pf_layer; r.max_ctx = ws->pf_max_prefix;
    { ProfScope ps("rope_kv", 0, 2.0 * rows * nqkv * m->esz, st);
      OAVSR_TRY(launch_rope_kv(r, st)); }
    AttnArgs a; a.dtype = dt; a.head_dim = hd; a.n_heads = H; a.n_kv_heads = KV; a.n_seqs = B;
    a.Q = ws->lqkv; a.K = (const char*)ws->lqkv + (size_t)nq * m->esz; a.V = (const char*)ws->lqkv + (size_t)(nq + KV * hd) * m->esz;
    a.ldq = a.ldk = a.ldv = nqkv; a.O = ws->lattn; a.ldo = nq; a.cu_q = ws->cu_s; a.cu_k = ws->cu_s; a.max_q = max_S;
    a.scale = 1.0f / sqrtf((float)hd); a.causal = 1;
    if (prune) {   // one query row per utterance: gathered q rows (ws->lh is free after the QKV GEMM) against the full K / V
      OAVSR_TRY(launch_copy_rows(dt, ws->lqkv, nqkv, last_rows, ws->lh, nqkv, B, nqkv, st));
      OAVSR_TRY(launch_copy_rows(dt, ws->lx, d, last_rows, ws->lastx, d, B, d, st));
      a.Q = ws->lh; a.cu_q = cu_last; a.max_q = 1;
    }
    { ProfScope ps(dt == DT_F32 ? "attention_f32" : "attention_bf16", 2.0 * (prune ? B : rows) * max_S * nq, 4.0 * (prune ? B : rows) * nq * m->esz, st);
      OAVSR_TRY(launch_attention(a, st)); }
    if (prune) {      // (ws->lh held the gathered queries: free again)
      OAVSR_TRY(decode_layer_tail(m, ws, L, B, ws->lastx, ws->lattn, ws->lact, ws->lh, m->norm_w, st));
      break;
    }
    g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = d; g.K = nq; g.A = ws->lattn; g.lda = nq; g.W = L.ow; g.ldw = nq;
    g.residual = ws->lx; g.ldr = d; g.C = ws->lx; g.ldc = d;
    OAVSR_TRY(run_gemm(g, st));
    { ProfScope ps("rmsnorm", 0, 2.0 * rows * d * m->esz, st);
      OAVSR_TRY(launch_rmsnorm(dt, ws->lx, d, ws->lh, d, L.ln2w, rows, d, c.llm_rms_eps, st)); }
    g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = 2 * I; g.K = d; g.A = ws->lh; g.lda = d; g.W = L.guw; g.ldw = d; g.C = ws->lgu; g.ldc = 2 * I;
    GemmArgs gf = g; gf.swiglu_pairs = 1; gf.C = ws->lact; gf.ldc = I;
    if (gemm_dma_swiglu_ok(gf)) {      // SwiGLU in the GEMM epilogue: the [rows][2I] tensor never exists
      OAVSR_TRY(run_gemm(gf, st));
    } else {
      OAVSR_TRY(run_gemm(g, st));
      ProfScope ps("swiglu", 0, 3.0 * rows * I * m->esz, st);
      OAVSR_TRY(launch_swiglu(dt, ws->lgu, 2 * I, ws->lact, I, rows, I, 1, st));
    }
    g = GemmArgs(); g.dtype = dt; g.M = rows; g.N = d; g.K = I; g.A = ws->lact; g.lda = I; g.W = L.downw; g.ldw = I;
    g.residual = ws->lx; g.ldr = d; g.C = ws->lx; g.ldc = d;
    OAVSR_TRY(run_gemm(g, st));
  }
  return 0;
}

// logits[rows][vocab] (fp32, Llama_LoRA.py:372-373) of decode-shaped rows: A = ws->lh rows (T(x * norm.w), written by the last
// down_proj's reduce), accumulator x 1 / rms(x) from the same reduce's sums of squares
// Row stride of the engine's own logits buffers: the vocabulary rounded up to 16 floats, so that every row starts on a 64-byte boundary and
// lm_head's epilogue stores 16-byte vectors (Llama-3's 128 261 = 128 256 + 5 added tokens is odd: rows at a stride of V floats are only
// 4-byte aligned and went out as sixteen scalar stores per lane and slab -- lm_head at 97 rows 183 -> 140 us).  Callers' logits buffers
// (oavsr_prefill) stay dense [B][V]: copied row by row.
static inline int64_t vstride(const oavsr_config& c) { return ((int64_t)c.llm_vocab + 15) & ~(int64_t)15; }
static int lm_head_rows(const oavsr_model* m, oavsr_workspace* ws, int rows, float* logits, hipStream_t st) {
  const oavsr_config& c = m->cfg;
  const int d = c.llm_hidden;
  GemmArgs g; g.dtype = m->dtype; g.M = rows; g.N = c.llm_vocab; g.K = d; g.W = m->lm_head; g.ldw = d;
  g.A = ws->lh; g.lda = d; g.C = logits; g.ldc = vstride(c); g.out_f32 = 1;
  // (N = the vocabulary: the column tiles alone fill the chip; decode_split gives 1)
  return decode_gemm(m, ws, g, st, ceil_div(d, 64));
}

// Where a prefill puts what the decode loop needs: the row slots slot0.. of a decode scheduler's pool instead of the workspace's own caches.
struct PrefillDest { void *kpre, *vpre; size_t layer_bytes; int max_prefix; float* logits; int* prefix_len; int slot0; };

// utt0: first utterance slot of this call (prefix-cache index, logits row); > 0 only for the chunked generate, whose chunks are
// prefilled one after another into consecutive slots and then decoded together.
static int prefill_impl(const oavsr_model* m, oavsr_workspace* ws, int task, int B, const void* embeds, const int32_t* S, float* logits_out,
                        hipStream_t st, int utt0 = 0, const PrefillDest* dest = nullptr) {
  const oavsr_config& c = m->cfg;
  OAVSR_REQUIRE(B >= 1 && utt0 >= 0 && utt0 + B <= ws->max_B, "prefill: B=%d (+%d) exceeds workspace (%d)", B, utt0, ws->max_B);
  int total = 0, maxS = 0;
  ws->pf_k = dest ? dest->kpre : ws->kpre; ws->pf_v = dest ? dest->vpre : ws->vpre;
  ws->pf_layer = dest ? dest->layer_bytes : ws->kpre_layer; ws->pf_max_prefix = dest ? dest->max_prefix : ws->max_prefix;
  const int slot_base = dest ? dest->slot0 : utt0;
  for (int b = 0; b < B; ++b) { OAVSR_REQUIRE(S[b] >= 1 && S[b] <= ws->pf_max_prefix, "prefill: S[%d]=%d exceeds prefix cache (%d)", b, S[b], ws->pf_max_prefix); total += S[b]; maxS = std::max(maxS, S[b]); }
  OAVSR_REQUIRE(total <= ws->max_S_total, "prefill: too many prompt rows");
  int* pos = ws->stage<int>(total); int* slot = ws->stage<int>(total); int* cu = ws->stage<int>(B + 1); int* last = ws->stage<int>(B);
  int* plen = ws->stage<int>(B);
  OAVSR_REQUIRE(pos && slot && cu && last && plen, "host staging exhausted");
  int r = 0; cu[0] = 0;
  for (int b = 0; b < B; ++b) { for (int i = 0; i < S[b]; ++i) { pos[r] = i; slot[r] = slot_base + b; ++r; } cu[b + 1] = r; last[b] = r - 1; plen[b] = S[b]; ws->S_host[utt0 + b] = S[b]; }
  ws->B_cur = dest ? 0 : utt0 + B;             // a pool-bound prefill leaves nothing for this workspace's own decode
  OAVSR_TRY(h2d(ws->pos, pos, sizeof(int) * total, st));
  OAVSR_TRY(h2d(ws->cpos, pos, sizeof(int) * total, st));
  OAVSR_TRY(h2d(ws->slot, slot, sizeof(int) * total, st));
  OAVSR_TRY(h2d(ws->cu_s, cu, sizeof(int) * (B + 1), st));
  OAVSR_TRY(h2d(ws->last_rows, last, sizeof(int) * B, st));
  OAVSR_TRY(h2d(dest ? dest->prefix_len + dest->slot0 : ws->prefix_len + utt0, plen, sizeof(int) * B, st));
  if (embeds != ws->lx) OAVSR_CHECK_HIP(hipMemcpyAsync(ws->lx, embeds, (size_t)total * c.llm_hidden * m->esz, hipMemcpyDeviceToDevice, st));
  // what this call's prompt rows hold (the spliced inputs_embeds, modeling_OmniAVSR.py:397-458), per utterance, before the layers overwrite them
  if (ws->digests) OAVSR_TRY(launch_digest(ws->lx, ws->cu_s, (int64_t)c.llm_hidden * m->esz / 4, B, ws->digests + (size_t)utt0 * 2, 2, 0, st));
  int* cul = ws->stage<int>(B + 1);
  OAVSR_REQUIRE(cul, "host staging exhausted");
  for (int b = 0; b <= B; ++b) cul[b] = b;
  OAVSR_TRY(h2d(ws->cu_b, cul, sizeof(int) * (B + 1), st));   // cu_b: encoder scratch, idle during prefill
  // logits of every utterance's last position only (the reference materialises all positions, Llama_LoRA.py:372-373)
  OAVSR_TRY(llm_prefill_layers(m, ws, task, total, B, maxS, ws->pos, ws->cpos, ws->last_rows, ws->cu_b, st));
  float* lg = dest ? dest->logits + (size_t)dest->slot0 * vstride(c) : ws->logits + (size_t)utt0 * vstride(c);
  OAVSR_TRY(lm_head_rows(m, ws, B, lg, st));
  if (ws->digests) OAVSR_TRY(launch_digest(lg, nullptr, c.llm_vocab, B, ws->digests + (size_t)utt0 * 2, 2, 1, st, vstride(c)));      // the first step's logits row
  if (logits_out) OAVSR_CHECK_HIP(hipMemcpy2DAsync(logits_out, (size_t)c.llm_vocab * 4, lg, (size_t)vstride(c) * 4, (size_t)c.llm_vocab * 4, (size_t)B, hipMemcpyDeviceToDevice, st));
  return 0;
}

static int decode_greedy(const oavsr_model* m, oavsr_workspace* ws, int task, int B, int max_new, int force_len, int32_t* out_ids, int32_t* out_len,
                         hipStream_t st) {
  const oavsr_config& c = m->cfg;
  OAVSR_REQUIRE(B == ws->B_cur, "decode: call oavsr_prefill for this batch first");
  OAVSR_REQUIRE(max_new >= 1 && max_new <= ws->max_new, "decode: max_new=%d exceeds workspace (%d)", max_new, ws->max_new);
  // state: pos = S-1, cpos = -1, step = 0, finished = 0, out_len = 0, out_ids = pad
  int* init = ws->stage<int>((size_t)B * 4 + 1 + (size_t)B * max_new);
  OAVSR_REQUIRE(init, "host staging exhausted");
  int* ipos = init; int* icpos = init + B; int* ifin = init + 2 * B; int* ilen = init + 3 * B; int* istep = init + 4 * B; int* iout = istep + 1;
  int maxS = 0;
  for (int b = 0; b < B; ++b) { ipos[b] = ws->S_host[b] - 1; icpos[b] = -1; ifin[b] = 0; ilen[b] = 0; maxS = std::max(maxS, ws->S_host[b]); }
  *istep = 0;
  for (int i = 0; i < B * max_new; ++i) iout[i] = c.id_pad;
  const bool graphs_on = knob("OAVSR_DECODE_GRAPH") != 0;
  // hipGraph replay needs stable addresses -> results land in workspace buffers and are copied out at the end; the eager
  // path (profiling, legacy default stream which cannot be captured) writes the caller's buffers directly.
  const bool use_graph = graphs_on && !g_prof.on && max_new > 3 && st != nullptr;
  GreedyState gs = ws->gs;
  gs.inv_freq = m->inv_freq; gs.rope_half = c.llm_head_dim / 2;
  if (!use_graph) { gs.out_ids = out_ids; gs.out_len = out_len; }
  OAVSR_TRY(h2d(gs.pos, ipos, sizeof(int) * B, st));
  OAVSR_TRY(h2d(gs.cpos, icpos, sizeof(int) * B, st));
  OAVSR_TRY(h2d(gs.finished, ifin, sizeof(int) * B, st));
  OAVSR_TRY(h2d(gs.out_len, ilen, sizeof(int) * B, st));
  OAVSR_TRY(h2d(gs.step, istep, sizeof(int), st));
  OAVSR_TRY(h2d(gs.out_ids, iout, sizeof(int) * B * max_new, st));
  // decode rows use slot = row index
  int* slot = ws->stage<int>(B);
  OAVSR_REQUIRE(slot, "host staging exhausted");
  for (int b = 0; b < B; ++b) slot[b] = b;
  OAVSR_TRY(h2d(ws->slot, slot, sizeof(int) * B, st));
  const int banned = force_len ? c.id_eos : -1;
  OAVSR_TRY(launch_argmax(ws->logits, vstride(c), B, c.llm_vocab, banned, ws->amax, ws->amax_scratch, st));
  OAVSR_TRY(launch_greedy_update(ws->amax, gs, B, max_new, c.id_eos, c.id_pad, st));
  auto one_step = [&]() -> int {
    OAVSR_TRY(launch_embed(m->dtype, m->embed, c.llm_hidden, gs.tok, ws->lx, c.llm_hidden, B, c.llm_hidden, st));
    OAVSR_TRY(llm_decode_layers(m, ws, task, B, 1, maxS + max_new, gs.pos, gs.cpos, nullptr, gs.step, st, gs.rope_cs));
    OAVSR_TRY(lm_head_rows(m, ws, B, ws->logits, st));
    OAVSR_TRY(launch_argmax(ws->logits, vstride(c), B, c.llm_vocab, banned, ws->amax, ws->amax_scratch, st));
    return launch_greedy_update(ws->amax, gs, B, max_new, c.id_eos, c.id_pad, st);
  };
  // The decode step is identical every iteration (all step-dependent values live in device memory), so it is captured
  // once per (B, task, max_new, force_len) into a hipGraph and replayed: one graph launch instead of ~110 kernel launches.
  int step = 1;
  if (use_graph) {
    const DecodeGraphKey key{B, task, max_new, banned};
    hipGraphExec_t exec = nullptr;
    for (auto& e : ws->graphs) if (e.first == key) exec = e.second;
    if (!exec) {
      OAVSR_TRY(one_step());            // eager once: sets the per-kernel attributes outside of the capture
      step = 2;
      hipGraph_t graph = nullptr;
      OAVSR_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
      int rc = one_step();
      hipError_t ce = hipStreamEndCapture(st, &graph);
      OAVSR_REQUIRE(rc == 0 && ce == hipSuccess && graph, "decode: hipGraph capture failed (%d / %d)", rc, (int)ce);
      OAVSR_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      hipGraphDestroy(graph);
      ws->graphs.push_back({key, exec});
    }
    for (; step < max_new; ++step) OAVSR_CHECK_HIP(hipGraphLaunch(exec, st));
  }
  for (; step < max_new; ++step) OAVSR_TRY(one_step());
  if (use_graph) {
    OAVSR_CHECK_HIP(hipMemcpyAsync(out_ids, gs.out_ids, sizeof(int) * B * max_new, hipMemcpyDeviceToDevice, st));
    OAVSR_CHECK_HIP(hipMemcpyAsync(out_len, gs.out_len, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
  }
  return 0;
}

// HF beam search (modeling_OmniAVSR.py:313-322 with num_beams > 1): rows = B*K share the utterance's prefix KV, tails are
// addressed through the ancestry table maintained by beam_step_kernel.  No host synchronisation; all max_new steps are
// issued (finished utterances keep padding, exactly as BeamSearchScorer.process does for done batches).
static int decode_beam(const oavsr_model* m, oavsr_workspace* ws, int task, int B, int K, int max_new, int32_t* out_ids, int32_t* out_len,
                       hipStream_t st) {
  const oavsr_config& c = m->cfg;
  OAVSR_REQUIRE(B == ws->B_cur, "decode: call oavsr_prefill for this batch first");
  OAVSR_REQUIRE(max_new >= 1 && max_new <= ws->max_new, "decode: max_new=%d exceeds workspace (%d)", max_new, ws->max_new);
  OAVSR_REQUIRE(K <= ws->max_beams && K <= 32, "decode: num_beams=%d exceeds workspace (%d) / kernel limit 32", K, ws->max_beams);
  OAVSR_REQUIRE(max_new == ws->max_new, "decode: beam search needs max_new == workspace max_new (tail-cache stride)");
  const int R = B * K;
  int maxS = 0;
  for (int b = 0; b < B; ++b) maxS = std::max(maxS, ws->S_host[b]);
  BeamArgs a;
  a.B = B; a.K = K; a.V = c.llm_vocab; a.max_new = max_new; a.eos = c.id_eos; a.pad = c.id_pad;
  a.step = ws->gs.step; a.beam_scores = ws->beam_scores; a.next_tok = ws->gs.tok; a.pos = ws->gs.pos; a.cpos = ws->gs.cpos;
  a.rope_cs = ws->gs.rope_cs; a.inv_freq = m->inv_freq; a.rope_half = c.llm_head_dim / 2;
  a.prefix_len = ws->prefix_len; a.cand_val = ws->cand_val; a.cand_tok = ws->cand_tok; a.done = ws->gs.finished; a.hyp_n = ws->hyp_n;
  a.hyp_score = ws->hyp_score; a.hyp_len = ws->hyp_len; a.hyp_tok = ws->hyp_tok; a.out_ids = out_ids; a.out_len = out_len;
  OAVSR_CHECK_HIP(hipMemsetAsync(ws->gs.step, 0, sizeof(int), st));
  OAVSR_CHECK_HIP(hipMemsetAsync(ws->gs.finished, 0, sizeof(int) * B, st));
  OAVSR_CHECK_HIP(hipMemsetAsync(ws->hyp_n, 0, sizeof(int) * B, st));
  int* seq[2] = {ws->seq[0], ws->seq[1]};
  int* anc[2] = {ws->anc[0], ws->anc[1]};
  // step 0: candidates from the prefill logits (one row per utterance; beam_scores = [0, -1e9, ...] <=> only beam 0 counts)
  { ProfScope ps("beam_search", 0, (double)B * c.llm_vocab * 16, st);
    OAVSR_TRY(launch_beam_topk(ws->logits, vstride(c), B, c.llm_vocab, K, nullptr, ws->cand_val, ws->cand_tok, ws->topk_flag, st));
    OAVSR_TRY(launch_beam_step(a, 1, seq[0], anc[0], seq[1], anc[1], st)); }
  // one step reads ping-pong side `cur` of the sequence / ancestry tables and writes the other one
  auto one_step = [&](int cur) -> int {
    OAVSR_TRY(launch_embed(m->dtype, m->embed, c.llm_hidden, ws->gs.tok, ws->lx, c.llm_hidden, R, c.llm_hidden, st));
    OAVSR_TRY(llm_decode_layers(m, ws, task, R, K, maxS + max_new, ws->gs.pos, ws->gs.cpos, anc[cur], ws->gs.step, st, ws->gs.rope_cs));
    OAVSR_TRY(lm_head_rows(m, ws, R, ws->logits, st));
    ProfScope ps("beam_search", 0, (double)R * c.llm_vocab * 16, st);
    OAVSR_TRY(launch_beam_topk(ws->logits, vstride(c), R, c.llm_vocab, K, ws->beam_scores, ws->cand_val, ws->cand_tok, ws->topk_flag, st));
    return launch_beam_step(a, 0, seq[cur], anc[cur], seq[cur ^ 1], anc[cur ^ 1], st);
  };
  // Every step-dependent value lives in device memory, so a PAIR of steps (side 0 -> 1 -> 0) is one hipGraph, captured once per
  // (B, K, task, max_new) and replayed: one graph launch instead of ~450 kernel launches.  Step 1 always runs eagerly (it also
  // sets the kernels' attributes before the first capture) and leaves the tables on side 0.
  const bool graphs_on = knob("OAVSR_DECODE_GRAPH") != 0;
  const bool use_graph = graphs_on && !g_prof.on && max_new > 4 && st != nullptr;
  int t = 1, cur = 1;
  if (t < max_new) { OAVSR_TRY(one_step(cur)); cur ^= 1; ++t; }
  if (use_graph && t + 1 < max_new) {
    const DecodeGraphKey key{B, task, max_new, -1000 - K};
    hipGraphExec_t exec = nullptr;
    for (auto& e : ws->graphs) if (e.first == key) exec = e.second;
    if (!exec) {
      hipGraph_t graph = nullptr;
      OAVSR_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
      int rc = one_step(0);
      if (rc == 0) rc = one_step(1);
      hipError_t ce = hipStreamEndCapture(st, &graph);
      OAVSR_REQUIRE(rc == 0 && ce == hipSuccess && graph, "decode: hipGraph capture of the beam step failed (%d / %d)", rc, (int)ce);
      OAVSR_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      hipGraphDestroy(graph);
      ws->graphs.push_back({key, exec});
    }
    for (; t + 1 < max_new; t += 2) OAVSR_CHECK_HIP(hipGraphLaunch(exec, st));
  }
  for (; t < max_new; ++t) { OAVSR_TRY(one_step(cur)); cur ^= 1; }
  return launch_beam_finalize(a, seq[cur], st);
}

}  // namespace oavsr

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int oavsr_abi_version(void) { return OAVSR_ABI_VERSION; }

extern char** environ;
int oavsr_env_report(char* buf, int32_t nbytes) {
  // every OAVSR_* variable of the process: "NAME=value" (the library's own switches) or "NAME=value?" (a name the library does not read --
  // the Python host's, or a switch that no longer exists), ';'-separated
  std::string out;
  int n = 0;
  for (char** e = environ; e && *e; ++e) {
    if (strncmp(*e, "OAVSR_", 6) != 0) continue;
    const char* eq = strchr(*e, '=');
    const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
    bool known = false;
    for (const KnobDef& k : kKnobs) known = known || name == k.name;
    if (!out.empty()) out += ';';
    out += *e;
    if (!known) out += '?';
    ++n;
  }
  if (buf && nbytes > 0) { strncpy(buf, out.c_str(), (size_t)nbytes - 1); buf[nbytes - 1] = 0; }
  return n;
}
const char* oavsr_last_error(void) { return g_err; }

int oavsr_model_create(const oavsr_config* cfg, const void* blob_dev, uint64_t blob_bytes, const oavsr_tensor* tensors, int32_t n_tensors,
                       oavsr_model** out) {
  OAVSR_REQUIRE(cfg && blob_dev && tensors && out, "model_create: null argument");
  OAVSR_REQUIRE(cfg->dtype == DT_F32 || cfg->dtype == DT_BF16, "model_create: bad dtype");
  OAVSR_REQUIRE(cfg->llm_head_dim == 64 || cfg->llm_head_dim == 128, "model_create: llm head_dim %d unsupported", cfg->llm_head_dim);
  OAVSR_REQUIRE(cfg->llm_heads % cfg->llm_kv_heads == 0 && cfg->llm_heads / cfg->llm_kv_heads <= 8, "model_create: GQA group must be <= 8");
  oavsr_model* m = new oavsr_model();
  m->cfg = *cfg; m->blob = (const char*)blob_dev; m->blob_bytes = blob_bytes; m->dtype = cfg->dtype; m->esz = cfg->dtype == DT_F32 ? 4 : 2;
  for (int i = 0; i < n_tensors; ++i) {
    if (tensors[i].offset + tensors[i].nbytes > blob_bytes || (tensors[i].offset & 255)) {
      set_last_error("model_create: tensor '%s' out of bounds or misaligned", tensors[i].name); delete m; return -1;
    }
    m->t[tensors[i].name] = {tensors[i].offset, tensors[i].nbytes};
  }
  if (resolve_model(m) != 0) { delete m; return -1; }
  // constants for the log-mel front-end
  std::vector<double> tw(800);
  for (int k = 0; k < 400; ++k) { tw[2 * k] = cos(2.0 * M_PI * k / 400.0); tw[2 * k + 1] = sin(2.0 * M_PI * k / 400.0); }
  std::vector<float> fb; build_melfb(fb);
  { hipError_t e1 = hipMalloc(&m->twiddle, 800 * sizeof(double)), e2 = e1 == hipSuccess ? hipMalloc(&m->melfb, fb.size() * sizeof(float)) : e1;
    if (e2 != hipSuccess) { set_last_error("model_create: hipMalloc of the log-mel constants failed: %s", hipGetErrorString(e2)); delete m; return -2; } }
  hipMemcpy(m->twiddle, tw.data(), 800 * sizeof(double), hipMemcpyHostToDevice);
  hipMemcpy(m->melfb, fb.data(), fb.size() * sizeof(float), hipMemcpyHostToDevice);
  if (m->has_video) {
    if (hipError_t e1 = hipMalloc(&m->stem_wp, 64 * 288 * m->esz); e1 != hipSuccess) { set_last_error("model_create: hipMalloc of the stem weights failed: %s", hipGetErrorString(e1)); oavsr_model_destroy(m); return -2; }
    if (launch_stem_repack(m->dtype, m->stem.w, m->stem_wp, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) {
      set_last_error("model_create: stem repack failed"); oavsr_model_destroy(m); return -2;
    }
  }
  if (m->has_video && enc_cat_ok(m->cfg.av_dim, m->cfg.av_lora_k2, m->esz)) {
    // [qkv.w | lora_up.w] per video layer: the LoRA term rides in the QKV GEMM's ONE K range (run_encoder_layers)
    const size_t D = m->cfg.av_dim, k2 = m->cfg.av_lora_k2, e = m->esz, per = 3 * D * (D + k2) * e;
    if (hipError_t e1 = hipMalloc(&m->enc_cat, per * m->vlayers.size()); e1 != hipSuccess) { set_last_error("model_create: hipMalloc of the concatenated QKV + LoRA weights (%zu bytes) failed: %s", per * m->vlayers.size(), hipGetErrorString(e1)); oavsr_model_destroy(m); return -2; }
    for (size_t i = 0; i < m->vlayers.size(); ++i) {
      char* dst = (char*)m->enc_cat + i * per;
      if (hipMemcpy2D(dst, (D + k2) * e, m->vlayers[i].qkvw, D * e, D * e, 3 * D, hipMemcpyDeviceToDevice) != hipSuccess ||
          hipMemcpy2D(dst + D * e, (D + k2) * e, m->vlayers[i].lora_up, k2 * e, k2 * e, 3 * D, hipMemcpyDeviceToDevice) != hipSuccess) {
        set_last_error("model_create: QKV / LoRA concatenation failed"); oavsr_model_destroy(m); return -2;
      }
      m->vlayers[i].qkvw_cat = dst;
    }
  }
  // (as in workspace_create: what was put on the null stream above must be done before a caller's non-blocking stream reads it)
  if (hipDeviceSynchronize() != hipSuccess) { set_last_error("model_create: device synchronisation failed"); oavsr_model_destroy(m); return -2; }
  *out = m;
  return 0;
}

void oavsr_model_destroy(oavsr_model* m) {
  if (!m) return;
  if (m->twiddle) hipFree(m->twiddle);
  if (m->melfb) hipFree(m->melfb);
  if (m->stem_wp) hipFree(m->stem_wp);
  if (m->enc_cat) hipFree(m->enc_cat);
  if (m->enc_done) hipEventDestroy(m->enc_done);
  delete m;
}

int oavsr_set_launch_objective(int32_t alone) { g_alone.store(alone ? 1 : 0, std::memory_order_relaxed); return 0; }
int oavsr_get_launch_objective(void) { return g_alone.load(std::memory_order_relaxed); }
int oavsr_model_serialize_encoders(oavsr_model* m, int32_t on) {
  OAVSR_REQUIRE(m, "serialize_encoders: null model");
  if (on && !m->enc_done) OAVSR_CHECK_HIP(hipEventCreateWithFlags(&m->enc_done, hipEventDisableTiming));
  m->serialize_enc = on != 0;
  m->enc_recorded = false;
  return 0;
}

int oavsr_workspace_create(const oavsr_model* m, int32_t max_B, int32_t max_Tv, int32_t max_Ts, int32_t max_beams, int32_t max_new,
                           oavsr_workspace** out) {
  OAVSR_REQUIRE(m && out, "workspace_create: null argument");
  OAVSR_REQUIRE(max_B >= 1 && max_B <= 1024 && max_new >= 1 && max_beams >= 1, "workspace_create: bad sizes");
  const oavsr_config& c = m->cfg;
  oavsr_workspace* ws = new oavsr_workspace();
  ws->m = m; ws->max_B = max_B; ws->max_Tv = max_Tv; ws->max_Ts = max_Ts; ws->max_beams = max_beams; ws->max_new = max_new;
  const size_t e = m->esz;
  const size_t F = m->has_video ? (size_t)max_B * max_Tv : 0;
  const size_t TA = m->has_audio ? (size_t)max_B * 1500 : 0;
  const size_t NT = std::max(F, TA);
  const int Dv = m->has_video ? c.av_dim : 0, Da = m->has_audio ? c.wh_dim : 0;
  const size_t De = std::max(Dv, Da);
  const size_t Fe = std::max(m->has_video ? c.av_ffn : 0, m->has_audio ? c.wh_ffn : 0);
  const int d = c.llm_hidden, H = c.llm_heads, KV = c.llm_kv_heads, hd = c.llm_head_dim, I = c.llm_mlp;
  const size_t nqkv = (size_t)(H + 2 * KV) * hd;
  // longest prompt: BOS + audio(1500/1 + 2) + video(max_Tv + 2) + prompt
  int maxP = std::max(c.prompt_len[0], std::max(c.prompt_len[1], c.prompt_len[2]));
  int min_ra = 1 << 30, min_rv = 1 << 30;
  for (int i = 0; i < c.n_rates_audio; ++i) min_ra = std::min(min_ra, c.rates_audio[i]);
  for (int i = 0; i < c.n_rates_video; ++i) min_rv = std::min(min_rv, c.rates_video[i]);
  int max_keep = std::min(1500, std::max(25, (int)((double)max_Ts / 16000.0 * 50.0) + 1));
  int max_na = m->has_audio ? max_keep / std::max(1, min_ra == (1 << 30) ? 1 : min_ra) : 0;
  int max_nv = m->has_video ? max_Tv / std::max(1, min_rv == (1 << 30) ? 1 : min_rv) : 0;
  ws->max_prefix = 1 + (m->has_audio ? max_na + 2 : 0) + (m->has_video ? max_nv + 2 : 0) + maxP;
  ws->max_prefix = (ws->max_prefix + 63) / 64 * 64;
  ws->max_S_total = ws->max_prefix * max_B;
  const size_t ST = (size_t)ws->max_S_total;
  const size_t rows_dec = (size_t)max_B * max_beams;
  const size_t Rl = std::max(ST, rows_dec);
  struct Req { void** p; size_t bytes; };
  std::vector<Req> reqs;
  auto want = [&](void** p, size_t bytes) { reqs.push_back({p, bytes}); };
#define WANT(field, bytes) want((void**)&ws->field, (bytes))
  const int stackmul_a = c.compression_stack ? 1 : 1;
  (void)stackmul_a;
  // One scratch set sized for either encoder; a model with both gets a second one of the video encoder's size, so that the two encoders of
  // one call can be in flight together (encode_prefill).  F x (6 Dv + ffn) elements: 65 MB at 8 x 400 frames.
  const bool two_sets = m->has_video && m->has_audio;
  WANT(esa.x, NT * De * e); WANT(esa.h, NT * De * e); WANT(esa.qkv, NT * 3 * De * e); WANT(esa.attn, NT * De * e);
  WANT(esa.ffn, NT * Fe * e); WANT(esa.lora, NT * (size_t)std::max(64, c.av_lora_k2) * e);
  WANT(esa.rowinfo, std::max(F, (size_t)max_B * 3000) * sizeof(int4)); WANT(esa.cu, (max_B + 1) * sizeof(int)); WANT(esa.seg, max_B * sizeof(int4));
  if (two_sets) {
    WANT(esv.x, F * Dv * e); WANT(esv.h, F * Dv * e); WANT(esv.qkv, F * 3 * Dv * e); WANT(esv.attn, F * Dv * e);
    WANT(esv.ffn, F * (size_t)c.av_ffn * e); WANT(esv.lora, F * (size_t)std::max(64, c.av_lora_k2) * e);
    WANT(esv.rowinfo, F * sizeof(int4)); WANT(esv.cu, (max_B + 1) * sizeof(int)); WANT(esv.seg, max_B * sizeof(int4));
  }
  if (m->has_video && m->enc_cat) WANT(esv.hcat, F * (size_t)(Dv + c.av_lora_k2) * e);
  WANT(enc_out_v, F * Dv * e); WANT(enc_out_a, TA * Da * e);
  if (m->has_video) {
    WANT(pool_out, F * 484 * 64 * e);
    for (int i = 0; i < 2; ++i) WANT(trunk[i], F * 484 * 64 * e);
    WANT(feat512, F * 512 * e); WANT(featproj, F * Dv * e); WANT(ln2d, F * 2 * Dv * e);
    WANT(frameinfo, F * sizeof(int2));
    WANT(comp_v, F * Dv * e); WANT(tok_v, F * (size_t)d * e);
  }
  WANT(rowinfo2, ((size_t)max_B * 1500 + 16) * sizeof(int4)); WANT(cu_b, (max_B + 1) * sizeof(int));
  if (m->has_audio) {
    WANT(mel, (size_t)max_B * 3000 * 80 * e); WANT(mel_ws, (size_t)max_B * 3000 * 80 * 4); WANT(max_ws, max_B * 4); WANT(lens_dev, max_B * 4);
    WANT(conv1_out, (size_t)max_B * 3000 * Da * e); WANT(pos_rep, TA * Da * e);
    WANT(comp_a, TA * Da * e); WANT(tok_a, TA * (size_t)d * e);
  }
  WANT(proj_h, std::max(F, TA) * (size_t)c.proj_inter * e);
  WANT(embeds, ST * d * e); WANT(maps, ST * 4 * sizeof(int));
  WANT(lx, Rl * d * e); WANT(lh, Rl * d * e); WANT(lqkv, Rl * nqkv * e); WANT(lattn, Rl * (size_t)H * hd * e);
  WANT(lgu, Rl * 2 * I * e); WANT(lact, Rl * (size_t)I * e); WANT(llora, Rl * (size_t)std::max(64, c.lora_k2) * e);
  WANT(lastx, rows_dec * d * e); WANT(logits, rows_dec * (size_t)vstride(c) * 4);
  ws->kpre_layer = (size_t)max_B * KV * ws->max_prefix * hd * e;
  ws->ktail_layer = rows_dec * KV * (size_t)max_new * hd * e;
  WANT(kpre, ws->kpre_layer * c.llm_layers); WANT(vpre, ws->kpre_layer * c.llm_layers);
  WANT(ktail, ws->ktail_layer * c.llm_layers); WANT(vtail, ws->ktail_layer * c.llm_layers);
  WANT(pos, Rl * 4); WANT(slot, Rl * 4); WANT(cpos, Rl * 4); WANT(cu_s, (max_B + 1) * 4); WANT(prefix_len, max_B * 4); WANT(last_rows, max_B * 4);
  WANT(amax, rows_dec * 4); WANT(amax_scratch, rows_dec * 64 * 8);
  WANT(chain_ctr, kChainInts * 4);
  WANT(gsync, kSyncInts * 4); WANT(rowsq, rowsq_chunk(d) * ceil_div64((int64_t)Rl, 128) * 4);
  WANT(gs.tok, rows_dec * 4); WANT(gs.finished, rows_dec * 4); WANT(gs.out_ids, rows_dec * (size_t)max_new * 4); WANT(gs.out_len, rows_dec * 4);
  WANT(gs.step, 256); WANT(gs.pos, rows_dec * 4); WANT(gs.cpos, rows_dec * 4); WANT(gs.rope_cs, rows_dec * (size_t)(hd / 2) * sizeof(float2));
  {   // split-K partial slabs of the decode-shaped GEMMs: ks(N, K) x rows x N fp32 per call site (decode_split; o_proj / down_proj split >= 2)
    const size_t rows_sl = 128;      // every decode-shaped GEMM launch covers <= 128 rows (decode_gemm)
    const int es = (int)e, qkv = (int)nqkv, nq_ = H * hd;
    const int sites[5][3] = {{c.lora_k2, d, 1}, {qkv, d + c.lora_k2, 1}, {d, nq_, 2}, {2 * I, d, 1}, {d, I, 2}};
    size_t pb = 0;
    for (auto& s2 : sites) if (s2[0] > 0) pb = std::max(pb, (size_t)decode_split(s2[0], s2[1], es, s2[2]) * rows_sl * s2[0] * 4);
    ws->partial_bytes = pb;
    WANT(partial, std::max<size_t>(pb, 256));
  }
  if (max_beams > 1) {
    WANT(beam_scores, rows_dec * 4); WANT(cand_val, rows_dec * 2 * max_beams * 4); WANT(cand_tok, rows_dec * 2 * max_beams * 4); WANT(topk_flag, rows_dec * 4);
    WANT(hyp_n, max_B * 4); WANT(hyp_len, rows_dec * 4); WANT(hyp_tok, rows_dec * (size_t)max_new * 4); WANT(hyp_score, rows_dec * 8);
    for (int i = 0; i < 2; ++i) { WANT(seq[i], rows_dec * (size_t)max_new * 4); WANT(anc[i], rows_dec * (size_t)max_new * 4); }
  }
#undef WANT
  size_t total = 0;
  for (auto& r : reqs) total = align_up(total, 256) + std::max<size_t>(r.bytes, 256);
  total += 4096;
  if (hipMalloc((void**)&ws->arena, total) != hipSuccess) { set_last_error("workspace_create: hipMalloc(%zu bytes) failed", total); delete ws; return -2; }
  ws->arena_bytes = total;
  // Start from zeros rather than from whatever the recycled device memory holds (the kernels guard their ragged edges themselves; this
  // keeps a missed guard from turning into run-to-run differences).
  if (hipMemset(ws->arena, 0, total) != hipSuccess) { set_last_error("workspace_create: hipMemset failed"); hipFree(ws->arena); delete ws; return -2; }
  for (auto& r : reqs) { *r.p = ws->take(std::max<size_t>(r.bytes, 256)); if (!*r.p) { set_last_error("workspace arena overflow"); hipFree(ws->arena); delete ws; return -1; } }
  ws->two_sets = two_sets;
  if (!two_sets) { void* hcat = ws->esv.hcat; ws->esv = ws->esa; ws->esv.hcat = hcat; }      // one encoder: one set
  // decode uses gs.pos / gs.cpos as the row position arrays of the layer stack
  ws->host_bytes = (size_t)(F * sizeof(int2) + ST * 8 * sizeof(int) + rows_dec * (size_t)(max_new + 16) * sizeof(int) + 1 * 1024 * 1024);
  if (hipHostMalloc((void**)&ws->host_base, 2 * ws->host_bytes) != hipSuccess) { set_last_error("workspace_create: hipHostMalloc failed"); hipFree(ws->arena); delete ws; return -2; }
  if (m->has_audio) {   // replicate embed_positions for every utterance slot (residual operand of conv2's epilogue)
    for (int b = 0; b < max_B; ++b)
      hipMemcpy((char*)ws->pos_rep + (size_t)b * 1500 * Da * e, m->apos, (size_t)1500 * Da * e, hipMemcpyDeviceToDevice);
  }
  ws->host = ws->host_base;
  hipEventCreateWithFlags(&ws->host_ev[0], hipEventDisableTiming);
  hipEventCreateWithFlags(&ws->host_ev[1], hipEventDisableTiming);
  // The device-side initialisation above runs on the null stream (a device-to-device hipMemcpy may return before it has executed), and the
  // callers' non-blocking streams do not wait for that stream: a workspace created while another call of the same model was in flight was
  // used before its position table had been written (wrong tokens, found by tests/test_stages_gpu.py).  Creation is rare: wait here.
  if (hipDeviceSynchronize() != hipSuccess) { set_last_error("workspace_create: device synchronisation failed"); oavsr_workspace_destroy(ws); return -2; }
  *out = ws;
  return 0;
}

void oavsr_workspace_destroy(oavsr_workspace* ws) {
  if (!ws) return;
  for (auto& e : ws->graphs) hipGraphExecDestroy(e.second);
  if (ws->own_stream) { hipStreamSynchronize(ws->own_stream); hipStreamDestroy(ws->own_stream); }
  if (ws->enc2) { hipStreamSynchronize(ws->enc2); hipStreamDestroy(ws->enc2); }
  if (ws->enc2_fork) hipEventDestroy(ws->enc2_fork);
  if (ws->enc2_join) hipEventDestroy(ws->enc2_join);
  if (ws->own_in) hipEventDestroy(ws->own_in);
  if (ws->own_out) hipEventDestroy(ws->own_out);
  if (ws->arena) hipFree(ws->arena);
  if (ws->host_base) hipHostFree(ws->host_base);
  for (int i = 0; i < 2; ++i) if (ws->host_ev[i]) hipEventDestroy(ws->host_ev[i]);
  delete ws;
}
uint64_t oavsr_workspace_bytes(const oavsr_workspace* ws) { return ws ? ws->arena_bytes : 0; }

int oavsr_workspace_set_video_format(oavsr_workspace* ws, int32_t channels, int32_t height, int32_t width) {
  OAVSR_REQUIRE(ws, "workspace_set_video_format: null workspace");
  if (channels == 0) { ws->vsrc = StemSrc(); return 0; }
  OAVSR_REQUIRE((channels == 1 || channels == 3) && height >= 88 && width >= 88 && height <= 4096 && width <= 4096,
                "workspace_set_video_format: uint8 frames must be [1 or 3][H >= 88][W >= 88], got [%d][%d][%d]", channels, height, width);
  ws->vsrc = StemSrc::u8(channels, height, width);
  return 0;
}

int oavsr_workspace_set_digests(oavsr_workspace* ws, uint64_t* digests_dev) {
  OAVSR_REQUIRE(ws, "workspace_set_digests: null workspace");
  ws->digests = digests_dev;
  return 0;
}

int oavsr_encode_video(const oavsr_model* m, oavsr_workspace* ws, const void* video_dev, const int32_t* Tv_host, int32_t B, int32_t rate,
                       void* out_dev, void* enc_dev, oavsr_stream stream) {
  OAVSR_REQUIRE(m && ws && video_dev && Tv_host, "encode_video: null argument");
  CallScope cs(ws, (hipStream_t)stream);
  return encode_video_impl(m, ws, video_dev, Tv_host, B, rate, out_dev, enc_dev, nullptr, (hipStream_t)stream);
}

int oavsr_encode_audio(const oavsr_model* m, oavsr_workspace* ws, const float* wav_dev, int64_t ld, const int32_t* Ts_host,
                       const int32_t* keep_host, int32_t B, int32_t rate, void* out_dev, void* enc_dev, void* mel_dev, oavsr_stream stream) {
  OAVSR_REQUIRE(m && ws && wav_dev && Ts_host && keep_host, "encode_audio: null argument");
  CallScope cs(ws, (hipStream_t)stream);
  return encode_audio_impl(m, ws, wav_dev, ld, Ts_host, keep_host, B, rate, out_dev, enc_dev, mel_dev, nullptr, (hipStream_t)stream);
}

int oavsr_project(const oavsr_model* m, oavsr_workspace* ws, int32_t which, int32_t rate, const void* feats_dev, int32_t rows, void* out_dev,
                  oavsr_stream stream) {
  OAVSR_REQUIRE(m && ws && feats_dev && out_dev, "project: null argument");
  return project_impl(m, ws, which, rate, feats_dev, rows, out_dev, (hipStream_t)stream);
}

int oavsr_build_prompt(const oavsr_model* m, oavsr_workspace* ws, int32_t task, int32_t B, const int32_t* na_host, const void* audio_tok_dev,
                       const int32_t* nv_host, const void* video_tok_dev, void* embeds_dev, int32_t* S_host_out, oavsr_stream stream) {
  OAVSR_REQUIRE(m && ws && embeds_dev && S_host_out, "build_prompt: null argument");
  CallScope cs(ws, (hipStream_t)stream);
  return build_prompt_impl(m, ws, task, B, na_host, audio_tok_dev, nv_host, video_tok_dev, embeds_dev, S_host_out, (hipStream_t)stream);
}

int oavsr_prefill(const oavsr_model* m, oavsr_workspace* ws, int32_t task, int32_t B, const void* embeds_dev, const int32_t* S_host,
                  float* logits_dev, oavsr_stream stream) {
  OAVSR_REQUIRE(m && ws && embeds_dev && S_host, "prefill: null argument");
  CallScope cs(ws, (hipStream_t)stream);
  return prefill_impl(m, ws, task, B, embeds_dev, S_host, logits_dev, (hipStream_t)stream);
}

int oavsr_decode(const oavsr_model* m, oavsr_workspace* ws, int32_t task, int32_t B, int32_t beams, int32_t max_new, int32_t force_len,
                 int32_t* out_ids_dev, int32_t* out_len_dev, oavsr_stream stream) {
  OAVSR_REQUIRE(m && ws && out_ids_dev && out_len_dev, "decode: null argument");
  OAVSR_REQUIRE(beams >= 1, "decode: num_beams=%d", beams);
  CallScope cs(ws, (hipStream_t)stream);
  if (beams > 1) {
    OAVSR_REQUIRE(!force_len, "decode: force_len is a greedy-only benchmark switch");
    return decode_beam(m, ws, task, B, beams, max_new, out_ids_dev, out_len_dev, (hipStream_t)stream);
  }
  return decode_greedy(m, ws, task, B, max_new, force_len, out_ids_dev, out_len_dev, (hipStream_t)stream);
}

// Everything of a generate() call up to and including the prefill: encoders + projector + prompt + prefill per chunk of chunk_B
// utterances (the launch shapes of a batch-of-chunk_B request); each chunk's prefix KV / last-token logits land in its own slots of the
// workspace -- or (dest) in the row slots of a decode scheduler's pool.  S_out (optional): prompt rows per utterance.
static int encode_prefill(const oavsr_model* m, oavsr_workspace* ws, int32_t task, const void* video_dev, const int32_t* Tv_host, const float* wav_dev,
                          int64_t wav_ld, const int32_t* Ts_host, const int32_t* keep_host, int32_t B, int32_t chunk_B, int32_t rate_audio,
                          int32_t rate_video, hipStream_t st, const PrefillDest* dest, int* S_out) {
  // Concurrent generate() calls on different streams: the MFMA-bound encoder + prefill phases gain nothing from sharing
  // the chip with each other, so they are chained one after another; only the latency-bound decode loop of one call
  // overlaps the encoders of the next.
  if (m->serialize_enc && m->enc_recorded && m->enc_last != st) OAVSR_CHECK_HIP(hipStreamWaitEvent(st, m->enc_done, 0));
  const bool use_a = task == OAVSR_TASK_AUDIO || task == OAVSR_TASK_AUDIOVISUAL;
  const bool use_v = task == OAVSR_TASK_VIDEO || task == OAVSR_TASK_AUDIOVISUAL;
  if (use_a) OAVSR_REQUIRE(wav_dev && Ts_host && keep_host, "generate: audio inputs missing");
  if (use_v) OAVSR_REQUIRE(video_dev && Tv_host, "generate: video inputs missing");
  int64_t frames_before = 0;
  for (int b0 = 0; b0 < B; b0 += chunk_B) {
    const int Bc = std::min<int>(chunk_B, B - b0);
    std::vector<int> na(Bc), nv(Bc);
    int tot_a = 0, tot_v = 0;
    // One call at a time (the latency objective): neither encoder fills the chip at a request-sized batch (Whisper at 1500 rows: 60
    // workgroups per GEMM), so the video encoder of an audio-visual call runs on a second stream beside the audio encoder, each on its
    // own scratch set; the projectors follow on the call's stream.  Same launches, same bits.  Calls that overlap each other (the
    // pipelined job) already fill the chip with the next call's encoders and keep the one-stream order.
    const bool side = use_a && use_v && ws->two_sets && launch_alone() && !g_prof.on && knob("OAVSR_ENC_SIDE_BY_SIDE");
    hipStream_t sv = st;
    if (side && !ws->enc2) {
      OAVSR_CHECK_HIP(hipStreamCreateWithFlags(&ws->enc2, hipStreamNonBlocking));
      OAVSR_CHECK_HIP(hipEventCreateWithFlags(&ws->enc2_fork, hipEventDisableTiming));
      OAVSR_CHECK_HIP(hipEventCreateWithFlags(&ws->enc2_join, hipEventDisableTiming));
    }
    if (side) {
      OAVSR_CHECK_HIP(hipEventRecord(ws->enc2_fork, st));
      OAVSR_CHECK_HIP(hipStreamWaitEvent(ws->enc2, ws->enc2_fork, 0));
      sv = ws->enc2;
    }
    if (use_v) {
      const int64_t frame_bytes = ws->vsrc.C ? (int64_t)ws->vsrc.C * ws->vsrc.H * ws->vsrc.W : (int64_t)7744 * 4;
      const int rc = encode_video_impl(m, ws, (const char*)video_dev + frames_before * frame_bytes, Tv_host + b0, Bc, rate_video, nullptr, nullptr, nv.data(), sv);
      if (side) {      // joined on every path: the caller's stream never runs ahead of work on the second stream
        hipEventRecord(ws->enc2_join, ws->enc2);
        if (rc) hipStreamWaitEvent(st, ws->enc2_join, 0);
      }
      if (rc) return rc;
      for (int b = 0; b < Bc; ++b) { tot_v += nv[b]; frames_before += Tv_host[b0 + b]; }
    }
    if (use_a) {
      const int rc = encode_audio_impl(m, ws, wav_dev + (int64_t)b0 * wav_ld, wav_ld, Ts_host + b0, keep_host + b0, Bc, rate_audio, nullptr, nullptr, nullptr,
                                       na.data(), st);
      if (rc) { if (side) hipStreamWaitEvent(st, ws->enc2_join, 0); return rc; }
      for (int b = 0; b < Bc; ++b) tot_a += na[b];
      OAVSR_TRY(project_impl(m, ws, 0, rate_audio, ws->comp_a, tot_a, ws->tok_a, st));
    }
    if (side) OAVSR_CHECK_HIP(hipStreamWaitEvent(st, ws->enc2_join, 0));
    if (use_v) OAVSR_TRY(project_impl(m, ws, 1, rate_video, ws->comp_v, tot_v, ws->tok_v, st));
    std::vector<int> S(Bc);
    OAVSR_TRY(build_prompt_impl(m, ws, task, Bc, use_a ? na.data() : nullptr, ws->tok_a, use_v ? nv.data() : nullptr, ws->tok_v, ws->lx, S.data(), st));
    PrefillDest d;
    if (dest) { d = *dest; d.slot0 += b0; }
    OAVSR_TRY(prefill_impl(m, ws, task, Bc, ws->lx, S.data(), nullptr, st, b0, dest ? &d : nullptr));
    if (S_out) for (int b = 0; b < Bc; ++b) S_out[b0 + b] = S[b];
  }
  if (m->serialize_enc) {
    OAVSR_CHECK_HIP(hipEventRecord(m->enc_done, st));
    m->enc_last = st; m->enc_recorded = true;
  }
  return 0;
}

static int generate_impl(const oavsr_model* m, oavsr_workspace* ws, int32_t task, const void* video_dev, const int32_t* Tv_host, const float* wav_dev,
                         int64_t wav_ld, const int32_t* Ts_host, const int32_t* keep_host, int32_t B, int32_t chunk_B, int32_t rate_audio,
                         int32_t rate_video, int32_t beams, int32_t max_new, int32_t force_len, int32_t* out_ids_dev, int32_t* out_len_dev,
                         hipStream_t st) {
  OAVSR_REQUIRE(m && ws && out_ids_dev && out_len_dev, "generate: null argument");
  OAVSR_REQUIRE(beams >= 1 && !(beams > 1 && force_len), "generate: bad num_beams / force_len");
  OAVSR_REQUIRE(B >= 1 && B <= ws->max_B, "generate: B=%d exceeds workspace (%d)", B, ws->max_B);
  OAVSR_REQUIRE(chunk_B >= 1, "generate: chunk_B must be >= 1");
  // The legacy default stream (what a caller who swaps the class and keeps the reference's one-call-at-a-time loop passes: torch's
  // current stream is the null stream) cannot be captured, so its decode steps would be ~100 eager launches each.  The call then runs on a
  // stream of the workspace's own, ordered behind everything queued on the null stream so far and ahead of everything queued after it --
  // for the caller the same stream semantics, with the hipGraph-replayed decode loop (sequential evaluation: 61 -> see DESIGN.md section 5).
  hipStream_t caller = st;
  if (!st) {
    if (!ws->own_stream) {
      OAVSR_CHECK_HIP(hipStreamCreateWithFlags(&ws->own_stream, hipStreamNonBlocking));
      OAVSR_CHECK_HIP(hipEventCreateWithFlags(&ws->own_in, hipEventDisableTiming));
      OAVSR_CHECK_HIP(hipEventCreateWithFlags(&ws->own_out, hipEventDisableTiming));
    }
    OAVSR_CHECK_HIP(hipEventRecord(ws->own_in, nullptr));
    OAVSR_CHECK_HIP(hipStreamWaitEvent(ws->own_stream, ws->own_in, 0));
    st = ws->own_stream;
  }
  int rc;
  { CallScope cs(ws, st);
    // ONE decode loop then serves all B rows: the LLM weights are streamed once per step for the whole group instead of once per chunk.
    rc = encode_prefill(m, ws, task, video_dev, Tv_host, wav_dev, wav_ld, Ts_host, keep_host, B, chunk_B, rate_audio, rate_video, st, nullptr, nullptr);
    if (rc == 0) rc = beams > 1 ? decode_beam(m, ws, task, B, beams, max_new, out_ids_dev, out_len_dev, st)
                                : decode_greedy(m, ws, task, B, max_new, force_len, out_ids_dev, out_len_dev, st); }
  if (!caller) {      // whatever was queued -- also after a refused input -- is ordered in front of the caller's next work
    OAVSR_CHECK_HIP(hipEventRecord(ws->own_out, ws->own_stream));
    OAVSR_CHECK_HIP(hipStreamWaitEvent(nullptr, ws->own_out, 0));
  }
  return rc;
}

int oavsr_generate(const oavsr_model* m, oavsr_workspace* ws, int32_t task, const void* video_dev, const int32_t* Tv_host, const float* wav_dev,
                   int64_t wav_ld, const int32_t* Ts_host, const int32_t* keep_host, int32_t B, int32_t rate_audio, int32_t rate_video,
                   int32_t beams, int32_t max_new, int32_t force_len, int32_t* out_ids_dev, int32_t* out_len_dev, oavsr_stream stream) {
  return generate_impl(m, ws, task, video_dev, Tv_host, wav_dev, wav_ld, Ts_host, keep_host, B, B, rate_audio, rate_video, beams, max_new, force_len,
                       out_ids_dev, out_len_dev, (hipStream_t)stream);
}

int oavsr_generate_chunked(const oavsr_model* m, oavsr_workspace* ws, int32_t task, const void* video_dev, const int32_t* Tv_host,
                           const float* wav_dev, int64_t wav_ld, const int32_t* Ts_host, const int32_t* keep_host, int32_t B, int32_t chunk_B,
                           int32_t rate_audio, int32_t rate_video, int32_t beams, int32_t max_new, int32_t force_len, int32_t* out_ids_dev,
                           int32_t* out_len_dev, oavsr_stream stream) {
  return generate_impl(m, ws, task, video_dev, Tv_host, wav_dev, wav_ld, Ts_host, keep_host, B, chunk_B, rate_audio, rate_video, beams, max_new,
                       force_len, out_ids_dev, out_len_dev, (hipStream_t)stream);
}

// ---- decode scheduler (struct oavsr_scheduler, above) --------------------------------------------------------------------------------
static int lane_rows(const oavsr_scheduler* s, const SchedLane& l) { int r = 0; for (int i : l.active) r += s->ring[i].B; return r; }
static int lane_min_remaining(const oavsr_scheduler* s, const SchedLane& l) { int r = 1 << 30; for (int i : l.active) r = std::min(r, s->ring[i].remaining); return l.active.empty() ? 0 : r; }
static int lane_max_remaining(const oavsr_scheduler* s, const SchedLane& l) { int r = 0; for (int i : l.active) r = std::max(r, s->ring[i].remaining); return r; }

// one decode step over the R compact rows of the lane's current membership (every launch on the lane's stream; hipGraph-capturable)
static int sched_one_step(oavsr_scheduler* s, SchedLane& l, int R, int task, int banned) {
  const oavsr_model* m = s->m; const oavsr_config& c = m->cfg; oavsr_workspace* w = l.dws;
  OAVSR_TRY(llm_decode_layers(m, w, task, R, 1, s->max_prefix + s->max_new, s->ss.pos, s->ss.cpos, nullptr, nullptr, l.st, s->ss.rope_cs));
  OAVSR_TRY(lm_head_rows(m, w, R, w->logits, l.st));
  { ProfScope ps("argmax_partial", 0, (double)R * c.llm_vocab * 4, l.st);
    OAVSR_TRY(launch_argmax_partial(w->logits, vstride(c), R, c.llm_vocab, banned, l.partials, l.st)); }
  ProfScope ps("sched_tail", 0, 2.0 * R * c.llm_hidden * m->esz, l.st);
  return launch_sched_tail(l.partials, s->ss, l.rowmap, 0, R, c.id_eos, c.id_pad, m->dtype, m->embed, c.llm_hidden, w->lx, c.llm_hidden, c.llm_hidden, l.st);
}

// membership changed (a call joined or left): new row map (padding rows -> the scratch slot), input rows regathered in the new order
static int sched_sync_membership(oavsr_scheduler* s, SchedLane& l) {
  if (!l.map_dirty) return 0;
  const oavsr_config& c = s->m->cfg;
  IntPack128 map;
  int r = 0;
  for (int i : l.active) { const SchedCall& k = s->ring[i]; for (int b = 0; b < k.B; ++b) map.v[r++] = k.slot0 + b; }
  const int R = std::max(8, (r + 7) / 8 * 8);
  OAVSR_REQUIRE(R <= 128, "scheduler: %d rows in the step", r);
  for (; r < R; ++r) map.v[r] = s->n_slots;
  OAVSR_TRY(launch_set_ints(l.rowmap, map, R, l.st));
  OAVSR_TRY(launch_sched_embed(s->ss, l.rowmap, R, s->m->dtype, s->m->embed, c.llm_hidden, l.dws->lx, c.llm_hidden, c.llm_hidden, l.st));
  l.cur_R = R; l.map_dirty = false;
  return 0;
}

static int sched_retire(oavsr_scheduler* s, SchedLane& l, int idx) {
  SchedCall& k = s->ring[idx];
  OAVSR_TRY(launch_sched_retire(s->ss, k.slot0, k.B, k.limit, k.out_ids, k.out_len, l.st));
  if (k.host_ids) OAVSR_CHECK_HIP(hipMemcpyAsync(k.host_ids, k.out_ids, sizeof(int32_t) * k.B * k.limit, hipMemcpyDeviceToHost, l.st));
  if (k.host_len) OAVSR_CHECK_HIP(hipMemcpyAsync(k.host_len, k.out_len, sizeof(int32_t) * k.B, hipMemcpyDeviceToHost, l.st));
  OAVSR_CHECK_HIP(hipEventRecord(k.done, l.st));
  ++s->free_seq;
  for (int b = 0; b < k.B; ++b) { s->slot_used[k.slot0 + b] = 0; s->slot_freed_at[k.slot0 + b] = s->free_seq; }
  k.retired = true;
  l.active.erase(std::find(l.active.begin(), l.active.end(), idx));
  l.map_dirty = true;
  return 0;
}

// issue up to n steps for the rows admitted to the lane; calls whose last step has been issued leave (result copy + `done` event) on the way
static int sched_run(oavsr_scheduler* s, SchedLane& l, int n) {
  const bool graphs_on = knob("OAVSR_DECODE_GRAPH") != 0;
  while (n > 0 && !l.active.empty()) {
    const int k = std::min(n, lane_min_remaining(s, l));
    OAVSR_TRY(sched_sync_membership(s, l));
    const SchedCall& head = s->ring[l.active[0]];
    const int task = head.task, banned = head.banned, R = l.cur_R;
    int done = 0;
    hipEvent_t tr0 = nullptr;
    if (s->tracing()) { tr0 = s->trace_event(); hipEventRecord(tr0, l.st); }
    if (graphs_on && !g_prof.on) {
      const DecodeGraphKey key{R, task, 0, banned};
      hipGraphExec_t exec = nullptr;
      for (auto& e : l.graphs) if (e.first == key) exec = e.second;
      if (!exec) {
        OAVSR_TRY(sched_one_step(s, l, R, task, banned));     // eager once: sets the per-kernel attributes outside of the capture
        done = 1;
        hipGraph_t graph = nullptr;
        OAVSR_CHECK_HIP(hipStreamBeginCapture(l.st, hipStreamCaptureModeRelaxed));
        int rc = sched_one_step(s, l, R, task, banned);
        hipError_t ce = hipStreamEndCapture(l.st, &graph);
        OAVSR_REQUIRE(rc == 0 && ce == hipSuccess && graph, "scheduler: hipGraph capture of the step failed (%d / %d)", rc, (int)ce);
        OAVSR_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        hipGraphDestroy(graph);
        l.graphs.push_back({key, exec});
      }
      for (; done < k; ++done) OAVSR_CHECK_HIP(hipGraphLaunch(exec, l.st));
    }
    for (; done < k; ++done) OAVSR_TRY(sched_one_step(s, l, R, task, banned));
    if (tr0) {
      hipEvent_t tr1 = s->trace_event();
      hipEventRecord(tr1, l.st);
      s->trace_recs.push_back({1, (int)(&l - s->lanes.data()), k, lane_rows(s, l), tr0, tr1});
    }
    s->steps_issued += k; l.steps += k; s->row_steps += (uint64_t)k * lane_rows(s, l);
    std::vector<int> leaving;
    for (int i : l.active) { s->ring[i].remaining -= k; if (s->ring[i].remaining <= 0) leaving.push_back(i); }
    for (int i : leaving) OAVSR_TRY(sched_retire(s, l, i));
    n -= k;
  }
  return 0;
}

// B contiguous free slots; of all candidates the block whose most recently freed slot was freed longest ago
static int sched_alloc(oavsr_scheduler* s, int B) {
  int best = -1; uint64_t best_score = ~0ull;
  for (int s0 = 0; s0 + B <= s->n_slots; ++s0) {
    uint64_t score = 0; bool ok = true;
    for (int b = 0; b < B && ok; ++b) { if (s->slot_used[s0 + b]) ok = false; else score = std::max(score, s->slot_freed_at[s0 + b]); }
    if (ok && score < best_score) { best = s0; best_score = score; }
  }
  return best;
}

int oavsr_scheduler_create(const oavsr_model* m, int32_t n_slots, int32_t max_rows, int32_t lanes, int32_t max_prefix, int32_t max_new,
                           oavsr_scheduler** out) {
  OAVSR_REQUIRE(m && out, "scheduler_create: null argument");
  OAVSR_REQUIRE(max_rows >= 1 && max_rows <= 128, "scheduler_create: max_rows=%d (a decode step is one 128-row tile of the tiled GEMMs at most)", max_rows);
  OAVSR_REQUIRE(lanes >= 1 && lanes <= 8, "scheduler_create: lanes=%d", lanes);
  OAVSR_REQUIRE(n_slots >= max_rows && n_slots <= 4096 && max_prefix >= 1 && max_new >= 1, "scheduler_create: bad sizes (n_slots >= max_rows)");
  const oavsr_config& c = m->cfg;
  oavsr_scheduler* s = new oavsr_scheduler();
  s->m = m; s->n_slots = n_slots; s->max_rows = max_rows; s->max_prefix = (max_prefix + 63) / 64 * 64; s->max_new = max_new;
  s->lanes.resize(lanes);
  const size_t e = m->esz, NS = (size_t)n_slots + 1, R = max_rows > 64 ? 128 : (max_rows > 32 ? 64 : 32);
  const int d = c.llm_hidden, H = c.llm_heads, KV = c.llm_kv_heads, hd = c.llm_head_dim, I = c.llm_mlp;
  const size_t nqkv = (size_t)(H + 2 * KV) * hd;
  oavsr_workspace* w = new oavsr_workspace();
  s->pool = w;
  w->m = m; w->max_B = n_slots; w->max_beams = 1; w->max_new = max_new; w->max_prefix = s->max_prefix;
  w->kpre_layer = NS * KV * (size_t)s->max_prefix * hd * e;
  w->ktail_layer = NS * KV * (size_t)max_new * hd * e;
  struct Req { void** p; size_t bytes; };
  std::vector<Req> reqs;
#define WANT(ptr, bytes) reqs.push_back({(void**)&(ptr), (size_t)(bytes)})
  WANT(w->kpre, w->kpre_layer * c.llm_layers); WANT(w->vpre, w->kpre_layer * c.llm_layers);
  WANT(w->ktail, w->ktail_layer * c.llm_layers); WANT(w->vtail, w->ktail_layer * c.llm_layers);
  WANT(w->prefix_len, NS * 4);
  WANT(s->logits_pre, (size_t)n_slots * vstride(c) * 4);
  WANT(s->ss.tok, NS * 4); WANT(s->ss.finished, NS * 4); WANT(s->ss.out_ids, NS * (size_t)max_new * 4); WANT(s->ss.out_len, NS * 4);
  WANT(s->ss.nstep, NS * 4); WANT(s->ss.limit, NS * 4); WANT(s->ss.pos, NS * 4); WANT(s->ss.cpos, NS * 4);
  WANT(s->ss.rope_cs, NS * (size_t)(hd / 2) * sizeof(float2));
  for (auto& l : s->lanes) {
    l.dws = new oavsr_workspace();
    oavsr_workspace* lw = l.dws;
    WANT(lw->lx, R * d * e); WANT(lw->lh, R * d * e); WANT(lw->lqkv, R * nqkv * e); WANT(lw->lattn, R * (size_t)H * hd * e);
    WANT(lw->lact, R * (size_t)I * e); WANT(lw->llora, R * (size_t)std::max(64, c.lora_k2) * e); WANT(lw->logits, R * (size_t)vstride(c) * 4);
    WANT(l.partials, R * 64 * 8); WANT(l.rowmap, 128 * 4);
    {   // the decode-shaped GEMMs' split-K slabs, arrival counters, deferred-norm scratch and (when SwiGLU is not fused) the [rows][2I] gate / up tensor
      const int sites[5][3] = {{c.lora_k2, d, 1}, {(int)nqkv, d + c.lora_k2, 1}, {d, H * hd, 2}, {2 * I, d, 1}, {d, I, 2}};
      size_t pb = 0;
      for (auto& s2 : sites) if (s2[0] > 0) pb = std::max(pb, (size_t)decode_split(s2[0], s2[1], (int)e, s2[2]) * 128 * s2[0] * 4);
      lw->partial_bytes = pb;
      WANT(lw->partial, std::max<size_t>(pb, 256)); WANT(lw->lgu, R * 2 * (size_t)I * e);
      WANT(lw->gsync, kSyncInts * 4); WANT(lw->rowsq, rowsq_chunk(d) * 4);
    }
  }
#undef WANT
  size_t total = 0;
  for (auto& r : reqs) total = align_up(total, 256) + std::max<size_t>(r.bytes, 256);
  total += 4096;
  auto fail = [&](const char* what) {
    set_last_error("scheduler_create: %s", what);
    for (auto& l : s->lanes) { if (l.st) hipStreamDestroy(l.st); delete l.dws; }
    if (s->arena) hipFree(s->arena);
    delete w; delete s;
    return -2;
  };
  if (hipMalloc((void**)&s->arena, total) != hipSuccess) { s->arena = nullptr; return fail("hipMalloc failed"); }
  s->arena_bytes = total;
  if (hipMemset(s->arena, 0, total) != hipSuccess) return fail("hipMemset failed");
  { size_t off = 0; for (auto& r : reqs) { off = align_up(off, 256); *r.p = s->arena + off; off += std::max<size_t>(r.bytes, 256); } }
  s->ss.inv_freq = m->inv_freq; s->ss.rope_half = hd / 2; s->ss.out_ld = max_new; s->ss.scratch_slot = n_slots;
  s->slot_owner.assign(n_slots, -1); s->slot_used.assign(n_slots, 0); s->slot_freed_at.assign(n_slots, 0);
  s->ring.resize(oavsr_scheduler::kRing);
  // Stream creation order matters for speed (never for results): ROCm binds a stream to one of its few hardware queues when the stream is
  // created, and which caller stream shares a queue with which lane decides how encoder phases and decode bursts take turns (DESIGN.md
  // section 5).  The engine therefore creates the CALLER streams it hands out (oavsr_scheduler_caller_stream) first, then the lanes: the
  // order the round-3 scan found in the fast regime.  (Priority streams, CU-masked lanes, eagerly bound queues and placeholder streams
  // were measured in round 3 and removed in round 4: profiles/r03_sched_scan.txt.)
  // (they belong to the PROCESS -- one process per GPU: created once, with the first scheduler, and never destroyed, so that a host
  // framework that has seen them, e.g. PyTorch's caching allocator after record_stream, never holds a dangling stream)
  if (hipGetDevice(&s->device) != hipSuccess) return fail("hipGetDevice failed");
  { std::lock_guard<std::mutex> lk(g_caller_mu);
    if (!g_caller_streams.count(s->device)) {      // all eight or none: a failure half way leaves no partial table behind
      std::vector<hipStream_t> v;
      for (int i = 0; i < kCallerStreams; ++i) {
        hipStream_t cs = nullptr;
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { for (hipStream_t x : v) hipStreamDestroy(x); return fail("stream creation failed"); }
        v.push_back(cs);
      }
      g_caller_streams[s->device] = v;
    } }
  for (auto& l : s->lanes) {
    oavsr_workspace* lw = l.dws;      // the lane's view: its own scratch, the shared pool
    lw->m = m; lw->max_B = n_slots; lw->max_beams = 1; lw->max_new = max_new; lw->max_prefix = s->max_prefix;
    lw->kpre = w->kpre; lw->vpre = w->vpre; lw->ktail = w->ktail; lw->vtail = w->vtail; lw->kpre_layer = w->kpre_layer; lw->ktail_layer = w->ktail_layer;
    lw->prefix_len = w->prefix_len; lw->rowmap = l.rowmap;
    if (hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking) != hipSuccess) {
      l.st = nullptr;
      return fail("stream creation failed");
    }
  }
  if (hipDeviceSynchronize() != hipSuccess) return fail("device synchronisation failed");
  *out = s;
  return 0;
}

void oavsr_scheduler_destroy(oavsr_scheduler* s) {
  if (!s) return;
  for (auto& l : s->lanes) {
    if (l.st) hipStreamSynchronize(l.st);
    for (auto& e : l.graphs) hipGraphExecDestroy(e.second);
    if (l.st) hipStreamDestroy(l.st);
    delete l.dws;
  }
  for (auto& k : s->ring) { if (k.prefilled) hipEventDestroy(k.prefilled); if (k.done) hipEventDestroy(k.done); }
  for (auto& r : s->trace_recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  for (auto e : s->trace_pool) hipEventDestroy(e);
  if (s->arena) hipFree(s->arena);
  delete s->pool;
  delete s;
}
uint64_t oavsr_scheduler_bytes(const oavsr_scheduler* s) { return s ? s->arena_bytes : 0; }
oavsr_stream oavsr_scheduler_caller_stream(oavsr_scheduler* s, int32_t i) {
  std::lock_guard<std::mutex> lk(g_caller_mu);
  if (!s || i < 0 || i >= kCallerStreams) { set_last_error("scheduler_caller_stream: index %d out of range (0..%d)", (int)i, kCallerStreams - 1); return nullptr; }
  auto it = g_caller_streams.find(s->device);      // the streams of the device THIS scheduler lives on
  if (it == g_caller_streams.end()) { set_last_error("scheduler_caller_stream: no streams for device %d", s->device); return nullptr; }
  return (oavsr_stream)it->second[i];
}

int oavsr_generate_submit(const oavsr_model* m, oavsr_scheduler* s, oavsr_workspace* ws, int32_t task, const void* video_dev, const int32_t* Tv_host,
                          const float* wav_dev, int64_t wav_ld, const int32_t* Ts_host, const int32_t* keep_host, int32_t B, int32_t rate_audio,
                          int32_t rate_video, int32_t max_new, int32_t force_len, int32_t* out_ids_dev, int32_t* out_len_dev, int32_t* out_ids_host,
                          int32_t* out_len_host, oavsr_stream stream, int64_t* ticket_out) {
  OAVSR_REQUIRE(m && s && ws && out_ids_dev && out_len_dev && ticket_out, "generate_submit: null argument");
  OAVSR_REQUIRE(s->m == m && ws->m == m, "generate_submit: scheduler / workspace belong to another model");
  OAVSR_REQUIRE(task >= 0 && task < 3, "generate_submit: bad task %d", task);
  OAVSR_REQUIRE(B >= 1 && B <= ws->max_B && B <= s->max_rows, "generate_submit: B=%d exceeds the workspace (%d) or the scheduler's rows per step (%d)", B, ws->max_B, s->max_rows);
  OAVSR_REQUIRE(max_new >= 1 && max_new <= s->max_new, "generate_submit: max_new=%d exceeds the scheduler's (%d)", max_new, s->max_new);
  hipStream_t st = (hipStream_t)stream;
  const oavsr_config& c = m->cfg;
  std::lock_guard<std::mutex> lk(s->mu);
  const int ltask = c.lora_task_specific ? task : 0;      // rows of one step share the LoRA adapter set (Llama_LoRA.py:250-259) and the banned token
  const int banned = force_len ? c.id_eos : -1;
  // the lane with the fewest rows (ties: round robin by ticket)
  const int64_t ticket = s->next_ticket;
  const int nl = (int)s->lanes.size();
  int li = (int)(ticket % nl);
  for (int i = 0; i < nl; ++i) if (lane_rows(s, s->lanes[i]) < lane_rows(s, s->lanes[li])) li = i;
  SchedLane& l = s->lanes[li];
  if (!l.active.empty() && (s->ring[l.active[0]].task != ltask || s->ring[l.active[0]].banned != banned)) OAVSR_TRY(sched_run(s, l, lane_max_remaining(s, l)));
  // room: rows per step of the lane and a block of row slots
  const uint64_t steps0 = l.steps;
  while (lane_rows(s, l) + B > s->max_rows) OAVSR_TRY(sched_run(s, l, lane_min_remaining(s, l)));
  int slot0 = sched_alloc(s, B);
  while (slot0 < 0) {      // no block: let the lane that holds the call closest to its end advance
    SchedLane* v = nullptr;
    for (auto& x : s->lanes) if (!x.active.empty() && (!v || lane_min_remaining(s, x) < lane_min_remaining(s, *v))) v = &x;
    if (!v) break;
    OAVSR_TRY(sched_run(s, *v, lane_min_remaining(s, *v)));
    slot0 = sched_alloc(s, B);
  }
  OAVSR_REQUIRE(slot0 >= 0, "generate_submit: no block of %d free row slots (pool of %d)", B, s->n_slots);
  SchedCall& k = s->ring[ticket % oavsr_scheduler::kRing];
  OAVSR_REQUIRE(k.retired, "generate_submit: %d calls in flight", oavsr_scheduler::kRing);
  if (k.done && k.ticket >= 0) OAVSR_CHECK_HIP(hipEventSynchronize(k.done));      // kRing tickets ago: long complete; makes "ticket expired => its device work is over" exact
  if (!k.prefilled) OAVSR_CHECK_HIP(hipEventCreateWithFlags(&k.prefilled, hipEventDisableTiming));
  if (!k.done) OAVSR_CHECK_HIP(hipEventCreateWithFlags(&k.done, hipEventDisableTiming));
  // the slots' previous owners must have left (their last step is already in their lane's queue) before this call's prefill writes there
  { int64_t seen = -1;
    for (int b = 0; b < B; ++b) {
      const int64_t t = s->slot_owner[slot0 + b];
      if (t < 0 || t == seen) continue;
      seen = t;
      const SchedCall& o = s->ring[t % oavsr_scheduler::kRing];
      if (o.ticket == t && o.done) OAVSR_CHECK_HIP(hipStreamWaitEvent(st, o.done, 0));
    } }
  std::vector<int> S(B);
  hipEvent_t tr0 = nullptr;
  if (s->tracing()) { tr0 = s->trace_event(); hipEventRecord(tr0, st); }
  const int ridx = (int)(ticket % oavsr_scheduler::kRing);
  // The call dies after it may have queued work that touches its slots (a refused input half way through the encoders, a failed launch
  // during admission): the ticket is consumed, the slots go back to the pool, and the ring entry's `done` event is recorded on `where` --
  // behind everything the call queued -- so the block's next owner (which waits for the previous owner's `done`) is ordered after it.
  // Nothing stays "in flight": flush() and the ring never see the call.
  auto abandon = [&](hipStream_t where) {
    k.ticket = ticket; k.task = ltask; k.B = B; k.slot0 = slot0; k.limit = max_new; k.remaining = 0; k.banned = banned; k.lane = li;
    k.out_ids = nullptr; k.out_len = nullptr; k.host_ids = nullptr; k.host_len = nullptr; k.retired = true;
    (void)hipEventRecord(k.done, where);
    auto it = std::find(l.active.begin(), l.active.end(), ridx);
    if (it != l.active.end()) { l.active.erase(it); l.map_dirty = true; }
    ++s->free_seq;
    for (int b = 0; b < B; ++b) { s->slot_used[slot0 + b] = 0; s->slot_owner[slot0 + b] = ticket; s->slot_freed_at[slot0 + b] = s->free_seq; }
    ++s->next_ticket;
  };
  { CallScope cs(ws, st);
    PrefillDest dest{s->pool->kpre, s->pool->vpre, s->pool->kpre_layer, s->max_prefix, s->logits_pre, s->pool->prefix_len, slot0};
    const int rc = encode_prefill(m, ws, task, video_dev, Tv_host, wav_dev, wav_ld, Ts_host, keep_host, B, B, rate_audio, rate_video, st, &dest, S.data());
    if (rc != 0) {                                   // refused inputs: nothing of the call has reached a lane
      if (tr0) s->trace_pool.push_back(tr0);
      abandon(st);
      return rc;
    } }
  // admission: everything below either completes or is rolled back (abandon) -- the scheduler's state is committed only with the last step
  auto admit = [&]() -> int {
    OAVSR_CHECK_HIP(hipEventRecord(k.prefilled, st));
    if (tr0) {
      hipEvent_t tr1 = s->trace_event();
      hipEventRecord(tr1, st);
      s->trace_recs.push_back({0, (int)(ticket & 0x7fffffff), 0, B, tr0, tr1});
    }
    // pacing: the lane advances 1/k of a hypothesis per admission (k = calls of this size that share a step), beside this call's MFMA-bound phase
    const int share = std::max(1, s->max_rows / B);
    const int stride = s->stride > 0 ? s->stride : std::max(1, (max_new - 1) / share);
    const int issued = (int)(l.steps - steps0);                 // what making room already cost this lane
    if (stride > issued) OAVSR_TRY(sched_run(s, l, stride - issued));
    // after the call's prefill; first token from the prefill's logits (step 0 of the row's own counter)
    OAVSR_CHECK_HIP(hipStreamWaitEvent(l.st, k.prefilled, 0));
    IntPack128 Sp;
    for (int b = 0; b < B; ++b) Sp.v[b] = S[b];
    OAVSR_TRY(launch_sched_init_slots(s->ss, slot0, B, Sp, max_new, c.id_pad, l.st));
    OAVSR_TRY(launch_argmax_partial(s->logits_pre + (size_t)slot0 * vstride(c), vstride(c), B, c.llm_vocab, banned, l.partials, l.st));
    OAVSR_TRY(launch_sched_tail(l.partials, s->ss, nullptr, slot0, B, c.id_eos, c.id_pad, m->dtype, m->embed, c.llm_hidden, nullptr, 0, c.llm_hidden, l.st));
    return 0;
  };
  // the slots are taken while the pacing steps run (sched_alloc of nobody else may see them free), the rest of the entry follows the launches
  for (int b = 0; b < B; ++b) s->slot_used[slot0 + b] = 1;
  const int arc = admit();
  if (arc != 0) {
    (void)hipStreamWaitEvent(l.st, k.prefilled, 0);    // best effort: `done` then also sits behind the encoder phase on the caller's stream
    abandon(l.st);
    return arc;
  }
  for (int b = 0; b < B; ++b) s->slot_owner[slot0 + b] = ticket;
  ++s->next_ticket; ++s->calls;
  k.ticket = ticket; k.task = ltask; k.B = B; k.slot0 = slot0; k.limit = max_new; k.remaining = max_new - 1; k.banned = banned; k.lane = li;
  k.out_ids = out_ids_dev; k.out_len = out_len_dev; k.host_ids = out_ids_host; k.host_len = out_len_host; k.retired = false;
  l.active.push_back(ridx);
  l.map_dirty = true;
  *ticket_out = ticket;
  if (k.remaining <= 0) OAVSR_TRY(sched_retire(s, l, ridx));      // max_new == 1: the first token is the whole hypothesis (a failure here leaves a consistent, retirable entry)
  return 0;
}

int oavsr_generate_wait(oavsr_scheduler* s, int64_t ticket, oavsr_stream stream) {
  OAVSR_REQUIRE(s, "generate_wait: null scheduler");
  std::lock_guard<std::mutex> lk(s->mu);
  OAVSR_REQUIRE(ticket >= 0 && ticket < s->next_ticket, "generate_wait: unknown ticket %lld", (long long)ticket);
  SchedCall& k = s->ring[ticket % oavsr_scheduler::kRing];
  OAVSR_REQUIRE(k.ticket == ticket, "generate_wait: ticket %lld has expired (more than %d calls submitted since)", (long long)ticket, oavsr_scheduler::kRing);
  while (!k.retired) OAVSR_TRY(sched_run(s, s->lanes[k.lane], k.remaining));
  OAVSR_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, k.done, 0));
  return 0;
}

int oavsr_generate_poll(oavsr_scheduler* s, int64_t ticket) {
  OAVSR_REQUIRE(s, "generate_poll: null scheduler");
  std::lock_guard<std::mutex> lk(s->mu);
  OAVSR_REQUIRE(ticket >= 0 && ticket < s->next_ticket, "generate_poll: unknown ticket %lld", (long long)ticket);
  const SchedCall& k = s->ring[ticket % oavsr_scheduler::kRing];
  OAVSR_REQUIRE(k.ticket == ticket, "generate_poll: ticket %lld has expired", (long long)ticket);
  if (!k.retired) return 0;                              // steps still to be issued: nothing is forced here
  const hipError_t e = hipEventQuery(k.done);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  set_last_error("generate_poll: HIP error %d (%s)", (int)e, hipGetErrorString(e));
  return -2;
}

int oavsr_scheduler_flush(oavsr_scheduler* s) {
  OAVSR_REQUIRE(s, "scheduler_flush: null scheduler");
  std::lock_guard<std::mutex> lk(s->mu);
  for (auto& l : s->lanes) while (!l.active.empty()) OAVSR_TRY(sched_run(s, l, lane_max_remaining(s, l)));
  return 0;
}

int oavsr_scheduler_set_stride(oavsr_scheduler* s, int32_t steps) {
  OAVSR_REQUIRE(s && steps >= 0, "scheduler_set_stride: bad argument");
  std::lock_guard<std::mutex> lk(s->mu);
  s->stride = steps;
  return 0;
}

int oavsr_scheduler_trace(oavsr_scheduler* s, int32_t on) {
  OAVSR_REQUIRE(s, "scheduler_trace: null scheduler");
  std::lock_guard<std::mutex> lk(s->mu);
  if (on && !s->trace) {
    OAVSR_CHECK_HIP(hipDeviceSynchronize());
    for (auto& r : s->trace_recs) { s->trace_pool.push_back(r.e0); s->trace_pool.push_back(r.e1); }
    s->trace_recs.clear();
  }
  s->trace = on != 0;
  return 0;
}

int oavsr_scheduler_trace_read(oavsr_scheduler* s, oavsr_trace_rec* out, int32_t max_recs) {
  OAVSR_REQUIRE(s && (out || max_recs == 0), "scheduler_trace_read: null argument");
  std::lock_guard<std::mutex> lk(s->mu);
  OAVSR_CHECK_HIP(hipDeviceSynchronize());
  const int n = std::min<int>((int)s->trace_recs.size(), max_recs);
  if (n == 0) return 0;
  const hipEvent_t base = s->trace_recs[0].e0;           // the first record's start: every time below is relative to it
  for (int i = 0; i < n; ++i) {
    const auto& r = s->trace_recs[i];
    float t0 = 0.f, t1 = 0.f;
    if (hipEventElapsedTime(&t0, base, r.e0) != hipSuccess || hipEventElapsedTime(&t1, base, r.e1) != hipSuccess) { t0 = t1 = -1.f; }
    out[i].kind = r.kind; out[i].id = r.id; out[i].steps = r.steps; out[i].rows = r.rows; out[i].t0_ms = t0; out[i].t1_ms = t1;
  }
  return n;
}

int oavsr_scheduler_stats(const oavsr_scheduler* s, uint64_t* out4) {
  OAVSR_REQUIRE(s && out4, "scheduler_stats: null argument");
  std::lock_guard<std::mutex> lk(s->mu);
  uint64_t g = 0;
  for (auto& l : s->lanes) g += l.graphs.size();
  out4[0] = s->steps_issued; out4[1] = s->row_steps; out4[2] = s->calls; out4[3] = g;
  return 0;
}

int oavsr_profile_enable(int32_t on) { g_prof.on = on != 0; return 0; }
int oavsr_profile_reset(void) { hipDeviceSynchronize(); g_prof.reset(); return 0; }
int oavsr_profile_read(oavsr_prof_entry* out, int32_t max_entries) {
  if (hipDeviceSynchronize() != hipSuccess) { set_last_error("profile_read: device sync failed"); return -2; }
  std::lock_guard<std::mutex> lk(g_prof.mu);
  const int nf = (int)g_prof.names.size();
  std::vector<oavsr_prof_entry> agg(nf);
  for (int i = 0; i < nf; ++i) { memset(&agg[i], 0, sizeof(oavsr_prof_entry)); strncpy(agg[i].name, g_prof.names[i].c_str(), 31); }
  // what an empty bracket reads (event-to-event processing on the stream): taken off every record, so that the per-launch averages
  // line up with the kernel durations of a rocprofv3 --kernel-trace of the same command also for the 5-10 us decode kernels
  float base = 0.f;
  { std::vector<float> v;
    for (auto& r : g_prof.empty) { float ms = 0.f; if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) v.push_back(ms); }
    if (!v.empty()) { std::sort(v.begin(), v.end()); base = v[v.size() / 2]; } }
  for (auto& r : g_prof.recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    ms = std::max(0.f, ms - base);
    agg[r.fam].launches += 1; agg[r.fam].ms += ms; agg[r.fam].flops += r.flops; agg[r.fam].bytes += r.bytes; agg[r.fam].cu_ms += ms * r.cu_share;
  }
  int n = std::min(nf, (int)max_entries);
  for (int i = 0; i < n; ++i) out[i] = agg[i];
  return n;
}

// ---- single-kernel entry points --------------------------------------------------------------------
static GemmArgs gemm_args_of(const oavsr_gemm_desc* d) {
  GemmArgs g; g.dtype = d->dtype; g.M = d->M; g.N = d->N; g.K = d->K; g.K2 = d->K2; g.A = d->A; g.W = d->W; g.A2 = d->A2; g.W2 = d->W2;
  g.lda = d->lda; g.ldw = d->ldw; g.lda2 = d->lda2; g.ldw2 = d->ldw2; g.C = d->C; g.ldc = d->ldc; g.out_f32 = d->out_f32;
  g.col_scale = d->col_scale; g.col_bias = d->col_bias; g.prelu = d->prelu; g.act = d->act; g.residual = d->residual; g.ldr = d->ldr;
  g.res_after_act = d->res_after_act; g.groups = d->groups < 1 ? 1 : d->groups; g.a_goff = d->a_goff; g.amode = d->amode;
  g.H = d->H; g.Wd = d->W_; g.Cin = d->Cin; g.OH = d->OH; g.OW = d->OW; g.stride = d->stride; g.pad = d->pad; g.ksize = d->ksize;
  g.rowinfo = (const int4*)d->rowinfo; g.taps = d->taps; g.Kvalid = d->Kvalid; g.frameinfo = (const int2*)d->frameinfo;
  g.swiglu_pairs = d->swiglu_pairs; g.decode_tiled = d->decode_tiled != 0;
  g.ksplit = d->ksplit < 1 ? 1 : d->ksplit; g.partial = d->partial; g.sync = d->sync;
  g.row_sumsq = d->row_sumsq; g.row_sumsq_tiles = d->row_sumsq_tiles; g.row_dim = d->row_dim; g.row_eps = d->row_eps;
  g.ng_out = d->ng_out; g.ng_ld = d->ng_ld; g.ng_gain = d->ng_gain; g.ng_rowsq = d->ng_rowsq;
  g.rowsq_strips = d->rowsq_strips; g.row_sumsq_form = d->row_sumsq_form;
  return g;
}

int oavsr_op_gemm(const oavsr_gemm_desc* d, oavsr_stream stream) {
  OAVSR_REQUIRE(d, "op_gemm: null desc");
  GemmArgs g = gemm_args_of(d);
  OAVSR_REQUIRE(!g.ng_out || g.ksplit > 1, "op_gemm: the next-norm outputs need split-K (ksplit > 1)");
  OAVSR_REQUIRE(!g.row_sumsq || d->kernel == 0, "op_gemm: the row factor is an epilogue of the decode-shaped launches (kernel selector 0)");
  if (g.row_sumsq) g.decode_tiled = 1;
  if (d->kernel == 2) {      // tests: the 8-wave ping-pong 256 x 256 kernel whatever the dispatcher's rule says
    OAVSR_REQUIRE(gemm_8p_supported(g), "op_gemm: shape / layout not supported by the ping-pong kernel");
    return launch_gemm_8p(g, (hipStream_t)stream);
  }
  if (d->kernel == 3) return launch_gemm_8pp(g, (hipStream_t)stream, d->grid, 256);      // tests: the persistent ping-pong kernel on d->grid workgroups, 256-row tiles
  if (d->kernel == 8) return launch_gemm_8pp(g, (hipStream_t)stream, d->grid, 192);      // ... its 192-row form
  if (d->kernel == 5) {      // tests / yardsticks: the 16-wave 256 x 256 LDS-DMA kernel whatever the dispatcher's rule says
    g.force_tile = 2;
    return launch_gemm(g, (hipStream_t)stream);
  }
  if (d->kernel == 7) {      // tests / yardsticks: the 256 x 128 LDS-DMA tile whatever the dispatcher's rule says
    g.force_tile = 8;
    return launch_gemm_dma(g, (hipStream_t)stream);
  }
  OAVSR_REQUIRE(d->kernel == 0, "op_gemm: unknown kernel selector %d", d->kernel);
  return launch_gemm(g, (hipStream_t)stream);
}

int oavsr_op_gemm_chain(const oavsr_gemm_desc* d, int32_t n, int32_t* counters, oavsr_stream stream) {
  OAVSR_REQUIRE(d && counters && n >= 1 && n <= kStripChainMax, "op_gemm_chain: bad arguments (n = %d, at most %d launches)", n, kStripChainMax);
  GemmArgs g[kStripChainMax];
  for (int i = 0; i < n; ++i) {
    OAVSR_REQUIRE(d[i].kernel == 0, "op_gemm_chain: kernel selector 0 only");
    g[i] = gemm_args_of(&d[i]);
    OAVSR_REQUIRE(!g[i].ng_out || g[i].ksplit > 1, "op_gemm_chain: the next-norm outputs need split-K (ksplit > 1)");
    if (g[i].row_sumsq) g[i].decode_tiled = 1;
    OAVSR_REQUIRE(gemm_strip_selected(g[i]), "op_gemm_chain: launch %d (%d x %d x %d) is not one the <= %d-row strips take", i, g[i].M, g[i].N, g[i].K + g[i].K2, kStripRows);
  }
  return launch_gemm_strip_chain(g, n, counters, counters + n, (hipStream_t)stream);
}

int oavsr_op_stem_pool(int32_t dtype, const void* video, int32_t channels, int32_t height, int32_t width, const int32_t* frameinfo_dev, int32_t F,
                       const void* w, void* wp_scratch, const float* scale, const float* shift, const float* slope, void* out, oavsr_stream stream) {
  OAVSR_REQUIRE(w && wp_scratch, "op_stem_pool: null weights");
  OAVSR_TRY(launch_stem_repack(dtype, w, wp_scratch, (hipStream_t)stream));
  return launch_stem_pool(dtype, video, (const int2*)frameinfo_dev, F, wp_scratch, scale, shift, slope, out, (hipStream_t)stream,
                          channels ? StemSrc::u8(channels, height, width) : StemSrc());
}

int oavsr_op_conv3x3_c64(const void* x, int32_t imgs, int32_t H, int32_t W, const void* w, const float* scale, const float* shift,
                         const float* slope, const void* residual, void* out, oavsr_stream stream) {
  OAVSR_REQUIRE(x && w && scale && shift && out, "op_conv3x3_c64: null argument");
  OAVSR_REQUIRE(conv3x3_c64_eligible(DT_BF16, H, W, 64, 64, 3, 1), "op_conv3x3_c64: %dx%d image does not fit", H, W);
  return launch_conv3x3_c64(x, imgs, H, W, w, scale, shift, slope, residual, out, (hipStream_t)stream);
}

int oavsr_op_conv3x3_c128(const void* x, int32_t imgs, int32_t H, int32_t W, const void* w, const float* scale, const float* shift,
                          const float* slope, const void* residual, void* out, oavsr_stream stream) {
  OAVSR_REQUIRE(x && w && scale && shift && out, "op_conv3x3_c128: null argument");
  OAVSR_REQUIRE(H * W <= 128 && H * W >= 16, "op_conv3x3_c128: %dx%d image does not fit (16 .. 128 pixels)", H, W);
  return launch_conv3x3_c128(x, imgs, H, W, w, scale, shift, slope, residual, out, (hipStream_t)stream);
}

int oavsr_op_beam_topk(const float* logits, int64_t ld, int32_t rows, int32_t V, int32_t K, const float* beam_scores, float* cand_val,
                       int32_t* cand_tok, int32_t* need_exact, oavsr_stream stream) {
  OAVSR_REQUIRE(logits && cand_val && cand_tok, "op_beam_topk: null argument");
  return launch_beam_topk(logits, ld, rows, V, K, beam_scores, cand_val, cand_tok, need_exact, (hipStream_t)stream);
}

int oavsr_op_attention(const oavsr_attn_desc* d, oavsr_stream stream) {
  OAVSR_REQUIRE(d, "op_attention: null desc");
  AttnArgs a; a.dtype = d->dtype; a.head_dim = d->head_dim; a.n_heads = d->n_heads; a.n_kv_heads = d->n_kv_heads; a.n_seqs = d->n_seqs;
  a.max_q = d->max_q; a.causal = d->causal; a.Q = d->Q; a.K = d->K; a.V = d->V; a.O = d->O; a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv;
  a.ldo = d->ldo; a.cu_q = d->cu_q; a.cu_k = d->cu_k; a.scale = d->scale; a.form = d->form;
  return launch_attention(a, (hipStream_t)stream);
}

int oavsr_op_decode_attention(const oavsr_decode_attn_desc* d, oavsr_stream stream) {
  OAVSR_REQUIRE(d, "op_decode_attention: null desc");
  DecodeAttnArgs a; a.dtype = d->dtype; a.head_dim = d->head_dim; a.n_heads = d->n_heads; a.n_kv_heads = d->n_kv_heads; a.rows = d->rows;
  a.beams = d->beams; a.max_prefix = d->max_prefix; a.max_tail = d->max_tail; a.Q = d->Q; a.Kp = d->Kp; a.Vp = d->Vp; a.Kt = d->Kt; a.Vt = d->Vt;
  a.O = d->O; a.ldq = d->ldq; a.ldo = d->ldo; a.prefix_len = d->prefix_len; a.step = d->step; a.anc = d->anc; a.scale = d->scale;
  a.inv_freq = d->inv_freq; a.pos = d->pos; a.cpos = d->cpos;
  return launch_decode_attention(a, (hipStream_t)stream);
}

int oavsr_op_norm(int32_t mode, int32_t dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, const float* b, int32_t rows,
                  int32_t D, float eps, oavsr_stream stream) {
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) return launch_layernorm(dtype, x, ldx, y, ldy, w, b, rows, D, eps, st);
  if (mode == 1) return launch_rmsnorm(dtype, x, ldx, y, ldy, w, rows, D, eps, st);
  if (mode == 2) return launch_layernorm_zero_prefix(dtype, x, ldx, y, ldy, w, b, rows, D, eps, st);
  set_last_error("op_norm: bad mode %d", mode);
  return -1;
}
int oavsr_op_rms_prep(int32_t dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, float* sumsq, int32_t rows, int32_t D,
                      oavsr_stream stream) {
  OAVSR_REQUIRE(x && y && w && sumsq, "op_rms_prep: null argument");
  return launch_rms_prep(dtype, x, ldx, y, ldy, w, sumsq, rows, D, (hipStream_t)stream);
}
int oavsr_op_maxpool3x3s2(int32_t dtype, const void* x, void* y, int32_t imgs, int32_t H, int32_t W, int32_t C, oavsr_stream stream) {
  return launch_maxpool3x3s2(dtype, x, y, imgs, H, W, C, (hipStream_t)stream);
}
int oavsr_op_avgpool_hw(int32_t dtype, const void* x, void* y, int32_t imgs, int32_t HW, int32_t C, oavsr_stream stream) {
  return launch_avgpool_hw(dtype, x, y, imgs, HW, C, (hipStream_t)stream);
}
int oavsr_op_time_compress(int32_t dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const int32_t* seg_dev, int32_t n_seqs, int32_t max_out,
                           int32_t rate, int32_t C, int32_t stack, oavsr_stream stream) {
  return launch_time_compress(dtype, x, ldx, y, ldy, (const int4*)seg_dev, n_seqs, max_out, rate, C, stack, (hipStream_t)stream);
}
int oavsr_op_copy_rows(int32_t dtype, const void* src, int64_t lds, const int32_t* map_dev, void* y, int64_t ldy, int32_t n, int32_t D,
                       oavsr_stream stream) {
  return launch_copy_rows(dtype, src, lds, map_dev, y, ldy, n, D, (hipStream_t)stream);
}
int oavsr_op_rope_kv(int32_t dtype, int32_t rows, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, void* qkv, int64_t ld, const float* inv_freq,
                     const int32_t* pos, const int32_t* slot, const int32_t* cpos, void* kcache, void* vcache, int32_t max_ctx, oavsr_stream stream) {
  RopeArgs r; r.dtype = dtype; r.rows = rows; r.n_heads = n_heads; r.n_kv_heads = n_kv_heads; r.head_dim = head_dim; r.qkv = qkv; r.ld = ld;
  r.inv_freq = inv_freq; r.pos = pos; r.slot = slot; r.cpos = cpos; r.kcache = kcache; r.vcache = vcache; r.max_ctx = max_ctx;
  return launch_rope_kv(r, (hipStream_t)stream);
}
int oavsr_op_swiglu(int32_t dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t I, int32_t interleaved,
                    oavsr_stream stream) {
  return launch_swiglu(dtype, x, ldx, y, ldy, rows, I, interleaved, (hipStream_t)stream);
}
int oavsr_op_argmax(const float* logits, int64_t ld, int32_t rows, int32_t V, int32_t banned, int32_t* out, void* scratch, oavsr_stream stream) {
  return launch_argmax(logits, ld, rows, V, banned, out, scratch, (hipStream_t)stream);
}
int oavsr_op_build_rowinfo(const int32_t* cu_out, const int32_t* cu_in, int32_t n_seqs, int32_t stride, int32_t* rowinfo, int32_t rows,
                           oavsr_stream stream) {
  return launch_build_rowinfo(cu_out, cu_in, n_seqs, stride, (int4*)rowinfo, rows, (hipStream_t)stream);
}
int oavsr_op_logmel(const oavsr_model* m, int32_t dtype, const float* wav, int64_t ld, const int32_t* lens_dev, int32_t B, float* mel_ws,
                    float* max_ws, void* out, int64_t ldo, oavsr_stream stream) {
  OAVSR_REQUIRE(m, "op_logmel: null model");
  return launch_logmel(dtype, wav, ld, lens_dev, B, m->twiddle, m->melfb, mel_ws, max_ws, out, ldo, (hipStream_t)stream);
}

}  // extern "C"
