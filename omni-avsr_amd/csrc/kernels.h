// Internal launcher API of the gfx950 kernels (host side).  Every launcher is asynchronous on the
// given stream, allocates nothing and never synchronises (hipGraph-capturable).
#pragma once
#include "common.h"

namespace oavsr {

enum AMode : int { A_DENSE = 0, A_CONV2D = 1, A_CONV1D = 2, A_STEM = 3 };
enum Act : int { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_PRELU = 3, ACT_SILU = 4 };

// C[M,N] = epi( A1[M,K1] . W1[N,K1]^T + A2[M,K2] . W2[N,K2]^T )
// W is stored [N][K] (K contiguous) exactly like torch.nn.Linear.weight.
struct GemmArgs {
  int dtype = DT_BF16;       // element type of A, W, residual (and C unless out_f32)
  int M = 0, N = 0;
  const void* A = nullptr;  int64_t lda = 0;
  const void* W = nullptr;  int64_t ldw = 0;  int K = 0;     // K1 (elements), multiple of 128B/elt
  const void* A2 = nullptr; int64_t lda2 = 0;
  const void* W2 = nullptr; int64_t ldw2 = 0; int K2 = 0;    // optional second K segment (LoRA)
  void* C = nullptr; int64_t ldc = 0; int out_f32 = 0;
  // epilogue: y = acc*col_scale[n] + col_bias[n]; then residual/activation
  const float* col_scale = nullptr;
  const float* col_bias = nullptr;
  const float* prelu = nullptr;      // per-column slope (ACT_PRELU)
  int act = ACT_NONE;
  const void* residual = nullptr; int64_t ldr = 0;
  int res_after_act = 0;             // 0: act(y + res)   1: res + act(y)
  // grouped / batched: blockIdx.z = g; A channel offset g*a_goff, W row offset g*N, C col offset g*N
  int groups = 1; int a_goff = 0;
  // A gather
  int amode = A_DENSE;
  // A_CONV2D: NHWC input [img][H][Wd][C], ksize x ksize, stride, pad; M = imgs*OH*OW, K = ksize^2*C
  int H = 0, Wd = 0, Cin = 0, OH = 0, OW = 0, stride = 1, pad = 0, ksize = 1;
  // A_CONV1D: token-major input x[row][Cin_total]; K index = tap*Cin + c; rowinfo[m] = {center_row, t_center, len_in, 0}
  const int4* rowinfo = nullptr; int taps = 0; int Kvalid = 0;   // Kvalid = taps*Cin (<= K, rest zero)
  // A_STEM: input f32 video [frame][88][88]; frameinfo[frame] = {t, len}; M = frames*44*44, K = 256 (245 valid)
  const int2* frameinfo = nullptr;
  int swiglu_pairs = 0;          // fused SwiGLU epilogue: gate / up weight rows interleaved in blocks of 16 (LDS-DMA kernels)
  // split-K over blockIdx.y (decode-shaped GEMMs only): fp32 partial slabs [ksplit][M][N], summed in slab order (deterministic) -- by the
  // LAST workgroup to arrive at the output tile, inside the GEMM launch (sync != nullptr: the workspace's arrival counters, zero between
  // launches), or by a reduce launch of its own running the SAME code (sync == nullptr): same bits, the caller picks by the row count
  int ksplit = 1; float* partial = nullptr;
  // Deferred RMSNorm of the decode path: the producer of A stored T(x * gain) and, per row, partial sums of x^2 (ng_* below, or
  // launch_rms_prep); this launch multiplies its accumulator, before col_scale / col_bias, by
  //   1 / rms(x[m]) = rsqrt(sum_{t < row_sumsq_tiles} row_sumsq[t * M + m] / row_dim + row_eps)      (norm(x) W^T = that x (A W^T))
  // computed once per workgroup while its first k-tiles are in flight.  nullptr: no row factor.
  const float* row_sumsq = nullptr; int row_sumsq_tiles = 0, row_dim = 0; float row_eps = 0.f;
  // "next norm" outputs of the split-K reduce (C = the residual stream x; !out_f32, no activation, whole 64-column tiles):
  // ng_out[m][n] = T(x[m][n] * ng_gain[n]), ng_rowsq[nt * M + m] = sum over column tile nt of x[m][n]^2 (x as rounded to T; 16 columns per
  // lane in column order, then a fixed butterfly over the row's 4 lanes) -- the consumer's row_sumsq with ceil(N / 64) tiles.
  void* ng_out = nullptr; int64_t ng_ld = 0; const float* ng_gain = nullptr; float* ng_rowsq = nullptr;
  // The <= 16-row form of the decode-shaped GEMMs (decode_strip.hip) keeps a 64-column tile's sum of squares as its four 16-column STRIP sums
  // ([N / 16][M]); whoever consumes them forms (s0 + s1) + (s2 + s3) per tile -- the butterfly of the tiled producer -- then adds the tiles in order:
  // rowsq_strips: this launch WRITES ng_rowsq in that layout; row_sumsq_form = 1: this launch READS row_sumsq in it (row_sumsq_tiles tiles = 4 x as many entries)
  int rowsq_strips = 0, row_sumsq_form = 0;
  int* sync = nullptr;       // [kSyncInts] arrival counters, one per output tile (zero between launches)
  // LDS-DMA kernels: order in which an XCD's contiguous range of workgroups walks the output tiles.  0: row-major (or column-major when
  // M < N); 1: blocks of 8 M-tiles x 4 N-tiles, so the 32 tiles an XCD's CUs work on at one time share 8 A panels and 4 W panels in
  // its L2 (12 panel fetches per k-step for 64 panel reads).  Speed only, never results.  Set by the launcher.
  int tile_order = 0;
  // parity tests / yardsticks (oavsr_op_gemm): 2 = the 16-wave 256 x 256 LDS-DMA kernel whatever the dispatcher would pick.  0 in the product.
  int force_tile = 0;
  int decode_tiled = 0;      // decode-shaped launch (M <= 128): the 128-row LDS-DMA tiles at ANY row count (speed only: the tile shape never changes a bit)
};
constexpr int kSyncInts = 4096;      // GemmArgs::sync
// (mt, nt) of linear tile index bid in the blocked order above; bijective for any ntm x ntn
__host__ __device__ inline void blocked_tile(int bid, int ntm, int ntn, int& mt, int& nt) {
  constexpr int GM = 8, GN = 4;
  const int per = GM * ntn, full = ntm / GM;
  int g = bid / per;
  if (g > full) g = full;
  const int gr = g < full ? GM : ntm - full * GM;
  const int r = bid - g * per;
  int chunk = r / (gr * GN);
  if (chunk * GN + GN > ntn && ntn % GN) chunk = ntn / GN;
  const int gc = (ntn - chunk * GN) < GN ? (ntn - chunk * GN) : GN;
  const int rr = r - chunk * gr * GN;
  mt = g * GM + rr / gc;
  nt = chunk * GN + rr % gc;
}
int launch_gemm(const GemmArgs& a, hipStream_t st);
// LDS-DMA pipelined 256x128 variant (gemm2.hip); launch_gemm dispatches to it when the grid fills the chip
bool gemm_dma_eligible(const GemmArgs& a);
bool gemm_small_tile(const GemmArgs& a);       // small dense GEMM: 128 x 64 tiles, no split-K (the caller then sets no split either)
bool gemm_dma_swiglu_ok(const GemmArgs& a);
const char* gemm_dma_variant(const GemmArgs& a);
int launch_gemm_dma(const GemmArgs& a, hipStream_t st);
// Decode-shaped GEMMs of <= kStripRows rows (decode_strip.hip): a wave streams one 16-column strip x one k-slice of the weights through a
// wave-private LDS FIFO (no barrier in the k loop), the k-slices of a strip -- GemmArgs::ksplit of them, the SAME slices the tiled kernels
// split over workgroups -- are summed in slab order inside the workgroup.  Same bits as the tiled kernels, no slabs, no arrival counters.
constexpr int kStripRows = 16;
bool gemm_strip_eligible(const GemmArgs& a);
bool gemm_strip_rows(int M);                    // the decode path takes the strips for a chunk of M rows (M <= kStripRows and the developer switch is on)
bool gemm_strip_selected(const GemmArgs& a);    // launch_gemm will run the strips for this launch
int launch_gemm_strip(const GemmArgs& a, hipStream_t st);
// Strip launches that feed each other, as the phases of one launch (decode_strip.hip: decode_chain_kernel; same bits as launching them one by one)
constexpr int kStripChainMax = 6;
int launch_gemm_strip_chain(const GemmArgs* g, int n, int* ctr, int* err, hipStream_t st);
// 256 x 256 tile with the 8-wave ping-pong k loop (gemm3.hip): dense bf16, no split-K / groups; picked by launch_gemm_dma
bool gemm_8p_eligible(const GemmArgs& a);      // the dispatcher's rule (deep K)
bool gemm_8p_supported(const GemmArgs& a);     // what the kernel can compute at all
int launch_gemm_8p(const GemmArgs& a, hipStream_t st);
// PERSISTENT ping-pong kernel (gemm4.hip): one workgroup per CU walks a list of 256 x 256 tiles, the LDS-DMA stream continues across
// tiles and the C stores are not waited for.  bf16, bias + GELU / ReLU / SiLU or fused SwiGLU, no residual; picked by launch_gemm_dma
// when a launch has more tiles than the chip has CUs.  Bit-identical to the other tile variants.
bool gemm_8pp_eligible(const GemmArgs& a);     // the dispatcher's rule (supported && multi-round)
bool gemm_8pp_supported(const GemmArgs& a);
int gemm_8pp_grid(int64_t ntiles);             // workgroups the launcher uses for `ntiles` tiles
// grid: workgroups (multiple of 8; 0 = gemm_8pp_grid); rows: tile rows, 256 / 192 (0 = gemm_8pp_rows: the cheaper form for the shape; same bits)
// 1: ONE call at a time on an otherwise idle chip (oavsr_set_launch_objective): tiles chosen for a launch's own duration, not for CU-time.
// Read by the dispatcher's rules directly (process-wide), so that every eligibility query and the launch itself agree.  Speed only.
int launch_alone();
int gemm_8pp_rows(const GemmArgs& a);
double gemm_cu_share(const GemmArgs& a);     // fraction of the CUs the launch can occupy (profiling: CU-time = duration x share)
int launch_gemm_8pp(const GemmArgs& a, hipStream_t st, int grid = 0, int rows = 0);

// ---- attention ---------------------------------------------------------------------------------
struct AttnArgs {
  int dtype = DT_BF16;
  int head_dim = 64;                 // 64 or 128
  int n_heads = 0, n_kv_heads = 0;
  int n_seqs = 0;
  const void* Q = nullptr; int64_t ldq = 0;     // Q[row][head*hd + d]
  const void* K = nullptr; int64_t ldk = 0;     // K[row][kvhead*hd + d]
  const void* V = nullptr; int64_t ldv = 0;
  void* O = nullptr; int64_t ldo = 0;
  const int* cu_q = nullptr;         // [n_seqs+1] row offsets of queries (device)
  const int* cu_k = nullptr;         // [n_seqs+1] row offsets of keys
  int max_q = 0;                     // max query length (host) -> grid
  float scale = 1.f;
  int causal = 0;                    // key j visible to query i iff j <= i + (klen - qlen)
  int form = 0;                      // bf16 / head_dim 64: 32 = the 32x32x16-MFMA kernel (another summation order: fixed per call site)
};
int launch_attention(const AttnArgs& a, hipStream_t st);

// Decode attention (one new token per row).  KV cache layout: [seq][kv_head][max_ctx][hd].
// Beams: row r belongs to utterance r / beams; positions < prefix_len[utt] are read from the
// utterance's shared prefix cache (slot = utt), later positions from tail caches through `anc`.
struct DecodeAttnArgs {
  int dtype = DT_BF16;
  int head_dim = 64, n_heads = 0, n_kv_heads = 0;
  int rows = 0, beams = 1;
  const void* Q = nullptr; int64_t ldq = 0;
  void* O = nullptr; int64_t ldo = 0;
  const void* Kp = nullptr; const void* Vp = nullptr;   // prefix caches [utt][kvh][max_prefix][hd]
  int max_prefix = 0;
  const void* Kt = nullptr; const void* Vt = nullptr;   // tail caches   [row][kvh][max_tail][hd]
  int max_tail = 0;
  const int* prefix_len = nullptr;   // [utts] device
  const int* step = nullptr;         // device scalar: number of tail tokens INCLUDING the current one (nullptr: cpos[row] + 1 per row)
  const int* rowmap = nullptr;       // [rows] state slot of each activation row (decode scheduler; nullptr: identity; needs beams == 1, no anc)
  const int* anc = nullptr;          // [rows][max_tail] row that holds tail position t (null: own row)
  const float2* rope_cs = nullptr;   // [rows][hd/2] {cos, sin} of pos[row] (optional; else sincosf(pos * inv_freq) in the kernel)
  float scale = 1.f;
  // fused RoPE + cache append (inv_freq != nullptr): Q then points at the fused [q|k|v] rows (ldq = row stride); q and the
  // new k are rotated with pos[row], k/v are written to the tail cache at cpos[row] before the keys are read
  const float* inv_freq = nullptr; const int* pos = nullptr; const int* cpos = nullptr;
};
int launch_decode_attention(const DecodeAttnArgs& a, hipStream_t st);

// ---- normalisation / elementwise ----------------------------------------------------------------
int launch_layernorm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, const float* b,
                     int rows, int D, float eps, hipStream_t st);
// LayerNorm over the virtual concat [zeros(D), x[D]] -> y[2D]  (hubert.py:709-721)
int launch_layernorm_zero_prefix(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w,
                                 const float* b, int rows, int D, float eps, hipStream_t st);
int launch_rmsnorm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, int rows, int D,
                   float eps, hipStream_t st);
// decode path's deferred RMSNorm for rows no GEMM produced: y = T(x * w), sumsq[row] = sum_c x[row][c]^2  (ops.hip: rms_prep_kernel)
int launch_rms_prep(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, float* sumsq, int rows, int D, hipStream_t st);
// Fused visual stem (stem.hip): Conv3d(1->64, 5x7x7, s (1,2,2)) + BN + PReLU + MaxPool(3x3, s2) on f32 video [F][88][88];
// frameinfo[f] = {t, len}; wp = weights repacked by launch_stem_repack ([64][256] -> [64][288]); out [F][22][22][64].
int launch_stem_repack(int dtype, const void* w, void* wp, hipStream_t st);
// Pixel source of the stem.  C == 0: f32 [frame][88][88] (already transformed).  C == 1 / 3: the decoded frames, uint8 [frame][C][H][W];
// the test-time transform (datamodule/transforms.py:93-99: / 255, CenterCrop(88) at (top, left), Grayscale, Normalize(0.421, 0.165)) runs in
// the kernel's window gather with the host code's fp32 operations in the host code's order.
struct StemSrc {
  int C = 0, H = 88, W = 88, top = 0, left = 0;
  static StemSrc u8(int C_, int H_, int W_) {          // torchvision CenterCrop: int(round((H - 88) / 2.0)) (Python round: half to even)
    StemSrc s; s.C = C_; s.H = H_; s.W = W_;
    auto off = [](int n) { const int d = n - 88; return (d % 2 == 0) ? d / 2 : ((d / 2) % 2 == 0 ? d / 2 : d / 2 + 1); };
    s.top = off(H_); s.left = off(W_);
    return s;
  }
};
int launch_stem_pool(int dtype, const void* video, const int2* frameinfo, int F, const void* wp, const float* scale, const float* shift,
                     const float* slope, void* out, hipStream_t st, const StemSrc& src = StemSrc());
// Image-resident 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, NHWC bf16 (conv.hip): folded BN + residual + PReLU
// (slope == nullptr: no activation).  Bit-identical to the implicit-GEMM path.
bool conv3x3_c64_eligible(int dtype, int H, int W, int Cin, int Cout, int ks, int stride);
int launch_conv3x3_c64(const void* x, int imgs, int H, int W, const void* w, const float* scale, const float* shift, const float* slope,
                       const void* residual, void* out, hipStream_t st);
// ... and 128 -> 128 channels on small maps (layer2: 11 x 11): groups of images resident, weights streamed tap by tap.  Bit-identical to the implicit GEMM.
bool conv3x3_c128_eligible(int dtype, int H, int W, int Cin, int Cout, int ks, int stride);
int launch_conv3x3_c128(const void* x, int imgs, int H, int W, const void* w, const float* scale, const float* shift, const float* slope,
                        const void* residual, void* out, hipStream_t st);
// 3x3 stride-2 pad-1 max pool, NHWC
int launch_maxpool3x3s2(int dtype, const void* x, void* y, int imgs, int H, int W, int C, hipStream_t st);
// global average pool over HW, NHWC -> [imgs][C]
int launch_avgpool_hw(int dtype, const void* x, void* y, int imgs, int HW, int C, hipStream_t st);
// Time compressor (modeling_OmniAVSR.py:480-483 / :507-514): per sequence, out row j = mean (or concat) of
// in rows [in_off + j*rate, +rate).  seg[] = {in_off, out_off, n_out} per sequence (device).
int launch_time_compress(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const int4* seg, int n_seqs,
                         int max_out, int rate, int C, int stack, hipStream_t st);
// gather rows: y[i] = table[ids[i]]
int launch_embed(int dtype, const void* table, int64_t ldt, const int* ids, void* y, int64_t ldy, int n, int D,
                 hipStream_t st);
// copy rows with per-row source map: y[i] = src[map[i]] (map<0 -> skip)
int launch_copy_rows(int dtype, const void* src, int64_t lds, const int* map, void* y, int64_t ldy, int n, int D,
                     hipStream_t st);
// RoPE (rotate-half, HF convention) applied in place to q and k slices of a fused qkv buffer; cos/sin computed
// from inv_freq[hd/2] and pos[row].  Optionally scatters k,v rows into a KV cache [slot][kvh][max_ctx][hd].
struct RopeArgs {
  int dtype = DT_BF16; int rows = 0; int n_heads = 0, n_kv_heads = 0, head_dim = 64;
  void* qkv = nullptr; int64_t ld = 0;          // [row][ (H + 2*KV) * hd ]
  const float* inv_freq = nullptr;              // [hd/2]
  const int* pos = nullptr;                     // [rows] absolute position
  const int* slot = nullptr;                    // [rows] cache slot (sequence / row)
  const int* cpos = nullptr;                    // [rows] position inside the cache
  void* kcache = nullptr; void* vcache = nullptr; int max_ctx = 0;
};
int launch_rope_kv(const RopeArgs& a, hipStream_t st);
// y[m][j] = silu(gate[m][j]) * up[m][j]; x = [gate | up] (interleaved = 0) or 16-column blocks g0 u0 g1 u1 ... (interleaved = 1)
int launch_swiglu(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int rows, int I, int interleaved, hipStream_t st);
// argmax over fp32 logits rows (ties -> lowest index, as torch.argmax); optional banned token
// scratch: optional [rows][64] float2 for the two-stage variant
int launch_argmax(const float* logits, int64_t ld, int rows, int V, int banned, int* out, void* scratch, hipStream_t st);
// per-utterance 64-bit digest of rows [cu[b], cu[b + 1]) (cu == nullptr: row b) of `words_per_row` 32-bit words each -> out[b * out_stride + out_col]
int launch_digest(const void* base, const int* cu, int64_t words_per_row, int n_utts, uint64_t* out, int out_stride, int out_col, hipStream_t st,
                  int64_t row_stride = 0);      // row_stride (words, cu == nullptr only): rows further apart than they are long (padded logits rows)
// segment metadata
int launch_build_rowinfo(const int* cu_out, const int* cu_in, int n_seqs, int stride, int4* rowinfo, int rows,
                         hipStream_t st);

// ---- audio front-end ----------------------------------------------------------------------------
// Whisper log-mel (K10): wav f32 [B][ld] (first lens[b] samples valid, rest treated as the 30 s zero padding)
// -> out [B][3000][80] in dtype.  twiddle = {cos,sin}(2*pi*k/400) as doubles [400][2]; melfb f32 [201][80];
// mel_ws f32 [B][3000][80]; max_ws f32 [B].
int launch_logmel(int dtype, const float* wav, int64_t ld, const int* lens, int B, const double* twiddle,
                  const float* melfb, float* mel_ws, float* max_ws, void* out, int64_t ldo, hipStream_t st);

// ---- greedy bookkeeping ---------------------------------------------------------------------------
struct GreedyState {
  int* tok;        // [B] current token per row (input of the next step)
  int* finished;   // [B]
  int* out_ids;    // [B][max_new]
  int* out_len;    // [B]
  int* step;       // scalar: number of generated tokens so far
  int* pos;        // [B] absolute position of the next token
  int* cpos;       // [B] cache position of the next token
  // optional: RoPE table of the NEXT position, refreshed by greedy_update so the 16 decode-attention launches of a step read
  // (cos, sin) instead of each running sincosf:  rope_cs[b][d] = {cos, sin}(pos[b] * inv_freq[d]), d < rope_half
  float2* rope_cs = nullptr; const float* inv_freq = nullptr; int rope_half = 0;
};
int launch_greedy_update(const int* argmax, GreedyState s, int B, int max_new, int eos, int pad, hipStream_t st);

// ---- decode scheduler: continuous batching of the greedy decode (model.hip) ---------------------------------------------
// State of the scheduler's ROW SLOTS (one slot = one utterance being decoded: its KV rows, position, hypothesis so far).  A decode
// step runs on compact activation rows 0..R-1; rowmap[r] names the slot row r belongs to.  Slot `scratch_slot` backs the padding rows
// of a step (R is rounded up to a multiple of 8): it is never updated.
struct SchedState {
  int* tok = nullptr;        // [slots + 1] input token of the row's next step
  int* finished = nullptr;   // [slots + 1]
  int* out_ids = nullptr;    // [slots + 1][out_ld]
  int* out_len = nullptr;    // [slots + 1]
  int* nstep = nullptr;      // [slots + 1] tokens generated so far (the row's own step counter)
  int* limit = nullptr;      // [slots + 1] max_new of the call the row belongs to
  int* pos = nullptr;        // [slots + 1] absolute position of the next token
  int* cpos = nullptr;       // [slots + 1] tail-cache position of the next token
  float2* rope_cs = nullptr; // [slots + 1][rope_half] {cos, sin} of pos
  const float* inv_freq = nullptr; int rope_half = 0;
  int out_ld = 0, scratch_slot = 0;
};
struct IntPack128 { int v[128]; };
// dst[i] = v[i], i < n <= 128 (kernel arguments by value: no pinned staging, replay-safe ordering on the stream)
int launch_set_ints(int* dst, const IntPack128& v, int n, hipStream_t st);
// a call's rows enter the pool: slot0 + b gets pos = S[b] - 1, cpos = -1, nstep = 0, finished = 0, out_len = 0, out_ids row = pad
int launch_sched_init_slots(const SchedState& s, int slot0, int B, const IntPack128& S, int limit, int pad, hipStream_t st);
// first stage of the two-stage arg-max only (partials [rows][64] {value, index}); any V
int launch_argmax_partial(const float* logits, int64_t ld, int rows, int V, int banned, void* partials, hipStream_t st);
// One launch for the tail of a decode step, one workgroup per row: final arg-max over the row's partials -> HF greedy bookkeeping of
// the row's slot (the expressions of greedy_update_kernel with the slot's own step counter) -> RoPE table of the next position ->
// (lx != nullptr) the next step's input row lx[r] = embed[tok].  rowmap == nullptr: row r is slot slot0 + r.
int launch_sched_tail(const void* partials, const SchedState& s, const int* rowmap, int slot0, int rows, int eos, int pad, int dtype,
                      const void* embed, int64_t ldt, void* lx, int64_t ldx, int D, hipStream_t st);
// lx[r] = embed[tok[rowmap[r]]] (membership of the step changed: the input rows are regathered in the new order)
int launch_sched_embed(const SchedState& s, const int* rowmap, int rows, int dtype, const void* embed, int64_t ldt, void* lx, int64_t ldx,
                       int D, hipStream_t st);
// a call leaves the pool: out_ids[b][0..max_new) / out_len[b] <- slot slot0 + b
int launch_sched_retire(const SchedState& s, int slot0, int B, int max_new, int* out_ids, int* out_len, hipStream_t st);

// ---- beam search (beam.hip) ------------------------------------------------------------------------
struct BeamArgs {
  int B = 0, K = 0, V = 0, max_new = 0, eos = 0, pad = 0;
  int* step = nullptr;            // device scalar: tokens generated so far
  float* beam_scores = nullptr;   // [B*K]
  int* next_tok = nullptr; int* pos = nullptr; int* cpos = nullptr;   // [B*K] inputs of the next decode step
  const int* prefix_len = nullptr;   // [B]
  float* cand_val = nullptr; int* cand_tok = nullptr;   // [rows][2K] per-row top-2K (value = logp + beam score)
  int* done = nullptr; int* hyp_n = nullptr;            // [B]
  double* hyp_score = nullptr; int* hyp_len = nullptr; int* hyp_tok = nullptr;   // [B][K], [B][K], [B][K][max_new]
  int* out_ids = nullptr; int* out_len = nullptr;
  // optional: {cos, sin}(pos[row] * inv_freq[d]) of the rows' NEXT step, written with the positions (the decode-attention kernels then read the
  // table instead of each running sincosf in every layer; the same expression, hence the same bits -- as the greedy update does)
  float2* rope_cs = nullptr; const float* inv_freq = nullptr; int rope_half = 0;
};
// need_exact: rows ints of scratch (nullptr = always run the exact radix-select kernel)
int launch_beam_topk(const float* logits, int64_t ld, int rows, int V, int K, const float* beam_scores, float* cand_val, int* cand_tok,
                     int* need_exact, hipStream_t st);
int launch_beam_step(const BeamArgs& a, int first, const int* seq_old, const int* anc_old, int* seq_new, int* anc_new, hipStream_t st);
int launch_beam_finalize(const BeamArgs& a, const int* seqs, hipStream_t st);

}  // namespace oavsr
