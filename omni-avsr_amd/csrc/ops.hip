// HBM-bound kernels: normalisation, pooling, token compressor, row gathers, RoPE + KV-cache write, SwiGLU,
// arg-max and greedy bookkeeping.  All loads/stores are 16-byte vectors (4 x f32 / 8 x bf16) per lane,
// one wave (64 lanes) per row where a row reduction is needed, statistics in fp32.
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = 4;

// ---------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm: one wave per row.  MODE 0 = LayerNorm, 1 = RMSNorm, 2 = LayerNorm over [zeros(D), x]
// ---------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(kThreads) void norm_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                         const float* __restrict__ w, const float* __restrict__ b, int rows, int D,
                                                         float eps) {
  constexpr int VEC = TypeInfo<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * kWavesPerBlock + wave;
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * ldx;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane * VEC; c < D; c += 64 * VEC) {
    float f[VEC];
    chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(xr + c), f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1 += f[e]; s2 += f[e] * f[e]; }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const float n = (MODE == 2) ? 2.f * D : (float)D;
  float mean = (MODE == 1) ? 0.f : s1 / n;
  float var = s2 / n - mean * mean;
  if (MODE != 1) {
    // two-pass variance for accuracy (matches torch's Welford result to fp32 rounding)
    float v2 = 0.f;
    for (int c = lane * VEC; c < D; c += 64 * VEC) {
      float f[VEC];
      chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(xr + c), f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) { float d = f[e] - mean; v2 += d * d; }
    }
    v2 = wave_sum(v2);
    if (MODE == 2) v2 += (float)D * mean * mean;   // the D zero inputs
    var = v2 / n;
  }
  const float rstd = rsqrtf(var + eps);
  T* yr = y + (int64_t)row * ldy;
  if (MODE == 2) {
    const float z = -mean * rstd;
    for (int c = lane * VEC; c < D; c += 64 * VEC) {
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = z * w[c + e] + b[c + e];
      *reinterpret_cast<Chunk16*>(yr + c) = chunk_from_f32<T>(o);
    }
    yr += D; w += D; b += D;
  }
  for (int c = lane * VEC; c < D; c += 64 * VEC) {
    float f[VEC], o[VEC];
    chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(xr + c), f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      if (MODE == 1) o[e] = w[c + e] * (f[e] * rstd);
      else o[e] = (f[e] - mean) * rstd * w[c + e] + b[c + e];
    }
    *reinterpret_cast<Chunk16*>(yr + c) = chunk_from_f32<T>(o);
  }
}

// Same arithmetic, row held in registers: D == NCH * 64 * VEC, all NCH chunks of the row are requested at once and the three
// passes (sums, centred variance, normalise) run from registers -- one memory round trip per row instead of three dependent ones.
// Bit-identical to norm_kernel (same accumulation order).
template <typename T, int MODE, int NCH>
__global__ __launch_bounds__(kThreads) void norm_reg_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                             const float* __restrict__ w, const float* __restrict__ b, int rows, int D,
                                                             float eps) {
  constexpr int VEC = TypeInfo<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * kWavesPerBlock + wave;
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * ldx;
  Chunk16 raw[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) raw[i] = *reinterpret_cast<const Chunk16*>(xr + (lane + i * 64) * VEC);
  float f[NCH][VEC];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    chunk_to_f32<T>(raw[i], f[i]);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1 += f[i][e]; s2 += f[i][e] * f[i][e]; }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const float n = (MODE == 2) ? 2.f * D : (float)D;
  float mean = (MODE == 1) ? 0.f : s1 / n;
  float var = s2 / n - mean * mean;
  if (MODE != 1) {
    float v2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) { float d = f[i][e] - mean; v2 += d * d; }
    v2 = wave_sum(v2);
    if (MODE == 2) v2 += (float)D * mean * mean;   // the D zero inputs
    var = v2 / n;
  }
  const float rstd = rsqrtf(var + eps);
  T* yr = y + (int64_t)row * ldy;
  if (MODE == 2) {
    const float z = -mean * rstd;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = (lane + i * 64) * VEC;
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = z * w[c + e] + b[c + e];
      *reinterpret_cast<Chunk16*>(yr + c) = chunk_from_f32<T>(o);
    }
    yr += D; w += D; b += D;
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + i * 64) * VEC;
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      if (MODE == 1) o[e] = w[c + e] * (f[i][e] * rstd);
      else o[e] = (f[i][e] - mean) * rstd * w[c + e] + b[c + e];
    }
    *reinterpret_cast<Chunk16*>(yr + c) = chunk_from_f32<T>(o);
  }
}

template <typename T, int MODE>
static bool launch_norm_reg(const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, const float* b, int rows, int D, float eps,
                            hipStream_t st) {
  constexpr int VEC = TypeInfo<T>::kVec;
  dim3 grid(ceil_div(rows, kWavesPerBlock));
  const int nch = D % (64 * VEC) == 0 ? D / (64 * VEC) : 0;
  if (nch == 1) hipLaunchKernelGGL((norm_reg_kernel<T, MODE, 1>), grid, dim3(kThreads), 0, st, (const T*)x, ldx, (T*)y, ldy, w, b, rows, D, eps);
  else if (nch == 2) hipLaunchKernelGGL((norm_reg_kernel<T, MODE, 2>), grid, dim3(kThreads), 0, st, (const T*)x, ldx, (T*)y, ldy, w, b, rows, D, eps);
  else if (nch == 4) hipLaunchKernelGGL((norm_reg_kernel<T, MODE, 4>), grid, dim3(kThreads), 0, st, (const T*)x, ldx, (T*)y, ldy, w, b, rows, D, eps);
  else return false;
  return true;
}

template <int MODE>
int launch_norm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, const float* b, int rows, int D,
                float eps, hipStream_t st) {
  if (rows <= 0) return 0;
  const int vec = dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(D % vec == 0 && ldx % vec == 0 && ldy % vec == 0, "norm: D=%d and strides must be multiples of %d", D, vec);
  if (dtype == DT_F32 ? launch_norm_reg<float, MODE>(x, ldx, y, ldy, w, b, rows, D, eps, st)
                      : launch_norm_reg<bf16_t, MODE>(x, ldx, y, ldy, w, b, rows, D, eps, st)) {
    OAVSR_CHECK_HIP(hipGetLastError());
    return 0;
  }
  dim3 grid(ceil_div(rows, kWavesPerBlock));
  if (dtype == DT_F32)
    hipLaunchKernelGGL((norm_kernel<float, MODE>), grid, dim3(kThreads), 0, st, (const float*)x, ldx, (float*)y, ldy, w, b, rows, D, eps);
  else
    hipLaunchKernelGGL((norm_kernel<bf16_t, MODE>), grid, dim3(kThreads), 0, st, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, w, b, rows, D, eps);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

// Deferred RMSNorm of the decode path (Llama_LoRA.py:587-655: input_layernorm / post_attention_layernorm / model.norm): the consumer GEMM
// takes y = T(x * gain) as its A operand and multiplies its accumulator by 1 / rms(x), which it derives from the row's sum of squares
// (GemmArgs::row_sumsq).  This kernel prepares the pair for rows that no GEMM produced (the embedding rows at the start of a decode step);
// inside a step the split-K reduce of o_proj / down_proj writes it (gemm2.hip).  One wave per row; sum per lane in column order, then wave_sum.
template <typename T>
__global__ __launch_bounds__(kThreads) void rms_prep_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                             const float* __restrict__ w, float* __restrict__ sumsq, int rows, int D) {
  constexpr int VEC = TypeInfo<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * kWavesPerBlock + wave;
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * ldx;
  T* yr = y + (int64_t)row * ldy;
  float s2 = 0.f;
  for (int c = lane * VEC; c < D; c += 64 * VEC) {
    float f[VEC], o[VEC];
    chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(xr + c), f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s2 = fmaf(f[e], f[e], s2); o[e] = f[e] * w[c + e]; }
    *reinterpret_cast<Chunk16*>(yr + c) = chunk_from_f32<T>(o);
  }
  s2 = wave_sum(s2);
  if (lane == 0) sumsq[row] = s2;
}

// ---------------------------------------------------------------------------------------------
// NHWC pools
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int imgs, int H, int W, int C) {
  constexpr int VEC = TypeInfo<T>::kVec;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1, CV = C / VEC;
  const int64_t total = (int64_t)imgs * OH * OW * CV;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    int cv = (int)(i % CV);
    int64_t r = i / CV;
    int ox = (int)(r % OW); r /= OW;
    int oy = (int)(r % OH);
    int img = (int)(r / OH);
    float m[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        float f[VEC];
        chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(x + (((int64_t)img * H + iy) * W + ix) * C + cv * VEC), f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], f[e]);
      }
    }
    *reinterpret_cast<Chunk16*>(y + (((int64_t)img * OH + oy) * OW + ox) * C + cv * VEC) = chunk_from_f32<T>(m);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void avgpool_hw_kernel(const T* __restrict__ x, T* __restrict__ y, int imgs, int HW, int C) {
  constexpr int VEC = TypeInfo<T>::kVec;
  const int CV = C / VEC;
  const int64_t total = (int64_t)imgs * CV;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    int cv = (int)(i % CV);
    int img = (int)(i / CV);
    float a[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[e] = 0.f;
    for (int p = 0; p < HW; ++p) {
      float f[VEC];
      chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(x + ((int64_t)img * HW + p) * C + cv * VEC), f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) a[e] += f[e];
    }
    const float inv = 1.f / HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[e] *= inv;
    *reinterpret_cast<Chunk16*>(y + (int64_t)img * C + cv * VEC) = chunk_from_f32<T>(a);
  }
}

// ---------------------------------------------------------------------------------------------
// Matryoshka time compressor: avg-pool (kernel = stride = rate, floor) or frame stacking.
// grid = (max_out, n_seqs); one workgroup per output token, lanes stride the channel chunks.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void time_compress_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                                  const int4* __restrict__ seg, int rate, int C, int stack) {
  constexpr int VEC = TypeInfo<T>::kVec;
  const int4 s = seg[blockIdx.y];
  const int j = blockIdx.x;
  if (j >= s.z) return;
  const T* xr = x + (int64_t)(s.x + j * rate) * ldx;
  T* yr = y + (int64_t)(s.y + j) * ldy;
  const int CV = C / VEC;
  if (stack) {
    for (int i = threadIdx.x; i < rate * CV; i += kThreads) {
      int r = i / CV, cv = i - r * CV;
      *reinterpret_cast<Chunk16*>(yr + (int64_t)r * C + cv * VEC) = *reinterpret_cast<const Chunk16*>(xr + (int64_t)r * ldx + cv * VEC);
    }
  } else {
    const float inv = 1.f / rate;
    for (int cv = threadIdx.x; cv < CV; cv += kThreads) {
      float a[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) a[e] = 0.f;
      for (int r = 0; r < rate; ++r) {
        float f[VEC];
        chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(xr + (int64_t)r * ldx + cv * VEC), f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] += f[e];
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) a[e] *= inv;
      *reinterpret_cast<Chunk16*>(yr + cv * VEC) = chunk_from_f32<T>(a);
    }
  }
}

// y[i] = src[map[i]]  (map < 0: leave untouched).  One wave per row.
template <typename T>
__global__ __launch_bounds__(kThreads) void copy_rows_kernel(const T* __restrict__ src, int64_t lds, const int* __restrict__ map,
                                                              T* __restrict__ y, int64_t ldy, int n, int D) {
  constexpr int VEC = TypeInfo<T>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * kWavesPerBlock + wave;
  if (row >= n) return;
  const int s = map[row];
  if (s < 0) return;
  for (int c = lane * VEC; c < D; c += 64 * VEC)
    *reinterpret_cast<Chunk16*>(y + (int64_t)row * ldy + c) = *reinterpret_cast<const Chunk16*>(src + (int64_t)s * lds + c);
}

// ---------------------------------------------------------------------------------------------
// RoPE (HF rotate_half convention) in place on the q and k slices of a fused [q|k|v] row, then scatter of
// the rotated k and of v into the KV cache.  grid = rows, block = 256.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void rope_kv_kernel(const RopeArgs p) {
  const int row = blockIdx.x;
  const int HD = p.head_dim, half = HD / 2;
  T* base = reinterpret_cast<T*>(p.qkv) + (int64_t)row * p.ld;
  const float pos = (float)p.pos[row];
  const int nrot = (p.n_heads + p.n_kv_heads) * half;
  for (int i = threadIdx.x; i < nrot; i += kThreads) {
    int h = i / half, d = i - h * half;
    float ang = pos * p.inv_freq[d];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    T* v = base + (int64_t)h * HD;
    float x0 = to_f32(v[d]), x1 = to_f32(v[d + half]);
    v[d] = from_f32<T>(x0 * cs - x1 * sn);
    v[d + half] = from_f32<T>(x1 * cs + x0 * sn);
  }
  if (p.kcache == nullptr) return;
  __syncthreads();
  const int slot = p.slot[row], cp = p.cpos[row];
  const T* k = base + (int64_t)p.n_heads * HD;
  const T* v = k + (int64_t)p.n_kv_heads * HD;
  T* kc = reinterpret_cast<T*>(p.kcache);
  T* vc = reinterpret_cast<T*>(p.vcache);
  for (int i = threadIdx.x; i < p.n_kv_heads * HD; i += kThreads) {
    int h = i / HD, d = i - h * HD;
    int64_t o = (((int64_t)slot * p.n_kv_heads + h) * p.max_ctx + cp) * HD + d;
    kc[o] = k[i];
    vc[o] = v[i];
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void swiglu_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int rows, int I, int interleaved) {
  constexpr int VEC = TypeInfo<T>::kVec;
  const int IV = I / VEC;
  const int64_t total = (int64_t)rows * IV;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    int cv = (int)(i % IV);
    int64_t r = i / IV;
    float g[VEC], u[VEC];
    const int c0 = cv * VEC;
    const int go = interleaved ? ((c0 >> 4) << 5) + (c0 & 15) : c0;
    const int uo = interleaved ? go + 16 : I + c0;
    chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(x + r * ldx + go), g);
    chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(x + r * ldx + uo), u);
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] = silu(g[e]) * u[e];
    *reinterpret_cast<Chunk16*>(y + r * ldy + cv * VEC) = chunk_from_f32<T>(g);
  }
}

// arg-max over fp32 logits; ties resolve to the lowest index (torch.argmax on CPU / HF greedy).
// Two stages so a [B x 128k] logits block fills the chip: grid (kArgChunks, rows) partial (value, index) pairs, then one
// wave per row combines them.  scratch == nullptr: single-stage (one workgroup per row).
constexpr int kArgChunks = 64;

__device__ __forceinline__ void argmax_combine(float& best, int& bi, float ov, int oi) {
  if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
}

__global__ __launch_bounds__(kThreads) void argmax_kernel(const float* __restrict__ logits, int64_t ld, int V, int banned,
                                                          int* __restrict__ out, float2* __restrict__ scratch) {
  __shared__ float sv[kWavesPerBlock];
  __shared__ int si[kWavesPerBlock];
  const int row = blockIdx.y;
  const float* r = logits + (int64_t)row * ld;
  int lo = 0, hi = V;
  if (scratch) {
    const int ch = ((V + kArgChunks - 1) / kArgChunks + 3) & ~3;
    lo = blockIdx.x * ch; hi = lo + ch < V ? lo + ch : V;
  }
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lo + threadIdx.x; i < hi; i += kThreads) {
    float v = r[i];
    if (i == banned) v = -INFINITY;
    argmax_combine(best, bi, v, i);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    argmax_combine(best, bi, ov, oi);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sv[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kWavesPerBlock; ++w) argmax_combine(best, bi, sv[w], si[w]);
    if (scratch) scratch[row * kArgChunks + blockIdx.x] = make_float2(best, __int_as_float(bi));
    else out[row] = bi == 0x7fffffff ? 0 : bi;     // all-NaN row: no comparison ever succeeds; keep the index inside the vocabulary
  }
}

__global__ __launch_bounds__(64) void argmax_final_kernel(const float2* __restrict__ scratch, int* __restrict__ out) {
  const int row = blockIdx.x;
  float2 p = scratch[row * kArgChunks + threadIdx.x];
  float best = p.x;
  int bi = __float_as_int(p.y);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    argmax_combine(best, bi, ov, oi);
  }
  if (threadIdx.x == 0) out[row] = bi == 0x7fffffff ? 0 : bi;   // all-NaN row (see argmax_kernel)
}

// Greedy bookkeeping, one thread per row (HF GenerationMixin greedy: finished rows emit pad).
__global__ void greedy_update_kernel(const int* __restrict__ amax, GreedyState s, int B, int max_new, int eos, int pad) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int step = *s.step;
  if (b < B) {
    int tok = amax[b];
    if (s.finished[b]) tok = pad;
    else {
      s.out_len[b] = step + 1;
      if (tok == eos) s.finished[b] = 1;
    }
    if (step < max_new) s.out_ids[(int64_t)b * max_new + step] = tok;
    s.tok[b] = tok;
    s.pos[b] += 1;
    s.cpos[b] += 1;
  }
  // all rows read `step` before anyone bumps it: single block launch guarantees this via the barrier below
  __syncthreads();
  if (b == 0) *s.step = step + 1;
  if (s.rope_cs) {                                   // same expression as the attention kernel's inline path => same bits
    for (int i = threadIdx.x; i < B * s.rope_half; i += blockDim.x) {
      const int r = i / s.rope_half, d = i - r * s.rope_half;
      float sn, cs;
      sincosf((float)s.pos[r] * s.inv_freq[d], &sn, &cs);
      s.rope_cs[i] = make_float2(cs, sn);
    }
  }
}

// ---- decode scheduler (kernels.h: SchedState) ------------------------------------------------------------------------------
__global__ void set_ints_kernel(int* __restrict__ dst, IntPack128 v, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = v.v[threadIdx.x];
}

__global__ void sched_init_slots_kernel(SchedState s, int slot0, int B, IntPack128 S, int limit, int pad) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int slot = slot0 + b;
  if (threadIdx.x == 0) {
    s.pos[slot] = S.v[b] - 1; s.cpos[slot] = -1; s.nstep[slot] = 0; s.finished[slot] = 0; s.out_len[slot] = 0; s.limit[slot] = limit; s.tok[slot] = pad;
  }
  for (int i = threadIdx.x; i < s.out_ld; i += blockDim.x) s.out_ids[(int64_t)slot * s.out_ld + i] = pad;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void sched_tail_kernel(const float2* __restrict__ partials, SchedState s, const int* __restrict__ rowmap, int slot0,
                                                              int eos, int pad, const T* __restrict__ embed, int64_t ldt, T* __restrict__ lx, int64_t ldx, int D) {
  __shared__ int sh_tok, sh_pos;
  const int r = blockIdx.x;
  const int slot = rowmap ? rowmap[r] : slot0 + r;
  const bool live = slot != s.scratch_slot;
  if (threadIdx.x < 64) {                               // argmax_final_kernel's combine, then greedy_update_kernel's bookkeeping
    float2 p = partials[r * kArgChunks + threadIdx.x];
    float best = p.x;
    int bi = __float_as_int(p.y);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o, 64);
      int oi = __shfl_xor(bi, o, 64);
      argmax_combine(best, bi, ov, oi);
    }
    if (threadIdx.x == 0) {
      int tok = bi == 0x7fffffff ? 0 : bi;
      int pos = 0;
      if (live) {
        const int step = s.nstep[slot];
        if (s.finished[slot]) tok = pad;
        else {
          s.out_len[slot] = step + 1;
          if (tok == eos) s.finished[slot] = 1;
        }
        if (step < s.limit[slot]) s.out_ids[(int64_t)slot * s.out_ld + step] = tok;
        s.tok[slot] = tok;
        pos = s.pos[slot] + 1;
        s.pos[slot] = pos;
        s.cpos[slot] += 1;
        s.nstep[slot] = step + 1;
      } else {
        tok = s.tok[slot];
      }
      sh_tok = tok; sh_pos = pos;
    }
  }
  __syncthreads();
  if (live)                                             // same expression as greedy_update_kernel / the attention kernel's inline path
    for (int d = threadIdx.x; d < s.rope_half; d += kThreads) {
      float sn, cs;
      sincosf((float)sh_pos * s.inv_freq[d], &sn, &cs);
      s.rope_cs[(int64_t)slot * s.rope_half + d] = make_float2(cs, sn);
    }
  if (lx) {
    constexpr int VEC = TypeInfo<T>::kVec;
    const T* src = embed + (int64_t)sh_tok * ldt;
    T* dst = lx + (int64_t)r * ldx;
    for (int i = threadIdx.x; i < D / VEC; i += kThreads) *reinterpret_cast<Chunk16*>(dst + i * VEC) = *reinterpret_cast<const Chunk16*>(src + i * VEC);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void sched_embed_kernel(const int* __restrict__ tok, const int* __restrict__ rowmap, const T* __restrict__ embed,
                                                               int64_t ldt, T* __restrict__ lx, int64_t ldx, int D) {
  constexpr int VEC = TypeInfo<T>::kVec;
  const int r = blockIdx.x;
  const T* src = embed + (int64_t)tok[rowmap[r]] * ldt;
  T* dst = lx + (int64_t)r * ldx;
  for (int i = threadIdx.x; i < D / VEC; i += kThreads) *reinterpret_cast<Chunk16*>(dst + i * VEC) = *reinterpret_cast<const Chunk16*>(src + i * VEC);
}

__global__ void sched_retire_kernel(SchedState s, int slot0, int max_new, int* __restrict__ out_ids, int* __restrict__ out_len) {
  const int b = blockIdx.x, slot = slot0 + b;
  for (int i = threadIdx.x; i < max_new; i += blockDim.x) out_ids[(int64_t)b * max_new + i] = s.out_ids[(int64_t)slot * s.out_ld + i];
  if (threadIdx.x == 0) out_len[b] = s.out_len[slot];
}

// Per-utterance 64-bit digest of a device region (tests / bench.py: does every call see ITS inputs?): sum over the region's 32-bit words w_i of
// splitmix64(w_i | (i + 1) << 32) mod 2^64 -- position-sensitive, order of summation irrelevant (integer adds), so any reduction tree gives
// the same value.  Utterance b owns rows [cu[b], cu[b + 1]) of `words_per_row` words (cu == nullptr: row b alone).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 27; z *= 0x94d049bb133111ebull; z ^= z >> 31;
  return z;
}
__global__ __launch_bounds__(kThreads) void digest_kernel(const unsigned* __restrict__ base, const int* __restrict__ cu, long long words_per_row,
                                                           unsigned long long* __restrict__ out, int out_stride, int out_col, long long row_stride) {
  __shared__ unsigned long long part[kWavesPerBlock];
  const int b = blockIdx.x;
  const long long r0 = cu ? cu[b] : b, r1 = cu ? cu[b + 1] : b + 1;
  const unsigned* p = base + r0 * row_stride;      // (row_stride != words_per_row only with cu == nullptr: one padded row per utterance)
  const long long n = (r1 - r0) * words_per_row;
  // blockIdx.y owns one contiguous share of the utterance's words; the shares meet in an integer atomic add (any order: same value)
  const long long per = (n + gridDim.y - 1) / gridDim.y, i0 = blockIdx.y * per, i1 = i0 + per < n ? i0 + per : n;
  unsigned long long acc = 0;
  for (long long i = i0 + threadIdx.x; i < i1; i += kThreads) acc += mix64((unsigned long long)p[i] | ((unsigned long long)(i + 1) << 32));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)acc, o, 64), hi = __shfl_xor((unsigned)(acc >> 32), o, 64);
    acc += ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += part[w];
    atomicAdd(out + (long long)b * out_stride + out_col, t);
  }
}

__global__ void rowinfo_kernel(const int* __restrict__ cu_out, const int* __restrict__ cu_in, int n_seqs, int stride, int4* __restrict__ ri, int rows) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= rows) return;
  int s = 0;
  while (s + 1 < n_seqs && cu_out[s + 1] <= m) ++s;
  int t = m - cu_out[s];
  int len_in = cu_in[s + 1] - cu_in[s];
  ri[m] = make_int4(cu_in[s] + t * stride, t * stride, len_in, s);
}

}  // namespace

int launch_layernorm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, const float* b, int rows, int D,
                     float eps, hipStream_t st) {
  return launch_norm<0>(dtype, x, ldx, y, ldy, w, b, rows, D, eps, st);
}
int launch_layernorm_zero_prefix(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, const float* b,
                                 int rows, int D, float eps, hipStream_t st) {
  return launch_norm<2>(dtype, x, ldx, y, ldy, w, b, rows, D, eps, st);
}
int launch_rmsnorm(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, int rows, int D, float eps,
                   hipStream_t st) {
  return launch_norm<1>(dtype, x, ldx, y, ldy, w, nullptr, rows, D, eps, st);
}

int launch_rms_prep(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const float* w, float* sumsq, int rows, int D, hipStream_t st) {
  if (rows <= 0) return 0;
  const int vec = dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(D % vec == 0 && ldx % vec == 0 && ldy % vec == 0, "rms_prep: D=%d must be a multiple of %d", D, vec);
  dim3 grid(ceil_div(rows, kWavesPerBlock));
  if (dtype == DT_F32) hipLaunchKernelGGL(rms_prep_kernel<float>, grid, dim3(kThreads), 0, st, (const float*)x, ldx, (float*)y, ldy, w, sumsq, rows, D);
  else hipLaunchKernelGGL(rms_prep_kernel<bf16_t>, grid, dim3(kThreads), 0, st, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, w, sumsq, rows, D);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

static int grid_for(int64_t total) {
  int64_t g = (total + kThreads - 1) / kThreads;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

int launch_maxpool3x3s2(int dtype, const void* x, void* y, int imgs, int H, int W, int C, hipStream_t st) {
  if (imgs <= 0) return 0;
  const int vec = dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(C % vec == 0, "maxpool: C=%d", C);
  int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  int g = grid_for((int64_t)imgs * OH * OW * (C / vec));
  if (dtype == DT_F32) hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(g), dim3(kThreads), 0, st, (const float*)x, (float*)y, imgs, H, W, C);
  else hipLaunchKernelGGL(maxpool3x3s2_kernel<bf16_t>, dim3(g), dim3(kThreads), 0, st, (const bf16_t*)x, (bf16_t*)y, imgs, H, W, C);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_avgpool_hw(int dtype, const void* x, void* y, int imgs, int HW, int C, hipStream_t st) {
  if (imgs <= 0) return 0;
  const int vec = dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(C % vec == 0, "avgpool: C=%d", C);
  int g = grid_for((int64_t)imgs * (C / vec));
  if (dtype == DT_F32) hipLaunchKernelGGL(avgpool_hw_kernel<float>, dim3(g), dim3(kThreads), 0, st, (const float*)x, (float*)y, imgs, HW, C);
  else hipLaunchKernelGGL(avgpool_hw_kernel<bf16_t>, dim3(g), dim3(kThreads), 0, st, (const bf16_t*)x, (bf16_t*)y, imgs, HW, C);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_time_compress(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, const int4* seg, int n_seqs, int max_out,
                         int rate, int C, int stack, hipStream_t st) {
  if (n_seqs <= 0 || max_out <= 0) return 0;
  const int vec = dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(C % vec == 0 && ldx % vec == 0 && ldy % vec == 0 && rate >= 1, "time_compress: bad geometry");
  dim3 grid(max_out, n_seqs);
  if (dtype == DT_F32) hipLaunchKernelGGL(time_compress_kernel<float>, grid, dim3(kThreads), 0, st, (const float*)x, ldx, (float*)y, ldy, seg, rate, C, stack);
  else hipLaunchKernelGGL(time_compress_kernel<bf16_t>, grid, dim3(kThreads), 0, st, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, seg, rate, C, stack);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_copy_rows(int dtype, const void* src, int64_t lds, const int* map, void* y, int64_t ldy, int n, int D, hipStream_t st) {
  if (n <= 0) return 0;
  const int vec = dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(D % vec == 0 && lds % vec == 0 && ldy % vec == 0, "copy_rows: D=%d", D);
  dim3 grid(ceil_div(n, kWavesPerBlock));
  if (dtype == DT_F32) hipLaunchKernelGGL(copy_rows_kernel<float>, grid, dim3(kThreads), 0, st, (const float*)src, lds, map, (float*)y, ldy, n, D);
  else hipLaunchKernelGGL(copy_rows_kernel<bf16_t>, grid, dim3(kThreads), 0, st, (const bf16_t*)src, lds, map, (bf16_t*)y, ldy, n, D);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_embed(int dtype, const void* table, int64_t ldt, const int* ids, void* y, int64_t ldy, int n, int D, hipStream_t st) {
  return launch_copy_rows(dtype, table, ldt, ids, y, ldy, n, D, st);
}

int launch_rope_kv(const RopeArgs& a, hipStream_t st) {
  if (a.rows <= 0) return 0;
  if (a.dtype == DT_F32) hipLaunchKernelGGL(rope_kv_kernel<float>, dim3(a.rows), dim3(kThreads), 0, st, a);
  else hipLaunchKernelGGL(rope_kv_kernel<bf16_t>, dim3(a.rows), dim3(kThreads), 0, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_swiglu(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int rows, int I, int interleaved, hipStream_t st) {
  if (rows <= 0) return 0;
  const int vec = dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(I % 16 == 0 && ldx % vec == 0 && ldy % vec == 0, "swiglu: I=%d", I);
  int g = grid_for((int64_t)rows * (I / vec));
  if (dtype == DT_F32) hipLaunchKernelGGL(swiglu_kernel<float>, dim3(g), dim3(kThreads), 0, st, (const float*)x, ldx, (float*)y, ldy, rows, I, interleaved);
  else hipLaunchKernelGGL(swiglu_kernel<bf16_t>, dim3(g), dim3(kThreads), 0, st, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, I, interleaved);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_argmax(const float* logits, int64_t ld, int rows, int V, int banned, int* out, void* scratch, hipStream_t st) {
  if (rows <= 0) return 0;
  if (scratch && V >= 4096) {
    hipLaunchKernelGGL(argmax_kernel, dim3(kArgChunks, rows), dim3(kThreads), 0, st, logits, ld, V, banned, out, (float2*)scratch);
    hipLaunchKernelGGL(argmax_final_kernel, dim3(rows), dim3(64), 0, st, (const float2*)scratch, out);
  } else {
    hipLaunchKernelGGL(argmax_kernel, dim3(1, rows), dim3(kThreads), 0, st, logits, ld, V, banned, out, (float2*)nullptr);
  }
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_argmax_partial(const float* logits, int64_t ld, int rows, int V, int banned, void* partials, hipStream_t st) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(argmax_kernel, dim3(kArgChunks, rows), dim3(kThreads), 0, st, logits, ld, V, banned, (int*)nullptr, (float2*)partials);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_set_ints(int* dst, const IntPack128& v, int n, hipStream_t st) {
  OAVSR_REQUIRE(n >= 0 && n <= 128, "set_ints: n=%d", n);
  if (n == 0) return 0;
  hipLaunchKernelGGL(set_ints_kernel, dim3(1), dim3(128), 0, st, dst, v, n);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_sched_init_slots(const SchedState& s, int slot0, int B, const IntPack128& S, int limit, int pad, hipStream_t st) {
  OAVSR_REQUIRE(B >= 1 && B <= 128, "sched_init_slots: B=%d", B);      // one call's rows: up to a whole 128-row step (IntPack128)
  hipLaunchKernelGGL(sched_init_slots_kernel, dim3(B), dim3(64), 0, st, s, slot0, B, S, limit, pad);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_sched_tail(const void* partials, const SchedState& s, const int* rowmap, int slot0, int rows, int eos, int pad, int dtype,
                      const void* embed, int64_t ldt, void* lx, int64_t ldx, int D, hipStream_t st) {
  if (rows <= 0) return 0;
  const int vec = dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(!lx || (D % vec == 0 && ldt % vec == 0 && ldx % vec == 0), "sched_tail: D=%d", D);
  if (dtype == DT_F32) hipLaunchKernelGGL(sched_tail_kernel<float>, dim3(rows), dim3(kThreads), 0, st, (const float2*)partials, s, rowmap, slot0, eos, pad, (const float*)embed, ldt, (float*)lx, ldx, D);
  else hipLaunchKernelGGL(sched_tail_kernel<bf16_t>, dim3(rows), dim3(kThreads), 0, st, (const float2*)partials, s, rowmap, slot0, eos, pad, (const bf16_t*)embed, ldt, (bf16_t*)lx, ldx, D);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_sched_embed(const SchedState& s, const int* rowmap, int rows, int dtype, const void* embed, int64_t ldt, void* lx, int64_t ldx, int D,
                       hipStream_t st) {
  if (rows <= 0) return 0;
  const int vec = dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(D % vec == 0 && ldt % vec == 0 && ldx % vec == 0, "sched_embed: D=%d", D);
  if (dtype == DT_F32) hipLaunchKernelGGL(sched_embed_kernel<float>, dim3(rows), dim3(kThreads), 0, st, s.tok, rowmap, (const float*)embed, ldt, (float*)lx, ldx, D);
  else hipLaunchKernelGGL(sched_embed_kernel<bf16_t>, dim3(rows), dim3(kThreads), 0, st, s.tok, rowmap, (const bf16_t*)embed, ldt, (bf16_t*)lx, ldx, D);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_sched_retire(const SchedState& s, int slot0, int B, int max_new, int* out_ids, int* out_len, hipStream_t st) {
  if (B <= 0) return 0;
  OAVSR_REQUIRE(max_new <= s.out_ld, "sched_retire: max_new=%d exceeds the pool's row (%d)", max_new, s.out_ld);
  hipLaunchKernelGGL(sched_retire_kernel, dim3(B), dim3(64), 0, st, s, slot0, max_new, out_ids, out_len);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_greedy_update(const int* amax, GreedyState s, int B, int max_new, int eos, int pad, hipStream_t st) {
  OAVSR_REQUIRE(B <= 1024, "greedy_update: B=%d > 1024", B);
  int threads = ((B + 63) / 64) * 64;
  if (s.rope_cs && threads < 256) threads = 256;
  hipLaunchKernelGGL(greedy_update_kernel, dim3(1), dim3(threads), 0, st, amax, s, B, max_new, eos, pad);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_digest(const void* base, const int* cu, int64_t words_per_row, int n_utts, uint64_t* out, int out_stride, int out_col, hipStream_t st, int64_t row_stride) {
  if (n_utts <= 0) return 0;
  // the utterances' slots of this column start from zero (a strided 8-byte clear), then 64 shares per utterance add into them
  OAVSR_CHECK_HIP(hipMemset2DAsync(out + out_col, (size_t)out_stride * 8, 0, 8, (size_t)n_utts, st));
  hipLaunchKernelGGL(digest_kernel, dim3(n_utts, 64), dim3(kThreads), 0, st, (const unsigned*)base, cu, (long long)words_per_row, (unsigned long long*)out, out_stride, out_col,
                     (long long)(row_stride > 0 ? row_stride : words_per_row));
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_build_rowinfo(const int* cu_out, const int* cu_in, int n_seqs, int stride, int4* rowinfo, int rows, hipStream_t st) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(rowinfo_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, st, cu_out, cu_in, n_seqs, stride, rowinfo, rows);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace oavsr
