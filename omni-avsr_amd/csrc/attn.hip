// Attention kernels for gfx950.
//
// (1) attn_varlen_kernel: flash-style attention over variable-length sequences (cu_seqlens), used for the
//     Whisper / AV-HuBERT encoders (bidirectional, multihead_attention.py:511-654) and the LLM prefill
//     (causal GQA, Llama_LoRA.py:284-307).  No padded rows are ever attended: each utterance only sees its
//     own keys, which is what makes a batch equal the reference's batch-1 runs.
//     Structure: workgroup = 4 waves = 64 query rows of one (sequence, head); K/V tiles of 64 keys staged in
//     LDS (16-B chunks, XOR-swizzled).  Each wave computes S^T = K.Q^T with the 16x16 MFMA so that a lane owns
//     one query column (lane & 15) -> softmax statistics are lane-local + two xor-shuffles, and the S^T
//     accumulator registers ARE the B operand of the second product O^T = V^T.P^T (no LDS round trip for P).
//     bf16: V^T fragments come from row-major V in LDS through ds_read_b64_tr_b16 (hardware transpose).
//     f32 : exact-f32 MFMA 16x16x4, V read with ds_read_b32.
// (2) decode_attn_kernel: one new token per row against a KV cache (shared utterance prefix + per-row tail
//     addressed through a beam ancestry table).  HBM-bound, VALU only.
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kThreads = 256;
constexpr int KT = 64;   // keys per LDS tile

template <typename T> struct MmaT;
template <> struct MmaT<bf16_t> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
  }
};
template <> struct MmaT<float> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    f32x4 fa = __builtin_bit_cast(f32x4, a.v), fb = __builtin_bit_cast(f32x4, b.v);
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb[s], c, 0, 0, 0);
  }
};

template <typename T, int HD>
__global__ __launch_bounds__(kThreads) void attn_varlen_kernel(const AttnArgs p) {
  constexpr int RB = HD * (int)sizeof(T);        // bytes per K/V row
  constexpr int NCH = RB / 16;                    // 16-B chunks per row
  constexpr int SWZ = (RB == 128) ? 7 : 15;
  constexpr int NBLK = RB / 64;                   // 64-byte k blocks along head_dim
  constexpr int NDT = HD / 16;                    // 16-wide output d tiles
  constexpr int NLD = KT * NCH / kThreads;        // chunks per thread per tile per tensor
  __shared__ __attribute__((aligned(16))) char sK[KT * RB];
  __shared__ __attribute__((aligned(16))) char sV[KT * RB];

  const int seq = blockIdx.z, head = blockIdx.y;
  const int q0 = p.cu_q[seq], qlen = p.cu_q[seq + 1] - q0;
  const int k0 = p.cu_k[seq], klen = p.cu_k[seq + 1] - k0;
  const int qt0 = blockIdx.x * 64;
  if (qt0 >= qlen) return;
  const int kvh = head / (p.n_heads / p.n_kv_heads);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;
  const int qrow = qt0 + wave * 16 + qi;                 // query index inside the sequence (may be >= qlen)
  const int qrow_c = qrow < qlen ? qrow : qlen - 1;
  const T* Q = reinterpret_cast<const T*>(p.Q);
  const T* K = reinterpret_cast<const T*>(p.K);
  const T* V = reinterpret_cast<const T*>(p.V);

  Chunk16 qf[NBLK];
  {
    const char* qp = reinterpret_cast<const char*>(Q + (int64_t)(q0 + qrow_c) * p.ldq + (int64_t)head * HD);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) qf[b] = *reinterpret_cast<const Chunk16*>(qp + b * 64 + g * 16);
  }
  const int shift = klen - qlen;                         // causal: key j visible iff j <= qrow + shift
  int kend = klen;
  if (p.causal) {
    int last = qt0 + 63 < qlen ? qt0 + 63 : qlen - 1;
    kend = last + shift + 1;
    if (kend > klen) kend = klen;
    if (kend < 0) kend = 0;
  }
  const float sc = p.scale * 1.44269504088896340736f;    // softmax in base 2

  f32x4 o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (int kt0 = 0; kt0 < kend; kt0 += KT) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      int id = tid + j * kThreads, row = id / NCH, c = id - row * NCH;
      int key = kt0 + row;
      Chunk16 ck, cv;
      if (key < klen) {
        ck = *reinterpret_cast<const Chunk16*>(reinterpret_cast<const char*>(K + (int64_t)(k0 + key) * p.ldk + (int64_t)kvh * HD) + c * 16);
        cv = *reinterpret_cast<const Chunk16*>(reinterpret_cast<const char*>(V + (int64_t)(k0 + key) * p.ldv + (int64_t)kvh * HD) + c * 16);
      } else {
        ck.v = u32x4{0u, 0u, 0u, 0u}; cv.v = ck.v;
      }
      int off = row * RB + ((c ^ (row & SWZ)) << 4);
      *reinterpret_cast<Chunk16*>(sK + off) = ck;
      *reinterpret_cast<Chunk16*>(sV + off) = cv;
    }
    __syncthreads();

    // ---- S^T[key][q] for 4 sub-tiles of 16 keys ----
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = kt * 16 + qi;                       // A-operand row = key (lane & 15)
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        Chunk16 kf = *reinterpret_cast<const Chunk16*>(sK + row * RB + (((b * 4 + g) ^ (row & SWZ)) << 4));
        MmaT<T>::run(kf, qf[b], s[kt]);
      }
    }
    // ---- mask + online softmax (lane owns query qi; its keys: kt*16 + 4g + r) ----
    float mt = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = kt0 + kt * 16 + 4 * g + r;
        bool vis = key < klen && (!p.causal || key <= qrow + shift);
        float v = vis ? s[kt][r] * sc : -INFINITY;
        s[kt][r] = v;
        mt = fmaxf(mt, v);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_use);
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = exp2f(s[kt][r] - m_use);
        s[kt][r] = e;
        lsum += e;
      }
    l_run = l_run * alpha + lsum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[i] *= alpha;

    // ---- O^T[d][q] += V^T[d][key] . P^T[key][q] ----
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {                       // 32 keys per MFMA: sub-tiles 2u, 2u+1
        bf16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[r] = (bf16_t)s[2 * u][r]; pf[4 + r] = (bf16_t)s[2 * u + 1][r]; }
        const int qq = qi >> 2, pp = qi & 3;             // tr-read: lane 4q+p addresses row q, cols 4p..4p+3
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const int c = dt * 2 + (pp >> 1);
          const int row0 = (2 * u) * 16 + 4 * g + qq, row1 = row0 + 16;
          auto a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(sV + row0 * RB + ((c ^ (row0 & SWZ)) << 4) + (pp & 1) * 8));
          auto a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(sV + row1 * RB + ((c ^ (row1 & SWZ)) << 4) + (pp & 1) * 8));
          bf16x4 v0 = __builtin_bit_cast(bf16x4, a0), v1 = __builtin_bit_cast(bf16x4, a1);
          bf16x8 vf;
#pragma unroll
          for (int r = 0; r < 4; ++r) { vf[r] = v0[r]; vf[4 + r] = v1[r]; }
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const int c = dt * 4 + (qi >> 2);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = kt * 16 + 4 * g + r;
            float vv = *reinterpret_cast<const float*>(sV + row * RB + ((c ^ (row & SWZ)) << 4) + (qi & 3) * 4);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, s[kt][r], o[dt], 0, 0, 0);
          }
        }
    }
  }
  // ---- finalize: l over the 4 lane groups, write O[q][d = dt*16 + 4g + r] ----
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (qrow < qlen) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    T* O = reinterpret_cast<T*>(p.O) + (int64_t)(q0 + qrow) * p.ldo + (int64_t)head * HD;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      if constexpr (sizeof(T) == 2) {
        bf16x4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (bf16_t)(o[dt][r] * inv);
        *reinterpret_cast<bf16x4*>(O + dt * 16 + 4 * g) = w;
      } else {
        f32x4 w = o[dt] * inv;
        *reinterpret_cast<f32x4*>(O + dt * 16 + 4 * g) = w;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(kThreads) void decode_attn_kernel(const DecodeAttnArgs p) {
  constexpr int VEC = TypeInfo<T>::kVec;
  constexpr int NCH = HD / VEC;                  // chunks per row: 8 (bf16 64), 16, 32
  constexpr int MAXG = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int row = blockIdx.x, kvh = blockIdx.y;
  const int G = p.n_heads / p.n_kv_heads;
  const int utt = row / p.beams;
  const int plen = p.prefix_len[utt];
  const int ntail = *p.step;
  const int ctx = plen + ntail;
  const int ctx_cap = p.max_prefix + p.max_tail;
  float* sq = reinterpret_cast<float*>(smem);            // [G][HD]
  float* ss = sq + MAXG * HD;                            // [G][ctx_cap]
  float* sred = ss + MAXG * ctx_cap;                     // [4][G][HD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* Q = reinterpret_cast<const T*>(p.Q) + (int64_t)row * p.ldq + (int64_t)kvh * G * HD;
  for (int i = tid; i < G * HD; i += kThreads) sq[i] = to_f32(Q[i]) * p.scale;
  __syncthreads();
  auto kv_row = [&](const void* pre, const void* tail, int j) -> const T* {
    if (j < plen) return reinterpret_cast<const T*>(pre) + (((int64_t)utt * p.n_kv_heads + kvh) * p.max_prefix + j) * HD;
    int t = j - plen;
    int src = p.anc ? p.anc[(int64_t)row * p.max_tail + t] : row;
    return reinterpret_cast<const T*>(tail) + (((int64_t)src * p.n_kv_heads + kvh) * p.max_tail + t) * HD;
  };
  // phase 1: scores
  for (int j = tid; j < ctx; j += kThreads) {
    const T* kr = kv_row(p.Kp, p.Kt, j);
    float acc[MAXG];
#pragma unroll
    for (int h = 0; h < MAXG; ++h) acc[h] = 0.f;
#pragma unroll 4
    for (int c = 0; c < NCH; ++c) {
      Chunk16 ck = *reinterpret_cast<const Chunk16*>(kr + c * VEC);
      float kf[VEC];
      chunk_to_f32<T>(ck, kf);
#pragma unroll
      for (int h = 0; h < MAXG; ++h) {
        if (h < G) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[h] += kf[e] * sq[h * HD + c * VEC + e];
        }
      }
    }
#pragma unroll
    for (int h = 0; h < MAXG; ++h)
      if (h < G) ss[h * ctx_cap + j] = acc[h];
  }
  __syncthreads();
  // phase 2: softmax per head (one wave per head, strided)
  for (int h = wave; h < G; h += 4) {
    float m = -INFINITY;
    for (int j = lane; j < ctx; j += 64) m = fmaxf(m, ss[h * ctx_cap + j]);
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < ctx; j += 64) { float e = __expf(ss[h * ctx_cap + j] - m); ss[h * ctx_cap + j] = e; l += e; }
    l = wave_sum(l);
    float inv = 1.f / l;
    for (int j = lane; j < ctx; j += 64) ss[h * ctx_cap + j] *= inv;
  }
  __syncthreads();
  // phase 3: out[h][d] = sum_j p[h][j] V[j][d];  thread -> (key group kg, chunk c)
  constexpr int KG = kThreads / NCH;
  const int c = tid % NCH, kg = tid / NCH;
  float acc[MAXG][VEC];
#pragma unroll
  for (int h = 0; h < MAXG; ++h)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[h][e] = 0.f;
  for (int j = kg; j < ctx; j += KG) {
    const T* vr = kv_row(p.Vp, p.Vt, j);
    Chunk16 cv = *reinterpret_cast<const Chunk16*>(vr + c * VEC);
    float vf[VEC];
    chunk_to_f32<T>(cv, vf);
#pragma unroll
    for (int h = 0; h < MAXG; ++h) {
      if (h < G) {
        float pj = ss[h * ctx_cap + j];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[h][e] += pj * vf[e];
      }
    }
  }
  // reduce over key groups: lanes with equal c inside the wave, then across waves through LDS
#pragma unroll
  for (int h = 0; h < MAXG; ++h)
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float v = acc[h][e];
      for (int o = NCH; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      acc[h][e] = v;
    }
  if (lane < NCH) {
#pragma unroll
    for (int h = 0; h < MAXG; ++h)
      if (h < G)
#pragma unroll
        for (int e = 0; e < VEC; ++e) sred[(wave * MAXG + h) * HD + lane * VEC + e] = acc[h][e];
  }
  __syncthreads();
  T* O = reinterpret_cast<T*>(p.O) + (int64_t)row * p.ldo + (int64_t)kvh * G * HD;
  for (int i = tid; i < G * HD; i += kThreads) {
    int h = i / HD, d = i - h * HD;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += sred[(w * MAXG + h) * HD + d];
    O[i] = from_f32<T>(v);
  }
}

template <typename T, int HD>
int launch_attn_t(const AttnArgs& a, hipStream_t st) {
  dim3 grid(ceil_div(a.max_q, 64), a.n_heads, a.n_seqs);
  hipLaunchKernelGGL((attn_varlen_kernel<T, HD>), grid, dim3(kThreads), 0, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int HD>
int launch_dec_t(const DecodeAttnArgs& a, hipStream_t st) {
  size_t smem = sizeof(float) * (8 * HD + 8 * (size_t)(a.max_prefix + a.max_tail) + 4 * 8 * HD);
  OAVSR_REQUIRE(smem <= 160 * 1024, "decode attention: context %d too long for LDS", a.max_prefix + a.max_tail);
  auto kern = decode_attn_kernel<T, HD>;
  static size_t attr = 0;
  if (smem > attr) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = smem;
  }
  dim3 grid(a.rows, a.n_kv_heads);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_attention(const AttnArgs& a, hipStream_t st) {
  if (a.n_seqs <= 0 || a.max_q <= 0) return 0;
  OAVSR_REQUIRE(a.head_dim == 64 || a.head_dim == 128, "attention: head_dim %d unsupported", a.head_dim);
  OAVSR_REQUIRE(a.n_heads % a.n_kv_heads == 0, "attention: heads %d not a multiple of kv heads %d", a.n_heads, a.n_kv_heads);
  const int vec = a.dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(a.ldq % vec == 0 && a.ldk % vec == 0 && a.ldv % vec == 0 && a.ldo % vec == 0, "attention: rows must be 16-byte aligned");
  OAVSR_REQUIRE((((uintptr_t)a.Q | (uintptr_t)a.K | (uintptr_t)a.V | (uintptr_t)a.O) & 15) == 0, "attention: pointers must be 16-byte aligned");
  if (a.dtype == DT_F32) return a.head_dim == 64 ? launch_attn_t<float, 64>(a, st) : launch_attn_t<float, 128>(a, st);
  return a.head_dim == 64 ? launch_attn_t<bf16_t, 64>(a, st) : launch_attn_t<bf16_t, 128>(a, st);
}

int launch_decode_attention(const DecodeAttnArgs& a, hipStream_t st) {
  if (a.rows <= 0) return 0;
  OAVSR_REQUIRE(a.head_dim == 64 || a.head_dim == 128, "decode attention: head_dim %d unsupported", a.head_dim);
  OAVSR_REQUIRE(a.n_heads % a.n_kv_heads == 0 && a.n_heads / a.n_kv_heads <= 8, "decode attention: GQA group must be <= 8");
  if (a.dtype == DT_F32) return a.head_dim == 64 ? launch_dec_t<float, 64>(a, st) : launch_dec_t<float, 128>(a, st);
  return a.head_dim == 64 ? launch_dec_t<bf16_t, 64>(a, st) : launch_dec_t<bf16_t, 128>(a, st);
}

}  // namespace oavsr
