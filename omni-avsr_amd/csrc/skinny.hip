// Skinny GEMM for the decode loop (M <= 128 rows: B utterances x K beams), gfx950.
//
// Weight-streaming regime: every weight byte is read exactly once per step, straight from HBM into the MFMA
// B-operand registers (lane l holds W[n0 + (l&15)][k + 8*(l>>4) ..] = one 16-byte global load, no LDS round
// trip); the (tiny, L2-resident) activations are the A operand.  The four waves of a workgroup interleave the
// 64-byte k blocks of the same NT*16 output columns and combine their partial sums through LDS, then apply the
// same fused epilogue as the big GEMM.  The optional second K segment adds the LoRA term (W + BA fused).
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ Chunk16 zchunk() { Chunk16 c; c.v = u32x4{0u, 0u, 0u, 0u}; return c; }

template <typename T> struct MmaS;
template <> struct MmaS<bf16_t> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
  }
};
template <> struct MmaS<float> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    f32x4 fa = __builtin_bit_cast(f32x4, a.v), fb = __builtin_bit_cast(f32x4, b.v);
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb[s], c, 0, 0, 0);
  }
};

template <typename T, int MT, int NT>
__global__ __launch_bounds__(kThreads) void skinny_kernel(const GemmArgs p) {
  constexpr int KBE = 64 / (int)sizeof(T);       // elements per 64-byte k block
  constexpr int VEC = TypeInfo<T>::kVec;
  constexpr int U = 4;                            // k blocks in flight per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sred = reinterpret_cast<float*>(smem);   // [4][MT*16][NT*16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (NT * 16);
  const T* A1 = reinterpret_cast<const T*>(p.A);
  const T* A2 = reinterpret_cast<const T*>(p.A2);
  const T* W1 = reinterpret_cast<const T*>(p.W);
  const T* W2 = reinterpret_cast<const T*>(p.W2);
  const int nkb1 = p.K / KBE, nkb = nkb1 + p.K2 / KBE;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int kb0 = wave; kb0 < nkb; kb0 += 4 * U) {
    Chunk16 bf[U][NT], af[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = kb0 + 4 * u;
      const bool in = kb < nkb;
      const bool s1 = kb < nkb1;
      const T* Ab = s1 ? A1 : A2;
      const T* Wb = s1 ? W1 : W2;
      const int64_t lda = s1 ? p.lda : p.lda2, ldw = s1 ? p.ldw : p.ldw2;
      const int k = (s1 ? kb : kb - nkb1) * KBE + g * VEC;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 16 + r16;
        bf[u][j] = (in && n < p.N) ? *reinterpret_cast<const Chunk16*>(Wb + (int64_t)n * ldw + k) : zchunk();
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = i * 16 + r16;
        af[u][i] = (in && m < p.M) ? *reinterpret_cast<const Chunk16*>(Ab + (int64_t)m * lda + k) : zchunk();
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) MmaS<T>::run(af[u][i], bf[u][j], acc[i][j]);
  }
  // partial sums -> LDS (C layout: col = lane&15, row = 4*(lane>>4)+r)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sred[(wave * (MT * 16) + i * 16 + g * 4 + r) * (NT * 16) + j * 16 + r16] = acc[i][j][r];
  __syncthreads();
  const T* R = reinterpret_cast<const T*>(p.residual);
  for (int idx = tid; idx < MT * 16 * NT * 16; idx += kThreads) {
    const int m = idx / (NT * 16), nn = idx - m * (NT * 16);
    const int n = n0 + nn;
    if (m >= p.M || n >= p.N) continue;
    float y = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) y += sred[(w * (MT * 16) + m) * (NT * 16) + nn];
    if (p.col_scale) y *= p.col_scale[n];
    if (p.col_bias) y += p.col_bias[n];
    float res = R ? to_f32(R[(int64_t)m * p.ldr + n]) : 0.f;
    if (!p.res_after_act) y += res;
    switch (p.act) {
      case ACT_GELU: y = gelu_erf(y); break;
      case ACT_RELU: y = fmaxf(y, 0.f); break;
      case ACT_PRELU: y = y >= 0.f ? y : p.prelu[n] * y; break;
      case ACT_SILU: y = silu(y); break;
      default: break;
    }
    if (p.res_after_act) y += res;
    if (p.out_f32) reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n] = y;
    else reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + n] = from_f32<T>(y);
  }
}

template <typename T, int MT, int NT>
int launch_sk(const GemmArgs& a, hipStream_t st) {
  constexpr int smem = 4 * MT * 16 * NT * 16 * 4;
  auto kern = skinny_kernel<T, MT, NT>;
  static bool attr_set = false;
  if (!attr_set && smem > 48 * 1024) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(ceil_div(a.N, NT * 16)), dim3(kThreads), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_sk_t(const GemmArgs& a, hipStream_t st) {
  const int mt = ceil_div(a.M, 16);
  const bool wide = a.N >= 16384;          // enough columns to fill the chip with 64-column workgroups
  if (mt <= 1) return wide ? launch_sk<T, 1, 4>(a, st) : launch_sk<T, 1, 1>(a, st);
  if (mt <= 2) return wide ? launch_sk<T, 2, 4>(a, st) : launch_sk<T, 2, 1>(a, st);
  if (mt <= 4) return launch_sk<T, 4, 1>(a, st);
  return launch_sk<T, 8, 1>(a, st);
}

}  // namespace

int launch_gemm_skinny(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return 0;
  const int esz = a.dtype == DT_F32 ? 4 : 2;
  const int kbe = 64 / esz, vec = 16 / esz;
  OAVSR_REQUIRE(a.M <= 128, "skinny gemm: M=%d > 128", a.M);
  OAVSR_REQUIRE(a.amode == A_DENSE && a.groups == 1, "skinny gemm: dense only");
  OAVSR_REQUIRE(a.K > 0 && a.K % kbe == 0 && a.K2 % kbe == 0, "skinny gemm: K=%d/K2=%d must be multiples of %d", a.K, a.K2, kbe);
  OAVSR_REQUIRE(a.lda % vec == 0 && a.ldw % vec == 0 && (((uintptr_t)a.A | (uintptr_t)a.W) & 15) == 0, "skinny gemm: operands must be 16-byte aligned");
  if (a.K2) OAVSR_REQUIRE(a.lda2 % vec == 0 && a.ldw2 % vec == 0 && (((uintptr_t)a.A2 | (uintptr_t)a.W2) & 15) == 0, "skinny gemm: segment-2 operands must be 16-byte aligned");
  return a.dtype == DT_F32 ? launch_sk_t<float>(a, st) : launch_sk_t<bf16_t>(a, st);
}

}  // namespace oavsr
