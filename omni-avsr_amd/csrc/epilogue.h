// Device-side epilogue shared by the decode-shaped GEMM kernels (gemm2.hip: the 128-row LDS-DMA tiles and their split-K reduce;
// decode_strip.hip: the register-streaming strips for <= 32 rows).  ONE definition of how a finished 16-column row segment becomes
// output, so a row has the same bits whichever kernel -- whichever row-count regime -- produced it.
#pragma once
#include "kernels.h"

namespace oavsr {
namespace {

// one MFMA step over a 16-byte chunk of K per lane: a = activation rows (A operand), b = weight rows (B operand).  Every decode-shaped kernel
// accumulates through this, k-tile by k-tile, chunk kb = 0 then 1: the summation order that makes a row's bits independent of the kernel
template <typename T> struct Mma2;
template <> struct Mma2<bf16_t> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
  }
};
template <> struct Mma2<float> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    f32x4 fa = __builtin_bit_cast(f32x4, a.v), fb = __builtin_bit_cast(f32x4, b.v);
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb[s], c, 0, 0, 0);
  }
};

// The epilogue of ONE finished 16-column row segment of a decode-shaped GEMM (yy: the fp32 sums): row factor, col_scale / bias, residual,
// activation, store; optionally the consumer's RMSNorm operands (GemmArgs::ng_*).  The four lanes tid % 4 = 0..3 hold the four segments of
// one row of a 64-column tile; every lane of the wave must call (the sum of squares crosses those lanes).  The one place where the results
// of a split decode-shaped GEMM are formed (the reduce inside the GEMM launch and the stand-alone reduce kernel both end here).
template <typename T>
__device__ __forceinline__ void segment_finish(const GemmArgs& p, int m, int nc, int nt_, bool live, float (&yy)[16],
                                               const Chunk16 (&rraw)[16 / TypeInfo<T>::kVec], float rsv, int tid) {
  constexpr int VEC = TypeInfo<T>::kVec;
  constexpr int SEG = 4;
  const T* Rr = reinterpret_cast<const T*>(p.residual);
  const bool want_ng = p.ng_out != nullptr;
  float ss = 0.f;
  if (live) {
    if (p.row_sumsq) {
#pragma unroll
      for (int e = 0; e < 16; ++e) yy[e] *= rsv;
    }
    float res[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) res[e] = 0.f;
    if (Rr) {
#pragma unroll
      for (int e = 0; e < 16; e += VEC) chunk_to_f32<T>(rraw[e / VEC], res + e);
    }
    const bool whole = nc + 16 <= p.N;                        // (a ragged last segment: un-split launches only, e.g. lm_head's vocabulary)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const bool in = whole || nc + e < p.N;
      float v = fmaf(yy[e], (p.col_scale && in) ? p.col_scale[nc + e] : 1.f, (p.col_bias && in) ? p.col_bias[nc + e] : 0.f);
      if (!p.res_after_act) v += res[e];
      switch (p.act) {
        case ACT_GELU: v = gelu_act<T>(v); break;
        case ACT_RELU: v = fmaxf(v, 0.f); break;
        case ACT_PRELU: v = v >= 0.f ? v : (in ? p.prelu[nc + e] : 0.f) * v; break;
        case ACT_SILU: v = silu(v); break;
        default: break;
      }
      yy[e] = p.res_after_act ? v + res[e] : v;
    }
    if (p.out_f32) {
      float* Cf = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + nc;
      if (whole && (p.ldc % 4) == 0 && ((uintptr_t)p.C & 15) == 0 && (nc % 4) == 0) {
#pragma unroll
        for (int e = 0; e < 16; e += 4) *reinterpret_cast<f32x4*>(Cf + e) = f32x4{yy[e], yy[e + 1], yy[e + 2], yy[e + 3]};
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) if (nc + e < p.N) Cf[e] = yy[e];
      }
    } else if (!(whole && (p.ldc % VEC) == 0 && ((uintptr_t)p.C & 15) == 0)) {
      T* Ct = reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + nc;
#pragma unroll
      for (int e = 0; e < 16; ++e) if (nc + e < p.N) Ct[e] = from_f32<T>(yy[e]);
    } else {
      T* Ct = reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + nc;
      Chunk16 oc[16 / VEC];
#pragma unroll
      for (int e = 0; e < 16; e += VEC) { oc[e / VEC] = chunk_from_f32<T>(yy + e); *reinterpret_cast<Chunk16*>(Ct + e) = oc[e / VEC]; }
      if (want_ng) {      // the consumer's RMSNorm: x * gain now (x as stored), the row's sum of squares per column tile
        float xr[16], xg[16];
#pragma unroll
        for (int e = 0; e < 16; e += VEC) chunk_to_f32<T>(oc[e / VEC], xr + e);
#pragma unroll
        for (int e = 0; e < 16; ++e) { xg[e] = xr[e] * p.ng_gain[nc + e]; ss = fmaf(xr[e], xr[e], ss); }
        T* Gt = reinterpret_cast<T*>(p.ng_out) + (int64_t)m * p.ng_ld + nc;
#pragma unroll
        for (int e = 0; e < 16; e += VEC) *reinterpret_cast<Chunk16*>(Gt + e) = chunk_from_f32<T>(xg + e);
      }
    }
  }
  if (want_ng && p.rowsq_strips) {      // the strip kernels: one 16-column sum per (strip, row); the CONSUMER forms the 64-column tile sums
    if (live) p.ng_rowsq[(int64_t)(nc >> 4) * p.M + m] = ss;      // ((s0 + s1) + (s2 + s3): the butterfly below, decode_strip.hip: strip_row_factor)
  } else if (want_ng) {      // the row's sum of squares over this tile's 64 columns: the row's four adjacent lanes, fixed butterfly
#pragma unroll
    for (int o = 1; o < SEG; o <<= 1) ss += __shfl_xor(ss, o, 64);
    if (live && (tid % SEG) == 0) p.ng_rowsq[(int64_t)nt_ * p.M + m] = ss;
  }
}

// 1 / rms of a row from the producer's per-tile sums of squares (GemmArgs::row_sumsq), tile by tile, in order
__device__ __forceinline__ float row_factor(const GemmArgs& p, int m) {
  float tot = 0.f;
  if (m < p.M) {
#pragma unroll 8
    for (int t = 0; t < p.row_sumsq_tiles; ++t) tot += p.row_sumsq[(int64_t)t * p.M + m];
  }
  return rsqrtf(tot / (float)p.row_dim + p.row_eps);
}

}  // namespace
}  // namespace oavsr
