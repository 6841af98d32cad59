// Shared device/host helpers for the Omni-AVSR gfx950 kernels.
// One code path only: CDNA4 (wave64, MFMA 16x16, 160 KiB LDS).  No CUDA compatibility layer.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace oavsr {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

enum DType : int { DT_F32 = 0, DT_BF16 = 1 };

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float> { static constexpr int kDType = DT_F32; static constexpr int kVec = 4; };
template <> struct TypeInfo<bf16_t> { static constexpr int kDType = DT_BF16; static constexpr int kVec = 8; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// A 16-byte chunk: 4 floats or 8 bf16.  All HBM<->LDS staging moves these.
struct __attribute__((aligned(16))) Chunk16 { u32x4 v; };

template <typename T> __device__ __forceinline__ void chunk_to_f32(const Chunk16& c, float* out);
template <> __device__ __forceinline__ void chunk_to_f32<float>(const Chunk16& c, float* out) {
  f32x4 f = __builtin_bit_cast(f32x4, c.v);
  out[0] = f[0]; out[1] = f[1]; out[2] = f[2]; out[3] = f[3];
}
template <> __device__ __forceinline__ void chunk_to_f32<bf16_t>(const Chunk16& c, float* out) {
  bf16x8 h = __builtin_bit_cast(bf16x8, c.v);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)h[i];
}
template <typename T> __device__ __forceinline__ Chunk16 chunk_from_f32(const float* in);
template <> __device__ __forceinline__ Chunk16 chunk_from_f32<float>(const float* in) {
  f32x4 f = {in[0], in[1], in[2], in[3]};
  Chunk16 c; c.v = __builtin_bit_cast(u32x4, f); return c;
}
template <> __device__ __forceinline__ Chunk16 chunk_from_f32<bf16_t>(const float* in) {
  bf16x8 h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (bf16_t)in[i];
  Chunk16 c; c.v = __builtin_bit_cast(u32x4, h); return c;
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU (erf form, torch.nn.functional.gelu default) of a value that will be rounded to T.
//   f32 : libm erff (parity mode).
//   bf16: erf by Abramowitz-Stegun 7.1.26 (|err| < 6.1e-7, 2^-9 of a bf16 ulp at |y| >= 2^-12): one rcp, one exp, six fma --
//         a third of erff's instructions in every fc1 epilogue.  Every GEMM variant calls this one function, so an element
//         has the same bits whichever kernel produced it.
template <typename T> __device__ __forceinline__ float gelu_act(float x);
template <> __device__ __forceinline__ float gelu_act<float>(float x) { return gelu_erf(x); }
// Written for two elements at a time in packed fp32 math (v_pk_mul / v_pk_fma / v_pk_add; only the rcp / exp2 stay scalar):
// that halves the VALU issue slots of the fc1 epilogues.  The scalar entry point is the same code on a duplicated lane.
__device__ __forceinline__ f32x2 gelu_act2_bf16(f32x2 x) {
  const f32x2 ax = f32x2{fabsf(x[0]), fabsf(x[1])};
  const f32x2 z = ax * 0.70710678118654752440f;
  const f32x2 den = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, z, f32x2{1.0f, 1.0f});
  const f32x2 t = f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2 pl = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
  pl = __builtin_elementwise_fma(pl, t, f32x2{1.421413741f, 1.421413741f});
  pl = __builtin_elementwise_fma(pl, t, f32x2{-0.284496736f, -0.284496736f});
  pl = __builtin_elementwise_fma(pl, t, f32x2{0.254829592f, 0.254829592f});
  pl = pl * t;
  const f32x2 nz2 = (-z) * z;
  const f32x2 ex = f32x2{__expf(nz2[0]), __expf(nz2[1])};
  const f32x2 e = __builtin_elementwise_fma(-pl, ex, f32x2{1.0f, 1.0f});   // erf(|x| / sqrt 2)
  const f32x2 se = f32x2{copysignf(e[0], x[0]), copysignf(e[1], x[1])};
  return (x * 0.5f) * (se + 1.0f);
}
template <> __device__ __forceinline__ float gelu_act<bf16_t>(float x) { return gelu_act2_bf16(f32x2{x, x})[0]; }
template <typename T> __device__ __forceinline__ void gelu_act4(float* v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = gelu_act<T>(v[q]);
}
template <> __device__ __forceinline__ void gelu_act4<bf16_t>(float* v) {
  const f32x2 a = gelu_act2_bf16(f32x2{v[0], v[1]}), b = gelu_act2_bf16(f32x2{v[2], v[3]});
  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- developer switches ---------------------------------------------------------------------
// The ONLY place the library reads its environment (model.hip: kKnobs).  Every switch left picks between forms that give the same bits
// (tests compare them) or turns a diagnostic on: nothing a process environment can set changes a result.  An unregistered name aborts.
int knob(const char* name);

// ---- host side error plumbing -------------------------------------------------------------
void set_last_error(const char* fmt, ...);
#define OAVSR_CHECK_HIP(expr)                                                              \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      oavsr::set_last_error("%s:%d HIP error %d (%s) in %s", __FILE__, __LINE__, (int)_e, \
                            hipGetErrorString(_e), #expr);                                 \
      return -2;                                                                           \
    }                                                                                      \
  } while (0)
#define OAVSR_REQUIRE(cond, ...)                    \
  do {                                              \
    if (!(cond)) {                                  \
      oavsr::set_last_error(__VA_ARGS__);           \
      return -1;                                    \
    }                                               \
  } while (0)
#define OAVSR_TRY(expr)       \
  do {                        \
    int _r = (expr);          \
    if (_r != 0) return _r;   \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace oavsr
