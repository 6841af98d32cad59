// Visual front-end stem, fused: Conv3d(1->64, k=(5,7,7), s=(1,2,2), p=(2,3,3)) + BatchNorm3d + PReLU + MaxPool3d((1,3,3),
// s=(1,2,2), p=(0,1,1))  (reference: resnet.py:137-141 via hubert.py frontend) for gfx950.
//
// One workgroup (4 waves) owns one output frame and walks it top to bottom in 11 steps of 4 conv rows (= 176 pixels =
// 11 MFMA tiles of 16 pixels):
//   1. the 5 frames x 13 rows x 94 columns input window of the step is gathered (f32 video -> T, zero padding for the
//      borders and for frames outside the utterance) into LDS with 3 pad columns on the left, so that the 8 taps
//      kx = 0..7 (7 real + 1 zero-weight) of output column ox start at the dword-aligned LDS column 2*ox;
//   2. implicit GEMM on MFMA with K ordered (dt, ky, kx8) = 35 x 8 (+8 zero) = 288: a lane's 16-byte fragment is 8 consecutive
//      window pixels of one (dt, ky) row -- no im2col buffer, no per-element index math.  The packed weights (64 x 288,
//      built once per model by stem_repack_kernel) are the A operand and live in registers for the whole frame (bf16);
//      the pixels are the B operand, so a lane ends up with 4 consecutive channels of one pixel;
//   3. BN (one fmaf, same formula as every GEMM epilogue) + PReLU, rounded to T, into a 5-row LDS ring of conv rows;
//   4. the two pooled rows that became complete are reduced (3x3 max, -inf padding) and stored as 16-byte vectors.
// The 44 x 44 x 64 conv activation never touches HBM (it was 793 MB written + read per 8 x 16 s batch).
//
// Pixel source (template SRC): 0 = f32 [frame][88][88], what the reference's dataset hands over after its test-time transform;
// 1 / 3 = the DECODED frames themselves, uint8 [frame][C][H][W] with C = 1 / 3 channels (H, W >= 88): the transform of
// datamodule/transforms.py:93-99 -- x / 255, CenterCrop(88), Grayscale (0.2989 R + 0.587 G + 0.114 B), Normalize(0.421, 0.165) --
// then happens in the window gather, with the same fp32 operations in the same order as the host code (explicitly rounded, no
// FMA contraction; the per-channel products come from 256-entry tables built in LDS), so the window holds the same bits either way.
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kStemThreads = 256;
constexpr int kWinRows = 13, kWinCols = 96, kKRows = 35, kKPacked = 288;
constexpr int kRing = 5;

template <typename T> struct MmaStem;
template <> struct MmaStem<bf16_t> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
  }
};
template <> struct MmaStem<float> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    f32x4 fa = __builtin_bit_cast(f32x4, a.v), fb = __builtin_bit_cast(f32x4, b.v);
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb[s], c, 0, 0, 0);
  }
};

// w: [64][256] with k = dt*49 + ky*7 + kx (245 valid)   ->   wp: [64][288] with k = (dt*7 + ky)*8 + kx, zero padded
template <typename T>
__global__ void stem_repack_kernel(const T* w, T* wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * kKPacked) return;
  const int n = i / kKPacked, k = i - n * kKPacked;
  const int krow = k >> 3, kx = k & 7;
  wp[i] = (krow < kKRows && kx < 7) ? w[n * 256 + krow * 7 + kx] : from_f32<T>(0.f);
}

template <typename T, int SRC>
__global__ __launch_bounds__(kStemThreads) void stem_pool_kernel(const void* __restrict__ vid_, const int2* __restrict__ frameinfo,
                                                                 const T* __restrict__ wp, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ slope,
                                                                 T* __restrict__ out, StemSrc src_fmt) {
  constexpr int VEC = TypeInfo<T>::kVec;
  constexpr int NKS = kKPacked / (4 * VEC);            // k steps: 9 (bf16, 32 deep) / 18 (f32, 16 deep)
  constexpr bool kWRegs = sizeof(T) == 2;              // bf16: the whole packed weight stays in registers (144 VGPRs)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* swin = reinterpret_cast<T*>(smem);                // [5][13][96]
  T* sconv = swin + 5 * kWinRows * kWinCols;           // [5 ring][44][64]
  float* spar = reinterpret_cast<float*>(sconv + kRing * 44 * 64);   // scale[64] shift[64] slope[64]
  float* slut = spar + 192;                            // SRC != 0: [SRC][256] per-channel tables of the pixel transform

  const int f = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int2 fi = frameinfo[f];

  if (tid < 64) { spar[tid] = scale[tid]; spar[64 + tid] = shift[tid]; spar[128 + tid] = slope[tid]; }
  if constexpr (SRC != 0) {
    // v / 255 (true division), times the grayscale weight for RGB -- or the whole transform for single-channel frames
    const float x = __fdiv_rn((float)tid, 255.0f);
    if constexpr (SRC == 1) slut[tid] = __fdiv_rn(__fsub_rn(x, 0.421f), 0.165f);
    else { slut[tid] = __fmul_rn(0.2989f, x); slut[256 + tid] = __fmul_rn(0.587f, x); slut[512 + tid] = __fmul_rn(0.114f, x); }
    __syncthreads();
  }

  Chunk16 wreg[kWRegs ? NKS : 1][kWRegs ? 4 : 1];
  if constexpr (kWRegs) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) wreg[ks][j] = *reinterpret_cast<const Chunk16*>(wp + (j * 16 + r16) * kKPacked + ks * 4 * VEC + g * VEC);
  }

  // per-lane window offset of k step ks: the (dt, ky) row this lane's fragment comes from (+ the kx half for f32)
  auto koff = [&](int ks) -> int {
    int krow, kx0;
    if (sizeof(T) == 2) { krow = ks * 4 + g; kx0 = 0; }
    else { krow = ks * 2 + (g >> 1); kx0 = 4 * (g & 1); }
    krow = krow < kKRows ? krow : kKRows - 1;          // zero-weight pad row: any finite data will do
    const int dt = krow / 7, ky = krow - dt * 7;
    return (dt * kWinRows + ky) * kWinCols + kx0;
  };

  const float* vid = reinterpret_cast<const float*>(vid_);
  const float* vbase = vid + (int64_t)f * 7744;
  const int plane = src_fmt.H * src_fmt.W;             // SRC != 0: bytes per channel plane; frame = SRC planes
  const unsigned char* ubase = reinterpret_cast<const unsigned char*>(vid_) + (int64_t)f * SRC * plane + src_fmt.top * src_fmt.W + src_fmt.left;
  for (int it = 0; it < 11; ++it) {
    // ---- 1. input window: rows 8*it-3 .. 8*it+9, columns -3 .. 92, frames t-2 .. t+2 ----
    // fully unrolled with unconditional loads (clamped address + select): all 26 loads of a thread are in flight together and
    // the step pays the global latency once, not once per loop trip
    constexpr int NQ = 5 * kWinRows * (kWinCols / 2);
    constexpr int kFill = (NQ + kStemThreads - 1) / kStemThreads;
    constexpr int kGrp = SRC == 3 ? 4 : 7;                 // loads in flight per thread: 2 * kGrp (x 3 channel bytes); register budget: two workgroups per CU
#pragma unroll 1
    for (int k0 = 0; k0 < kFill; k0 += kGrp) {
      float pv0[kGrp], pv1[kGrp];
      unsigned char rb0[SRC ? kGrp : 1][SRC ? SRC : 1], rb1[SRC ? kGrp : 1][SRC ? SRC : 1];
      bool okk0[kGrp], okk1[kGrp];
#pragma unroll
      for (int k = 0; k < kGrp; ++k) {
        const int q = tid + (k0 + k) * kStemThreads, qq = q < NQ ? q : 0;
        const int dt = qq / (kWinRows * (kWinCols / 2)), rem = qq - dt * (kWinRows * (kWinCols / 2));
        const int wr = rem / (kWinCols / 2), cp = rem - wr * (kWinCols / 2);
        const int iy = 8 * it - 3 + wr, ix = 2 * cp - 3;
        const bool row = (unsigned)(fi.x + dt - 2) < (unsigned)fi.y && (unsigned)iy < 88u;
        const bool ok0 = row && (unsigned)ix < 88u, ok1 = row && (unsigned)(ix + 1) < 88u;
        okk0[k] = ok0; okk1[k] = ok1;
        if constexpr (SRC == 0) {
          const float* src = vbase + (int64_t)(dt - 2) * 7744 + iy * 88;
          const float a = *(ok0 ? src + ix : vbase), b = *(ok1 ? src + ix + 1 : vbase);
          pv0[k] = ok0 ? a : 0.f; pv1[k] = ok1 ? b : 0.f;
        } else {
          const unsigned char* src = ubase + (int64_t)(dt - 2) * SRC * plane + iy * src_fmt.W;
#pragma unroll
          for (int c = 0; c < SRC; ++c) {
            rb0[k][c] = *(ok0 ? src + c * plane + ix : ubase);
            rb1[k][c] = *(ok1 ? src + c * plane + ix + 1 : ubase);
          }
        }
      }
      if constexpr (SRC != 0) {
#pragma unroll
        for (int k = 0; k < kGrp; ++k) {
          float a, b;
          if constexpr (SRC == 1) { a = slut[rb0[k][0]]; b = slut[rb1[k][0]]; }
          else {
            a = __fdiv_rn(__fsub_rn(__fadd_rn(__fadd_rn(slut[rb0[k][0]], slut[256 + rb0[k][1]]), slut[512 + rb0[k][2]]), 0.421f), 0.165f);
            b = __fdiv_rn(__fsub_rn(__fadd_rn(__fadd_rn(slut[rb1[k][0]], slut[256 + rb1[k][1]]), slut[512 + rb1[k][2]]), 0.421f), 0.165f);
          }
          pv0[k] = okk0[k] ? a : 0.f; pv1[k] = okk1[k] ? b : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < kGrp; ++k) {
        const int q = tid + (k0 + k) * kStemThreads;
        if (q < NQ) {
          const int dt = q / (kWinRows * (kWinCols / 2)), rem = q - dt * (kWinRows * (kWinCols / 2));
          const int wr = rem / (kWinCols / 2), cp = rem - wr * (kWinCols / 2);
          T* dst = swin + (dt * kWinRows + wr) * kWinCols + 2 * cp;
          dst[0] = from_f32<T>(pv0[k]); dst[1] = from_f32<T>(pv1[k]);
        }
      }
    }
    __syncthreads();

    // ---- 2./3. 11 pixel tiles (16 pixels each) over the 4 waves: conv + BN + PReLU -> ring ----
    for (int mt = wave; mt < 11; mt += 4) {
      const int q = mt * 16 + r16;                     // pixel of this lane's B fragment
      const int lr = q / 44, ox = q - lr * 44;
      const T* pix = swin + (2 * lr) * kWinCols + 2 * ox;
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const T* p = pix + koff(ks);
        Chunk16 pf;
        if constexpr (sizeof(T) == 2) {                // 4-byte aligned: four dword reads
          const uint32_t* p32 = reinterpret_cast<const uint32_t*>(p);
          pf.v = u32x4{p32[0], p32[1], p32[2], p32[3]};
        } else {                                       // 8-byte aligned: two qword reads
          const uint64_t* p64 = reinterpret_cast<const uint64_t*>(p);
          const uint64_t a = p64[0], b = p64[1];
          pf.v = u32x4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (kWRegs) MmaStem<T>::run(wreg[ks][j], pf, acc[j]);
          else MmaStem<T>::run(*reinterpret_cast<const Chunk16*>(wp + (j * 16 + r16) * kKPacked + ks * 4 * VEC + g * VEC), pf, acc[j]);
        }
      }
      // D[channel = 16j + 4g + r][pixel = r16]: 4 consecutive channels of one pixel per lane
      const int cy = 4 * it + lr;
      T* dst = sconv + (((cy % kRing) * 44) + ox) * 64;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = j * 16 + 4 * g;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(spar + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(spar + 64 + c0);
        const f32x4 sl = *reinterpret_cast<const f32x4*>(spar + 128 + c0);
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[j][r], sc[r], sh[r]);
          y[r] = v >= 0.f ? v : sl[r] * v;
        }
        if constexpr (sizeof(T) == 2) {
          bf16x4 h = {(bf16_t)y[0], (bf16_t)y[1], (bf16_t)y[2], (bf16_t)y[3]};
          *reinterpret_cast<bf16x4*>(dst + c0) = h;
        } else {
          *reinterpret_cast<f32x4*>(dst + c0) = f32x4{y[0], y[1], y[2], y[3]};
        }
      }
    }
    __syncthreads();

    // ---- 4. pooled rows 2*it, 2*it+1: 3x3 max over conv rows 2py-1 .. 2py+1, columns 2px-1 .. 2px+1 ----
    constexpr int NG = 64 / VEC;
    for (int idx = tid; idx < 2 * 22 * NG; idx += kStemThreads) {
      const int pr = idx / (22 * NG), rem = idx - pr * (22 * NG);
      const int px = rem / NG, cg = rem - px * NG;
      const int py = 2 * it + pr;
      float m[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int cy = 2 * py + dy;
        if ((unsigned)cy >= 44u) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int cx = 2 * px + dx;
          if ((unsigned)cx >= 44u) continue;
          float v[VEC];
          chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(sconv + (((cy % kRing) * 44) + cx) * 64 + cg * VEC), v);
#pragma unroll
          for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], v[e]);
        }
      }
      *reinterpret_cast<Chunk16*>(out + (((int64_t)f * 22 + py) * 22 + px) * 64 + cg * VEC) = chunk_from_f32<T>(m);
    }
    // the next window load only touches swin (its readers passed the barrier above); the ring rows it will overwrite are
    // protected by the barrier after that load
  }
}

template <typename T, int SRC>
int launch_stem_t(const void* video, const int2* frameinfo, int F, const void* wp, const float* scale, const float* shift, const float* slope,
                  void* out, const StemSrc& src, hipStream_t st) {
  constexpr size_t smem = (5 * kWinRows * kWinCols + kRing * 44 * 64) * sizeof(T) + (192 + SRC * 256) * sizeof(float);
  static_assert(smem <= 160 * 1024, "LDS");
  auto kern = stem_pool_kernel<T, SRC>;
  static bool attr = false;
  if (!attr && smem > 48 * 1024) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3(F), dim3(kStemThreads), smem, st, video, frameinfo, reinterpret_cast<const T*>(wp), scale, shift, slope,
                     reinterpret_cast<T*>(out), src);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_stem_repack(int dtype, const void* w, void* wp, hipStream_t st) {
  const int n = 64 * kKPacked;
  if (dtype == DT_F32) hipLaunchKernelGGL(stem_repack_kernel<float>, dim3(ceil_div(n, 256)), dim3(256), 0, st, (const float*)w, (float*)wp);
  else hipLaunchKernelGGL(stem_repack_kernel<bf16_t>, dim3(ceil_div(n, 256)), dim3(256), 0, st, (const bf16_t*)w, (bf16_t*)wp);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_stem_pool(int dtype, const void* video, const int2* frameinfo, int F, const void* wp, const float* scale, const float* shift,
                     const float* slope, void* out, hipStream_t st, const StemSrc& src) {
  if (F <= 0) return 0;
  OAVSR_REQUIRE(video && frameinfo && wp && scale && shift && slope && out, "stem: null argument");
  OAVSR_REQUIRE(src.C == 0 || ((src.C == 1 || src.C == 3) && src.H >= 88 && src.W >= 88 && src.top >= 0 && src.left >= 0 && src.top + 88 <= src.H &&
                               src.left + 88 <= src.W), "stem: uint8 frames must be [C = 1 or 3][H >= 88][W >= 88] (got C=%d, %d x %d)", src.C, src.H, src.W);
#define STEM_GO(T) (src.C == 0 ? launch_stem_t<T, 0>(video, frameinfo, F, wp, scale, shift, slope, out, src, st) \
                    : src.C == 1 ? launch_stem_t<T, 1>(video, frameinfo, F, wp, scale, shift, slope, out, src, st) \
                                 : launch_stem_t<T, 3>(video, frameinfo, F, wp, scale, shift, slope, out, src, st))
  return dtype == DT_F32 ? STEM_GO(float) : STEM_GO(bf16_t);
#undef STEM_GO
}

}  // namespace oavsr
