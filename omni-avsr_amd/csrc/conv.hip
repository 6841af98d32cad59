// Image-resident 3x3 convolution for the 64-channel stage of the visual ResNet trunk (layer1: four 64 -> 64 convs on
// 22 x 22 maps; reference resnet.py:61-74 BasicBlock), gfx950, bf16.
//
// The implicit-GEMM path gathers every input pixel nine times (once per tap) through L2 -> LDS; here a whole NHWC image
// (484 pixels x 128 B = 61 KB) plus the complete weight matrix (64 x 576 = 72 KB) live in LDS, so HBM/L2 see each input
// byte once and the MFMA fragments are plain 16-byte LDS reads:
//   * persistent workgroups (one per CU, 8 waves) load the weights once and walk images b, b + G, b + 2G, ...;
//     the next image is prefetched into registers while the current one is computed;
//   * a wave owns 4 pixel tiles (64 output pixels) and shares every weight fragment among them;
//   * K runs tap-major then channels -- the same order as the implicit GEMM -- and the epilogue is the same
//     fmaf(acc, scale, shift) (+ residual) -> PReLU, so the result is bit-identical to that path;
//   * operands are swapped (weights = A, pixels = B) so a lane ends up with 4 consecutive channels of one pixel and
//     residual / output move as 8-byte vectors.
// LDS rows (128 B per pixel / per 64 weights) are XOR-swizzled by (row & 7) on their 16-byte chunks.
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kConvThreads = 512, kConvWaves = 8;
constexpr int kC = 64;                         // channels in == channels out
constexpr int kKW = 9 * kC;                    // 576

__global__ __launch_bounds__(kConvThreads) void conv3x3_c64_kernel(const bf16_t* __restrict__ x, int imgs, int H, int W,
                                                                   const bf16_t* __restrict__ w, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ slope,
                                                                   const bf16_t* residual, bf16_t* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = H * W;
  char* simg = smem;                                          // [HW + 1][128 B]; pixel HW is all zero (padding taps)
  char* sw = smem + (size_t)(HW + 1) * 128;                   // [64][1152 B]
  float* spar = reinterpret_cast<float*>(sw + kC * kKW * 2);  // scale[64] shift[64] slope[64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int nchunk = HW * 8;                                  // 16-byte chunks per image
  constexpr int kPre = 8;                                     // prefetch registers: up to 8 chunks per thread (HW <= 512)

  // ---- once per workgroup: weights, parameters, the zero pixel ----
  for (int i = tid; i < kC * (kKW / 8); i += kConvThreads) {
    const int n = i / (kKW / 8), c = i - n * (kKW / 8);
    const Chunk16 v = *reinterpret_cast<const Chunk16*>(w + (size_t)n * kKW + c * 8);
    *reinterpret_cast<Chunk16*>(sw + n * (kKW * 2) + (((c & ~7) | ((c ^ n) & 7)) << 4)) = v;
  }
  if (tid < 64) { spar[tid] = scale[tid]; spar[64 + tid] = shift[tid]; spar[128 + tid] = slope ? slope[tid] : 1.f; }
  if (tid < 8) { Chunk16 z; z.v = u32x4{0u, 0u, 0u, 0u}; *reinterpret_cast<Chunk16*>(simg + (size_t)HW * 128 + tid * 16) = z; }

  // next image, fetched into registers while the current one is computed.  (Plain unrolled loops over a vector array: as
  // lambdas over a struct array the buffer stayed in scratch, turning the prefetch into load - wait - spill.)
  u32x4 pre[kPre];
#define OAVSR_CONV_PREFETCH(IMG)                                                                       \
  {                                                                                                    \
    const bf16_t* src_ = x + (size_t)(IMG) * HW * kC;                                                  \
    _Pragma("unroll") for (int j = 0; j < kPre; ++j) {                                                 \
      const int i = tid + j * kConvThreads; /* out-of-range slots re-read chunk 0, never committed */  \
      pre[j] = *reinterpret_cast<const u32x4*>(src_ + (size_t)(i < nchunk ? i : 0) * 8);              \
    }                                                                                                  \
  }
#define OAVSR_CONV_COMMIT()                                                                            \
  _Pragma("unroll") for (int j = 0; j < kPre; ++j) {                                                   \
    const int i = tid + j * kConvThreads;                                                              \
    if (i < nchunk) { const int p_ = i >> 3, c_ = i & 7; *reinterpret_cast<u32x4*>(simg + p_ * 128 + ((c_ ^ (p_ & 7)) << 4)) = pre[j]; } \
  }

  // the 4 pixel tiles of this wave: output pixel q = (wave*4 + i)*16 + r16
  int qy[4], qx[4], qv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = (wave * 4 + i) * 16 + r16;
    qv[i] = q < HW;
    const int qq = qv[i] ? q : 0;
    qy[i] = qq / W; qx[i] = qq - qy[i] * W;
  }

  int img = blockIdx.x;
  if (img < imgs) OAVSR_CONV_PREFETCH(img)
  for (; img < imgs; img += gridDim.x) {
    __syncthreads();                                          // every wave is done with the previous image
    OAVSR_CONV_COMMIT()
    __syncthreads();
    if (img + (int)gridDim.x < imgs) OAVSR_CONV_PREFETCH(img + gridDim.x)

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3 - 1, kx = tap % 3 - 1;
      int pin[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = qy[i] + ky, ix = qx[i] + kx;
        pin[i] = (qv[i] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? iy * W + ix : HW;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int c = kb * 4 + g;                             // 16-byte chunk of the 64 channels
        Chunk16 wf[4], pf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = j * 16 + r16, wc = tap * 8 + c;
          wf[j] = *reinterpret_cast<const Chunk16*>(sw + n * (kKW * 2) + (((wc & ~7) | ((wc ^ n) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) pf[i] = *reinterpret_cast<const Chunk16*>(simg + pin[i] * 128 + ((c ^ (pin[i] & 7)) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j].v), __builtin_bit_cast(bf16x8, pf[i].v),
                                                                acc[i][j], 0, 0, 0);
      }
    }
    // D[channel = 16j + 4g + r][pixel = r16]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!qv[i]) continue;
      const size_t o = ((size_t)img * HW + (size_t)(wave * 4 + i) * 16 + r16) * kC;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = j * 16 + 4 * g;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(spar + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(spar + 64 + c0);
        const f32x4 sl = *reinterpret_cast<const f32x4*>(spar + 128 + c0);
        float res[4] = {0.f, 0.f, 0.f, 0.f};
        if (residual) {
          const bf16x4 rv = *reinterpret_cast<const bf16x4*>(residual + o + c0);
#pragma unroll
          for (int r = 0; r < 4; ++r) res[r] = (float)rv[r];
        }
        bf16x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[i][j][r], sc[r], sh[r]);
          v += res[r];
          if (slope) v = v >= 0.f ? v : sl[r] * v;
          h[r] = (bf16_t)v;
        }
        *reinterpret_cast<bf16x4*>(out + o + c0) = h;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// The 128-channel stage (layer2: three 128 -> 128 convs on 11 x 11 maps, resnet.py:61-74).  As an implicit GEMM (256 x 128 tiles) these gather
// every input pixel nine times through L2: 892 MB per launch, 176 us at 5 TB/s with the matrix pipe at 0.26 of its peak (profiles/r06_conv_bench.txt).
// Image-resident instead: a GROUP of G = 256 / (H W) images (2 x 121 pixels x 256 B = 61 KB) stays in LDS while the weights stream past it one TAP
// at a time (128 x 128 x 2 B = 32 KB from L2, double-buffered through registers: the next tap's rows are requested before the current tap's MFMAs
// and committed after them) -- memory sees each input byte once and 295 KB of weights per group.  Persistent workgroups of 8 waves walk the
// groups; the next group's pixels are prefetched into registers during the current one.  A wave owns 4 pixel tiles x 4 channel tiles; K runs
// tap-major then channels and the epilogue is the implicit GEMM's, so the result is BIT-IDENTICAL to that path (tests).
constexpr int kC2 = 128;
constexpr int kKW2 = 9 * kC2;                  // 1152

__global__ __launch_bounds__(kConvThreads) void conv3x3_c128_kernel(const bf16_t* __restrict__ x, int imgs, int H, int W, int G,
                                                                    const bf16_t* __restrict__ w, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, const float* __restrict__ slope,
                                                                    const bf16_t* residual, bf16_t* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = H * W, HWg = G * HW;                         // pixels per image / per group (<= 256)
  char* simg = smem;                                          // [HWg + 1][256 B]; pixel HWg is all zero (padding taps)
  char* sw = smem + (size_t)(HWg + 1) * 256;                  // [2][128][256 B]: two taps of weights
  float* spar = reinterpret_cast<float*>(sw + 2 * kC2 * 256); // scale[128] shift[128] slope[128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  constexpr int kPre = 8;                                     // image prefetch: up to 8 chunks per thread (256 pixels x 16 chunks)
  const int ngroups = (imgs + G - 1) / G;

  if (tid < kC2) { spar[tid] = scale[tid]; spar[kC2 + tid] = shift[tid]; spar[2 * kC2 + tid] = slope ? slope[tid] : 1.f; }
  if (tid < 16) { Chunk16 z; z.v = u32x4{0u, 0u, 0u, 0u}; *reinterpret_cast<Chunk16*>(simg + (size_t)HWg * 256 + tid * 16) = z; }

  u32x4 pre[kPre], wpre[4];
#define OAVSR_C128_IMG_PREFETCH(GRP)                                                                                 \
  {                                                                                                                  \
    const int npix_ = min(G, imgs - (GRP) * G) * HW;                                                                 \
    const bf16_t* src_ = x + (size_t)(GRP) * HWg * kC2;                                                              \
    _Pragma("unroll") for (int j = 0; j < kPre; ++j) {                                                               \
      const int i = tid + j * kConvThreads; /* past the group's last pixel: chunk 0 again, committed but never used */ \
      pre[j] = *reinterpret_cast<const u32x4*>(src_ + (size_t)(i < npix_ * 16 ? i : 0) * 8);                         \
    }                                                                                                                \
  }
#define OAVSR_C128_IMG_COMMIT()                                                                                      \
  _Pragma("unroll") for (int j = 0; j < kPre; ++j) {                                                                 \
    const int i = tid + j * kConvThreads;                                                                            \
    if (i < HWg * 16) { const int p_ = i >> 4, c_ = i & 15; *reinterpret_cast<u32x4*>(simg + p_ * 256 + ((c_ ^ (p_ & 15)) << 4)) = pre[j]; } \
  }
  // one tap of weights: 128 rows x 16 chunks = 2048 chunks, 4 per thread
#define OAVSR_C128_W_PREFETCH(TAP)                                                                                   \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
    const int i = tid + j * kConvThreads, n_ = i >> 4, c_ = i & 15;                                                  \
    wpre[j] = *reinterpret_cast<const u32x4*>(w + (size_t)n_ * kKW2 + (TAP) * kC2 + c_ * 8);                          \
  }
#define OAVSR_C128_W_COMMIT(BUF)                                                                                     \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
    const int i = tid + j * kConvThreads, n_ = i >> 4, c_ = i & 15;                                                  \
    *reinterpret_cast<u32x4*>(sw + (BUF) * (kC2 * 256) + n_ * 256 + ((c_ ^ (n_ & 15)) << 4)) = wpre[j];               \
  }

  // a wave owns 4 pixel tiles x 4 channel tiles (waves 0-3: channels 0..63, waves 4-7: 64..127): 8 fragment reads feed 16 MFMAs
  // group pixel q = ((wave & 3) * 4 + i) * 16 + r16 -> (image in group, y, x)
  const int pg = wave & 3, ng = wave >> 2;
  int qb[4], qy[4], qx[4], qin[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = (pg * 4 + i) * 16 + r16;
    qin[i] = q < HWg;
    const int qq = qin[i] ? q : 0;
    const int im = qq / HW, pos = qq - im * HW;
    qb[i] = im * HW; qy[i] = pos / W; qx[i] = pos - qy[i] * W;
  }

  int grp = blockIdx.x;
  if (grp < ngroups) OAVSR_C128_IMG_PREFETCH(grp)
  OAVSR_C128_W_PREFETCH(0)
  OAVSR_C128_W_COMMIT(0)                                      // tap 0 of the first group
  int stage = 0;                                              // taps done by this workgroup: buffer stage & 1 holds the current tap (9 per group: the parity flips)
  for (; grp < ngroups; grp += gridDim.x) {
    __syncthreads();                                          // every wave is done with the previous group's pixels (and tap 8's weights)
    OAVSR_C128_IMG_COMMIT()
    __syncthreads();
    if (grp + (int)gridDim.x < ngroups) OAVSR_C128_IMG_PREFETCH(grp + gridDim.x)
    const int npix = min(G, imgs - grp * G) * HW;             // valid pixels of this group

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      // the next tap's weights (tap 0 of the next group behind tap 8) into registers while this tap computes
      OAVSR_C128_W_PREFETCH(tap == 8 ? 0 : tap + 1)
      const int ky = tap / 3 - 1, kx = tap - (tap / 3) * 3 - 1;
      int pin[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = qy[i] + ky, ix = qx[i] + kx;
        pin[i] = (qin[i] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? qb[i] + iy * W + ix : HWg;
      }
      const char* swt = sw + (stage & 1) * (kC2 * 256);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int c = kb * 4 + g;                             // 16-byte chunk of the 128 channels
        Chunk16 pf[4], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pf[i] = *reinterpret_cast<const Chunk16*>(simg + pin[i] * 256 + ((c ^ (pin[i] & 15)) << 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = (ng * 4 + j) * 16 + r16;
          wf[j] = *reinterpret_cast<const Chunk16*>(swt + n * 256 + ((c ^ (n & 15)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j].v), __builtin_bit_cast(bf16x8, pf[i].v), acc[i][j], 0, 0, 0);
      }
      OAVSR_C128_W_COMMIT((stage + 1) & 1)                    // (that buffer's readers passed the barrier that ended the previous tap)
      ++stage;
      __syncthreads();
    }
    // D[channel = 16 (4 ng + j) + 4 g + r][pixel = r16]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = (pg * 4 + i) * 16 + r16;
      if (q >= npix) continue;
      const size_t o = ((size_t)grp * HWg + q) * kC2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = (ng * 4 + j) * 16 + 4 * g;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(spar + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(spar + kC2 + c0);
        const f32x4 sl = *reinterpret_cast<const f32x4*>(spar + 2 * kC2 + c0);
        float res[4] = {0.f, 0.f, 0.f, 0.f};
        if (residual) {
          const bf16x4 rv = *reinterpret_cast<const bf16x4*>(residual + o + c0);
#pragma unroll
          for (int r = 0; r < 4; ++r) res[r] = (float)rv[r];
        }
        bf16x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[i][j][r], sc[r], sh[r]);
          v += res[r];
          if (slope) v = v >= 0.f ? v : sl[r] * v;
          h[r] = (bf16_t)v;
        }
        *reinterpret_cast<bf16x4*>(out + o + c0) = h;
      }
    }
  }
}

}  // namespace

bool conv3x3_c64_eligible(int dtype, int H, int W, int Cin, int Cout, int ks, int stride) {
  return dtype == DT_BF16 && Cin == kC && Cout == kC && ks == 3 && stride == 1 && H * W <= 512 &&
         (size_t)(H * W + 1) * 128 + kC * kKW * 2 + 768 <= 160 * 1024;
}

int launch_conv3x3_c64(const void* x, int imgs, int H, int W, const void* w, const float* scale, const float* shift, const float* slope,
                       const void* residual, void* out, hipStream_t st) {
  if (imgs <= 0) return 0;
  const size_t smem = (size_t)(H * W + 1) * 128 + kC * kKW * 2 + 768;
  static bool attr = false;
  if (!attr) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  const int grid = imgs < ncu ? imgs : ncu;
  hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(grid), dim3(kConvThreads), smem, st, (const bf16_t*)x, imgs, H, W, (const bf16_t*)w, scale, shift,
                     slope, (const bf16_t*)residual, (bf16_t*)out);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

bool conv3x3_c128_eligible(int dtype, int H, int W, int Cin, int Cout, int ks, int stride) {
  static const bool on = knob("OAVSR_CONV_C128") != 0;
  return on && dtype == DT_BF16 && Cin == kC2 && Cout == kC2 && ks == 3 && stride == 1 && H * W <= 128 && H * W >= 16;
}

int launch_conv3x3_c128(const void* x, int imgs, int H, int W, const void* w, const float* scale, const float* shift, const float* slope,
                        const void* residual, void* out, hipStream_t st) {
  if (imgs <= 0) return 0;
  const int G = 256 / (H * W);                                 // images per group: <= 256 pixels = 16 pixel tiles = 2 per wave
  const size_t smem = (size_t)(G * H * W + 1) * 256 + 2 * kC2 * 256 + 3 * kC2 * 4;
  static bool attr = false;
  if (!attr) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  const int ngroups = (imgs + G - 1) / G;
  const int grid = ngroups < ncu ? ngroups : ncu;
  hipLaunchKernelGGL(conv3x3_c128_kernel, dim3(grid), dim3(kConvThreads), smem, st, (const bf16_t*)x, imgs, H, W, G, (const bf16_t*)w, scale, shift,
                     slope, (const bf16_t*)residual, (bf16_t*)out);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace oavsr
