// Pipelined MFMA GEMM for gfx950 (the encoder / prefill workhorse):  C[M,N] = epi(A1.W1^T + A2.W2^T).
//
// Same math, operand layouts, gather functors (dense / NHWC conv2d / token-major conv1d) and fused epilogue as gemm.hip,
// but built around LDS-DMA (`global_load_lds_dwordx4`): tiles go HBM -> LDS without touching VGPRs, so three 48-KiB
// stages fit (144 of the CU's 160 KiB) and TWO k-tiles stay in flight across the single barrier per k-tile
// (counted `s_waitcnt vmcnt(N)`, raw `s_barrier`; never vmcnt(0) inside the loop).
//   * workgroup = 512 threads = 8 waves (4 x 2), tile 256 x 128 x 128 B of K, wave tile 64 x 64 (4 x 4 MFMA 16x16)
//   * the DMA writes LDS linearly (wave-uniform base + lane*16), so the XOR swizzle that makes the ds_read_b128 fragment
//     reads conflict-free is applied to the per-lane SOURCE address (chunk ^ (row & 7)) and again on the read
//   * out-of-range rows / conv padding read a 16-byte zero page instead of branching
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kThreads = 512;
constexpr int kRowBytes = 128;
constexpr int BM = 256, BN = 128;
constexpr int kStageBytes = (BM + BN) * kRowBytes;      // 48 KiB
constexpr int kStages = 3;
constexpr int NLD = (BM + BN) * 8 / kThreads;           // 6 DMA instructions per thread per k-tile
constexpr int NLA = BM * 8 / kThreads;                  // 4 of them fetch A rows

__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];

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

struct ARow2 {
  int valid;
  int64_t base;   // dense: element offset of the row (incl. group offset); conv2d: image base; conv1d: center row
  int a, b;       // conv2d: iy0, ix0; conv1d: t_center, len_in
};

template <typename T, int AMODE>
__device__ __forceinline__ void init_arow2(const GemmArgs& p, int m, int g, ARow2& r) {
  r.valid = m < p.M; r.base = 0; r.a = r.b = 0;
  if (!r.valid) return;
  if (AMODE == A_DENSE) {
    r.base = (int64_t)m * p.lda + (int64_t)g * p.a_goff;
  } else if (AMODE == A_CONV2D) {
    int ohw = p.OH * p.OW;
    int img = m / ohw, rem = m - img * ohw;
    int oy = rem / p.OW, ox = rem - oy * p.OW;
    r.base = (int64_t)img * p.H * p.Wd * p.Cin;
    r.a = oy * p.stride - p.pad; r.b = ox * p.stride - p.pad;
  } else {
    int4 ri = p.rowinfo[m];
    r.base = ri.x; r.a = ri.y; r.b = ri.z;
  }
}

// address of the 16-byte chunk that holds K elements [k, k + VEC) of gathered row r (or the zero page)
template <typename T, int AMODE>
__device__ __forceinline__ const void* a_src(const GemmArgs& p, const ARow2& r, int k, int g) {
  const T* A = reinterpret_cast<const T*>(p.A);
  if (!r.valid) return g_zero_page;
  if (AMODE == A_DENSE) return A + r.base + k;
  if (AMODE == A_CONV2D) {
    int tap = k / p.Cin, c = k - tap * p.Cin;
    int ky = tap / p.ksize, kx = tap - ky * p.ksize;
    int iy = r.a + ky, ix = r.b + kx;
    if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.Wd) return g_zero_page;
    return A + r.base + ((int64_t)iy * p.Wd + ix) * p.Cin + c;
  }
  if (k >= p.Kvalid) return g_zero_page;
  int tap = k / p.Cin, c = k - tap * p.Cin;
  int d = tap - p.pad, tt = r.a + d;
  if ((unsigned)tt >= (unsigned)r.b) return g_zero_page;
  return A + (r.base + d) * p.lda + (int64_t)g * p.a_goff + c;
}

__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, int AMODE>
__global__ __launch_bounds__(kThreads) void gemm_dma_kernel(const GemmArgs p) {
  constexpr int VEC = TypeInfo<T>::kVec;
  constexpr int BKE = kRowBytes / (int)sizeof(T);
  constexpr int TM = 4, TN = 4;
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int bid = blockIdx.x;
  const int mt = bid / ntn, nt_ = bid - mt * ntn;
  const int m0 = mt * BM, n0 = nt_ * BN;
  const int g = blockIdx.z;

  // DMA instruction j of this wave fills the 1-KiB LDS segment s = j*8 + wave = rows s*8 .. s*8+7 of the [A | W] image;
  // lane i lands at physical chunk (i & 7) of row s*8 + (i >> 3) and therefore fetches logical chunk (i & 7) ^ (i >> 3).
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  ARow2 arow[NLA];
#pragma unroll
  for (int j = 0; j < NLA; ++j) init_arow2<T, AMODE>(p, m0 + (j * 8 + wave) * 8 + lrow, g, arow[j]);
  int64_t wbase[NLD - NLA]; int wvalid[NLD - NLA];
#pragma unroll
  for (int j = 0; j < NLD - NLA; ++j) {
    int n = n0 + ((j + NLA) * 8 + wave) * 8 + lrow - BM;
    wvalid[j] = n < p.N;
    wbase[j] = (int64_t)(g * p.N + n);
  }
  const T* W1 = reinterpret_cast<const T*>(p.W);
  const T* W2 = reinterpret_cast<const T*>(p.W2);
  const T* A2 = reinterpret_cast<const T*>(p.A2);
  const int nt1 = p.K / BKE, ntot = nt1 + p.K2 / BKE;

  auto issue = [&](int t) {
    char* st = smem + (t % kStages) * kStageBytes;
    if (t < nt1) {
      const int k = t * BKE + lchunk * VEC;
#pragma unroll
      for (int j = 0; j < NLA; ++j) dma16(a_src<T, AMODE>(p, arow[j], k, g), st + (j * 8 + wave) * 1024);
#pragma unroll
      for (int j = 0; j < NLD - NLA; ++j)
        dma16(wvalid[j] ? (const void*)(W1 + wbase[j] * p.ldw + k) : (const void*)g_zero_page, st + ((j + NLA) * 8 + wave) * 1024);
    } else {
      const int k = (t - nt1) * BKE + lchunk * VEC;
#pragma unroll
      for (int j = 0; j < NLA; ++j) {
        const int m = m0 + (j * 8 + wave) * 8 + lrow;
        dma16(m < p.M ? (const void*)(A2 + (int64_t)m * p.lda2 + k) : (const void*)g_zero_page, st + (j * 8 + wave) * 1024);
      }
#pragma unroll
      for (int j = 0; j < NLD - NLA; ++j)
        dma16(wvalid[j] ? (const void*)(W2 + wbase[j] * p.ldw2 + k) : (const void*)g_zero_page, st + ((j + NLA) * 8 + wave) * 1024);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  if (ntot > 1) issue(1);
  const int frow = lane & 15, fg = lane >> 4;
  for (int t = 0; t < ntot; ++t) {
    // tile t has landed once at most one younger tile (NLD loads) is still outstanding
    if (t + 1 < ntot) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < ntot) issue(t + 2);      // stage (t+2)%3 == (t-1)%3: every wave finished reading it before the barrier
    const char* sA = smem + (t % kStages) * kStageBytes;
    const char* sB = sA + BM * kRowBytes;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      Chunk16 af[TM], bf[TN];
      const int chunk = kb * 4 + fg;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int row = wm * 64 + i * 16 + frow;
        af[i] = *reinterpret_cast<const Chunk16*>(sA + row * kRowBytes + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int row = wn * 64 + j * 16 + frow;
        bf[j] = *reinterpret_cast<const Chunk16*>(sB + row * kRowBytes + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma2<T>::run(af[i], bf[j], acc[i][j]);
    }
    asm volatile("" ::: "memory");
  }

  const T* R = reinterpret_cast<const T*>(p.residual);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * 64 + j * 16 + frow;
    if (n >= p.N) continue;
    const int ng = g * p.N + n;
    const float sc = p.col_scale ? p.col_scale[ng] : 1.f;
    const float bi = p.col_bias ? p.col_bias[ng] : 0.f;
    const float sl = (p.act == ACT_PRELU) ? p.prelu[ng] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + i * 16 + fg * 4 + r;
        if (m >= p.M) continue;
        float y = acc[i][j][r] * sc + bi;
        float res = R ? to_f32(R[(int64_t)m * p.ldr + ng]) : 0.f;
        if (!p.res_after_act) y += res;
        switch (p.act) {
          case ACT_GELU: y = gelu_erf(y); break;
          case ACT_RELU: y = fmaxf(y, 0.f); break;
          case ACT_PRELU: y = y >= 0.f ? y : sl * y; break;
          case ACT_SILU: y = silu(y); break;
          default: break;
        }
        if (p.res_after_act) y += res;
        if (p.out_f32) reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + ng] = y;
        else reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + ng] = from_f32<T>(y);
      }
    }
  }
}

template <typename T, int AMODE>
int launch_dma(const GemmArgs& a, hipStream_t st) {
  auto kern = gemm_dma_kernel<T, AMODE>;
  constexpr int smem = kStages * kStageBytes;
  static bool attr_set = false;
  if (!attr_set) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  int64_t nblk = (int64_t)ceil_div(a.M, BM) * ceil_div(a.N, BN);
  dim3 grid((unsigned)nblk, 1, a.groups);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int launch_dma_mode(const GemmArgs& a, hipStream_t st) {
  switch (a.amode) {
    case A_DENSE: return launch_dma<T, A_DENSE>(a, st);
    case A_CONV2D: return launch_dma<T, A_CONV2D>(a, st);
    case A_CONV1D: return launch_dma<T, A_CONV1D>(a, st);
  }
  return -1;
}

}  // namespace

bool gemm_dma_eligible(const GemmArgs& a) {
  if (a.amode == A_STEM || a.N < 128) return false;
  const int64_t nblk = (int64_t)ceil_div(a.M, BM) * ceil_div(a.N, BN) * a.groups;
  return nblk >= 192;      // enough 256x128 tiles for (nearly) every CU; smaller grids keep the 128x128 kernel at 2 WG/CU
}

int launch_gemm_dma(const GemmArgs& a, hipStream_t st) {
  return a.dtype == DT_F32 ? launch_dma_mode<float>(a, st) : launch_dma_mode<bf16_t>(a, st);
}

}  // namespace oavsr
