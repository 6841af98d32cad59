// MFMA GEMM for gfx950:  C[M,N] = epi(A1.W1^T + A2.W2^T)  with W stored [N][K] (K contiguous).
//
// * wave64, v_mfma_f32_16x16x32_bf16 (bf16 mode) / v_mfma_f32_16x16x4_f32 (exact-f32 parity mode)
// * 256 threads = 4 waves per workgroup; tile BM x BN x 128 bytes of K; LDS double buffer, one barrier per k-tile
// * operands staged HBM -> VGPR -> LDS in 16-byte chunks; LDS rows are 128 B with the 16-B chunk index
//   XOR-swizzled by (row & 7) so the ds_read_b128 fragment reads are bank-conflict free
// * the A operand is produced by a gather functor: dense rows, NHWC conv2d (implicit im2col), token-major
//   conv1d (Whisper convs / AV-HuBERT grouped pos-conv) or the 5x7x7 video stem.  im2col is never
//   materialised in HBM.
// * optional second K segment (A2,W2) accumulates the LoRA term (x A^T)(s B)^T in the same accumulators
//   (Llama_LoRA.py:250-259, multihead_attention.py:485-494) -> "fused W + BA".
// * fused epilogue: per-column scale/shift (BatchNorm eval, bias), residual, GELU / ReLU / PReLU / SiLU.
#include <stdlib.h>
#include "kernels.h"

namespace oavsr {

namespace {

constexpr int kThreads = 256;
constexpr int kRowBytes = 128;   // bytes of K per LDS row (one k-tile)

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // lane (g = lane>>4) holds k = 4g..4g+3 of a 16-wide k block; MFMA step s consumes element s of every lane,
  // i.e. k-slot g <-> k = 4g+s for both operands (a permutation of the k sum, which is order-free).
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    f32x4 fa = __builtin_bit_cast(f32x4, a.v), fb = __builtin_bit_cast(f32x4, b.v);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], c, 0, 0, 0);
  }
};

__device__ __forceinline__ Chunk16 zero_chunk() { Chunk16 c; c.v = u32x4{0u, 0u, 0u, 0u}; return c; }
__device__ __forceinline__ Chunk16 ld_chunk(const void* p) { return *reinterpret_cast<const Chunk16*>(p); }

// Per-thread, per-staged-row state of the A gather (fixed over the k loop).
struct ARow {
  int valid;
  int64_t base;      // dense: element offset of the row; conv2d: image base element offset; conv1d: center_row
  int a, b, c;       // conv2d: iy0, ix0, -   conv1d: t_center, len_in, -    stem: t, len, (oy<<8)|ox
};

template <typename T, int AMODE>
__device__ __forceinline__ void init_arow(const GemmArgs& p, int m, int g, ARow& r) {
  r.valid = m < p.M;
  r.base = 0; r.a = r.b = r.c = 0;
  if (!r.valid) return;
  if (AMODE == A_DENSE) {
    r.base = (int64_t)m * p.lda + (int64_t)g * p.a_goff;
  } else if (AMODE == A_CONV2D) {
    int ohw = p.OH * p.OW;
    int img = m / ohw, rem = m - img * ohw;
    int oy = rem / p.OW, ox = rem - oy * p.OW;
    r.base = (int64_t)img * p.H * p.Wd * p.Cin;
    r.a = oy * p.stride - p.pad;
    r.b = ox * p.stride - p.pad;
  } else if (AMODE == A_CONV1D) {
    int4 ri = p.rowinfo[m];
    r.base = ri.x; r.a = ri.y; r.b = ri.z;
  } else {  // A_STEM
    int frame = m / 1936, rem = m - frame * 1936;
    int oy = rem / 44, ox = rem - oy * 44;
    int2 fi = p.frameinfo[frame];
    r.base = frame; r.a = fi.x; r.b = fi.y; r.c = (oy << 8) | ox;
  }
}

template <typename T, int AMODE>
__device__ __forceinline__ Chunk16 load_a(const GemmArgs& p, const ARow& r, int k /*element index of the chunk*/, int g) {
  constexpr int VEC = TypeInfo<T>::kVec;
  if (!r.valid) return zero_chunk();
  const T* A = reinterpret_cast<const T*>(p.A);
  if (AMODE == A_DENSE) {
    return ld_chunk(A + r.base + k);
  } else if (AMODE == A_CONV2D) {
    int tap = k / p.Cin, c = k - tap * p.Cin;
    int ky = tap / p.ksize, kx = tap - ky * p.ksize;
    int iy = r.a + ky, ix = r.b + kx;
    if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.Wd) return zero_chunk();
    return ld_chunk(A + r.base + ((int64_t)iy * p.Wd + ix) * p.Cin + c);
  } else if (AMODE == A_CONV1D) {
    if (k >= p.Kvalid) return zero_chunk();
    int tap = k / p.Cin, c = k - tap * p.Cin;
    int d = tap - p.pad;
    int tt = r.a + d;
    if ((unsigned)tt >= (unsigned)r.b) return zero_chunk();
    return ld_chunk(A + (r.base + d) * p.lda + (int64_t)g * p.a_goff + c);
  } else {  // A_STEM: f32 video in, 1 channel, taps (dt,ky,kx) = 5x7x7, stride (1,2,2), pad (2,3,3)  resnet.py:137
    const float* vid = reinterpret_cast<const float*>(p.A);
    float e[VEC];
    int oy = r.c >> 8, ox = r.c & 255;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      int kk = k + i;
      float v = 0.f;
      if (kk < 245) {
        int dt = kk / 49, rr = kk - dt * 49;
        int ky = rr / 7, kx = rr - ky * 7;
        int tt = r.a + dt - 2, iy = 2 * oy + ky - 3, ix = 2 * ox + kx - 3;
        if ((unsigned)tt < (unsigned)r.b && (unsigned)iy < 88u && (unsigned)ix < 88u)
          v = vid[(r.base + dt - 2) * 7744 + iy * 88 + ix];
      }
      e[i] = v;
    }
    return chunk_from_f32<T>(e);
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int AMODE>
__global__ __launch_bounds__(kThreads) void gemm_kernel(const GemmArgs p) {
  constexpr int VEC = TypeInfo<T>::kVec;          // elements per 16-B chunk
  constexpr int BKE = kRowBytes / (int)sizeof(T);  // k elements per tile
  constexpr int TM = BM / (WAVES_M * 16), TN = BN / (WAVES_N * 16);
  constexpr int NA = BM * 8 / kThreads, NB = BN * 8 / kThreads;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kStage = (BM + BN) * kRowBytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int ntn = (p.N + BN - 1) / BN;
  const int bid = blockIdx.x;
  const int mt = bid / ntn, nt_ = bid - mt * ntn;
  const int m0 = mt * BM, n0 = nt_ * BN;
  const int g = blockIdx.z;

  ARow arow[NA];
  int a_lds[NA], a_kc[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int id = tid + j * kThreads, row = id >> 3, kc = id & 7;
    init_arow<T, AMODE>(p, m0 + row, g, arow[j]);
    a_lds[j] = row * kRowBytes + ((kc ^ (row & 7)) << 4);
    a_kc[j] = kc * VEC;
  }
  int b_row[NB], b_lds[NB], b_kc[NB], b_valid[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    int id = tid + j * kThreads, row = id >> 3, kc = id & 7;
    b_valid[j] = (n0 + row) < p.N;
    b_row[j] = g * p.N + n0 + row;
    b_lds[j] = BM * kRowBytes + row * kRowBytes + ((kc ^ (row & 7)) << 4);
    b_kc[j] = kc * VEC;
  }
  const T* W1 = reinterpret_cast<const T*>(p.W);
  const T* W2 = reinterpret_cast<const T*>(p.W2);
  const T* A2 = reinterpret_cast<const T*>(p.A2);
  const int nt1 = p.K / BKE, ntot = nt1 + p.K2 / BKE;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Chunk16 ra[NA], rb[NB];
  auto gload = [&](int t) {
    if (t < nt1) {
      int k0 = t * BKE;
#pragma unroll
      for (int j = 0; j < NA; ++j) ra[j] = load_a<T, AMODE>(p, arow[j], k0 + a_kc[j], g);
#pragma unroll
      for (int j = 0; j < NB; ++j)
        rb[j] = b_valid[j] ? ld_chunk(W1 + (int64_t)b_row[j] * p.ldw + k0 + b_kc[j]) : zero_chunk();
    } else {
      int k0 = (t - nt1) * BKE;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int row = (tid + j * kThreads) >> 3;
        ra[j] = (m0 + row) < p.M ? ld_chunk(A2 + (int64_t)(m0 + row) * p.lda2 + k0 + a_kc[j]) : zero_chunk();
      }
#pragma unroll
      for (int j = 0; j < NB; ++j)
        rb[j] = b_valid[j] ? ld_chunk(W2 + (int64_t)b_row[j] * p.ldw2 + k0 + b_kc[j]) : zero_chunk();
    }
  };
  auto swrite = [&](int buf) {
    char* s = smem + buf * kStage;
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<Chunk16*>(s + a_lds[j]) = ra[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<Chunk16*>(s + b_lds[j]) = rb[j];
  };

  const int t0 = 0, t1 = ntot;      // (split-K lives in the LDS-DMA kernels only, gemm2.hip)
  const int frow = lane & 15, fg = lane >> 4;
  gload(t0);
  swrite(0);
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    if (t + 1 < t1) gload(t + 1);
    const char* sA = smem + cur * kStage;
    const char* sB = sA + BM * kRowBytes;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      Chunk16 af[TM], bf[TN];
      const int chunk = kb * 4 + fg;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int row = wm * (TM * 16) + i * 16 + frow;
        af[i] = *reinterpret_cast<const Chunk16*>(sA + row * kRowBytes + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int row = wn * (TN * 16) + j * 16 + frow;
        bf[j] = *reinterpret_cast<const Chunk16*>(sB + row * kRowBytes + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
    }
    if (t + 1 < t1) swrite(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = 4*(lane>>4) + reg -------------
  const T* R = reinterpret_cast<const T*>(p.residual);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (TN * 16) + j * 16 + frow;
    if (n >= p.N) continue;
    const int ng = g * p.N + n;
    const float sc = p.col_scale ? p.col_scale[ng] : 1.f;
    const float bi = p.col_bias ? p.col_bias[ng] : 0.f;
    const float sl = (p.act == ACT_PRELU) ? p.prelu[ng] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * (TM * 16) + i * 16 + fg * 4 + r;
        if (m >= p.M) continue;
        float y = fmaf(acc[i][j][r], sc, bi);
        float res = R ? to_f32(R[(int64_t)m * p.ldr + ng]) : 0.f;
        if (!p.res_after_act) y += res;
        switch (p.act) {
          case ACT_GELU: y = gelu_act<T>(y); break;
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

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int AMODE>
int launch_one(const GemmArgs& a, hipStream_t st) {
  auto kern = gemm_kernel<T, BM, BN, WAVES_M, WAVES_N, AMODE>;
  constexpr int smem = 2 * (BM + BN) * kRowBytes;
  static bool attr_set = false;
  if (!attr_set) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  int64_t nblk = (int64_t)ceil_div(a.M, BM) * ceil_div(a.N, BN);
  OAVSR_REQUIRE(nblk < (1ll << 31), "gemm: grid too large");
  dim3 grid((unsigned)nblk, 1, a.groups);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int AMODE>
int launch_tile(const GemmArgs& a, hipStream_t st) {
  if (a.N <= 64) return launch_one<T, 128, 64, 2, 2, AMODE>(a, st);
  return launch_one<T, 128, 128, 2, 2, AMODE>(a, st);
}

template <typename T>
int launch_mode(const GemmArgs& a, hipStream_t st) {
  switch (a.amode) {
    case A_DENSE: return launch_tile<T, A_DENSE>(a, st);
    case A_CONV2D: return launch_tile<T, A_CONV2D>(a, st);
    case A_CONV1D: return launch_tile<T, A_CONV1D>(a, st);
    case A_STEM: return launch_tile<T, A_STEM>(a, st);
  }
  set_last_error("gemm: bad amode %d", a.amode);
  return -1;
}

}  // namespace

bool gemm_strip_rows(int M) {
  static const bool on = knob("OAVSR_DECODE_STRIPS") != 0;
  return on && M >= 1 && M <= kStripRows;
}
bool gemm_strip_selected(const GemmArgs& a) { return gemm_strip_rows(a.M) && gemm_strip_eligible(a); }

int launch_gemm(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return 0;
  const int esz = a.dtype == DT_F32 ? 4 : 2;
  const int bke = kRowBytes / esz, vec = 16 / esz;
  OAVSR_REQUIRE(a.dtype == DT_F32 || a.dtype == DT_BF16, "gemm: bad dtype %d", a.dtype);
  OAVSR_REQUIRE(a.K > 0 && a.K % bke == 0, "gemm: K=%d must be a positive multiple of %d", a.K, bke);
  OAVSR_REQUIRE(a.K2 % bke == 0, "gemm: K2=%d must be a multiple of %d", a.K2, bke);
  OAVSR_REQUIRE(a.K2 == 0 || a.amode == A_DENSE, "gemm: second K segment needs dense A");
  OAVSR_REQUIRE(a.ldw % vec == 0 && ((uintptr_t)a.W & 15) == 0, "gemm: W must be 16-byte aligned (ldw=%lld)", (long long)a.ldw);
  if (a.amode == A_DENSE)
    OAVSR_REQUIRE(a.lda % vec == 0 && ((uintptr_t)a.A & 15) == 0 && a.a_goff % vec == 0, "gemm: A must be 16-byte aligned (lda=%lld)", (long long)a.lda);
  if (a.K2) {
    OAVSR_REQUIRE(a.lda2 % vec == 0 && a.ldw2 % vec == 0 && ((uintptr_t)a.A2 & 15) == 0 && ((uintptr_t)a.W2 & 15) == 0,
                  "gemm: segment-2 operands must be 16-byte aligned");
  }
  if (a.amode == A_CONV2D) OAVSR_REQUIRE(a.Cin % vec == 0 && a.K == a.ksize * a.ksize * a.Cin, "gemm/conv2d: bad geometry");
  if (a.amode == A_CONV1D) OAVSR_REQUIRE(a.Cin % vec == 0 && a.rowinfo && a.Kvalid == a.taps * a.Cin && a.Kvalid <= a.K, "gemm/conv1d: bad geometry");
  if (a.amode == A_STEM) OAVSR_REQUIRE(a.frameinfo && a.K == 256 && a.M % 1936 == 0, "gemm/stem: bad geometry");
  OAVSR_REQUIRE(a.act != ACT_PRELU || a.prelu, "gemm: PReLU needs slopes");
  OAVSR_REQUIRE(!a.row_sumsq || (a.amode == A_DENSE && a.groups == 1 && a.M <= 128 && a.N >= 64 && a.row_sumsq_tiles >= 1 && a.row_dim > 0 && (a.ksplit > 1 || a.decode_tiled)),
                "gemm: the row factor (row_sumsq) is an epilogue of the decode-shaped launches (dense, M <= 128, N >= 64, decode_tiled)");
  OAVSR_REQUIRE(!a.ng_out || a.ksplit > 1, "gemm: the next-norm outputs (ng_out) are written by the split-K reduce only");
  // <= 16 rows: the weight-streaming strips (decode_strip.hip) -- the tiled kernels' arithmetic without slabs, counters or a reduce launch
  if (gemm_strip_selected(a)) return launch_gemm_strip(a, st);
  // the strip layout of the sums of squares: written by the strips only; read by the strips and by the un-split vocabulary-sized tiles (128 x 128 / 128 x 256)
  OAVSR_REQUIRE(!a.rowsq_strips, "gemm: the strip layout of the sums of squares is written by the <= %d-row strip kernels (M = %d)", kStripRows, a.M);
  OAVSR_REQUIRE(!a.row_sumsq_form || (a.M <= kStripRows && a.ksplit <= 1 && a.amode == A_DENSE && a.groups == 1 && a.N >= 256 * 128 && a.row_sumsq_tiles <= 128),
                "gemm: the strip layout of the sums of squares is read by the strips and by the vocabulary-sized tiles (M = %d, N = %d)", a.M, a.N);
  if (a.ksplit > 1) {
    // split-K: the 128 x 64 LDS-DMA tile only -- the decode-shaped GEMMs, <= 128 rows per launch -- with the reduce inside the launch (a.sync:
    // arrival counters) or as a second launch of the same reduce code (gemm2.hip)
    const int vec_t = a.dtype == DT_F32 ? 4 : 8;
    OAVSR_REQUIRE(a.partial && a.groups == 1 && a.amode == A_DENSE && !a.swiglu_pairs, "gemm: split-K needs slabs, dense A, no groups");
    OAVSR_REQUIRE(a.M <= 128 && a.N >= 64, "gemm: split-K serves decode-shaped launches (M <= 128, N >= 64), got %d x %d", a.M, a.N);
    OAVSR_REQUIRE(a.N % 16 == 0 && ((uintptr_t)a.partial & 15) == 0 && ((uintptr_t)a.C & 15) == 0 && a.ldc % (a.out_f32 ? 4 : vec_t) == 0 &&
                  (!a.residual || (a.ldr % vec_t == 0 && ((uintptr_t)a.residual & 15) == 0)), "gemm: split-K reduce needs N %% 16 == 0 and 16-byte rows");
    const int bn = 64;
    OAVSR_REQUIRE(ceil_div(a.N, bn) <= kSyncInts, "gemm: split-K launch of %d column tiles exceeds the arrival counters", ceil_div(a.N, bn));
    OAVSR_REQUIRE((int64_t)a.ksplit * a.M * a.N * 4 < (1ll << 31), "gemm: split-K slabs exceed 2 GiB");
    if (a.ng_out) {
      OAVSR_REQUIRE(a.ng_gain && a.ng_rowsq && !a.out_f32 && a.act == ACT_NONE && a.N % bn == 0 && a.ng_ld % vec_t == 0 && ((uintptr_t)a.ng_out & 15) == 0,
                    "gemm: next-norm outputs need a plain epilogue, whole column tiles and 16-byte rows");
    }
    return launch_gemm_dma(a, st);
  }
  if (a.swiglu_pairs) OAVSR_REQUIRE(gemm_dma_swiglu_ok(a), "gemm: fused SwiGLU epilogue unavailable for this shape (check gemm_dma_swiglu_ok)");
  if (gemm_dma_eligible(a)) return launch_gemm_dma(a, st);
  return a.dtype == DT_F32 ? launch_mode<float>(a, st) : launch_mode<bf16_t>(a, st);
}

}  // namespace oavsr
