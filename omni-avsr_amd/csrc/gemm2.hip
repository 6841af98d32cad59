// Pipelined MFMA GEMM for gfx950 (the encoder / prefill workhorse):  C[M,N] = epi(A1.W1^T + A2.W2^T).
//
// Same math, operand layouts, gather functors (dense / NHWC conv2d / token-major conv1d) and fused epilogue as gemm.hip,
// but built around LDS-DMA (`global_load_lds_dwordx4`): tiles go HBM -> LDS without touching VGPRs, one raw `s_barrier` per
// k-tile, counted `s_waitcnt vmcnt(N)` where more than one k-tile is in flight.  Wave tile 64 x 64 (4 x 4 MFMA 16x16) throughout;
// tile shapes (launch_dma_mode): 256 x 256 / 16 waves / 2 stages (the workhorse), 256 x 128 / 8 waves / 3 stages (mid-size grids,
// conv gathers, split-K), 128 x 64, 128 x 128 and 128 x 256 / 4-8 waves (<= 128 rows: beam-search decode; small GEMMs).
//   * the DMA writes LDS linearly (wave-uniform base + lane*16), so the XOR swizzle that makes the ds_read_b128 fragment
//     reads conflict-free is applied to the per-lane SOURCE address (chunk ^ (row & 7)) and again on the read
//   * out-of-range rows / conv padding read a 16-byte zero page instead of branching
#include <stdlib.h>
#include "kernels.h"
#include "epilogue.h"

namespace oavsr {
namespace {

constexpr int kRowBytes = 128;

__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];

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
__device__ __forceinline__ const void* a_src(const GemmArgs& p, const ARow2& r, int k, int g, const void* g_zero_page) {
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

// conv2d / conv1d with C_in a multiple of the k-tile: the whole k-tile lies inside ONE tap, so (ky, kx) / the time offset are
// wave-uniform scalars computed once per k-tile (kt0 = first element of the k-tile, kc = this lane's offset inside it)
template <typename T, int AMODE>
__device__ __forceinline__ const void* a_src_tap(const GemmArgs& p, const ARow2& r, int tap, int c, int g, const void* g_zero_page) {
  const T* A = reinterpret_cast<const T*>(p.A);
  if (!r.valid) return g_zero_page;
  if (AMODE == A_CONV2D) {
    const int ky = tap / p.ksize, kx = tap - ky * p.ksize;           // scalar
    const int iy = r.a + ky, ix = r.b + kx;
    if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.Wd) return g_zero_page;
    return A + r.base + ((int64_t)iy * p.Wd + ix) * p.Cin + c;
  }
  const int d = tap - p.pad, tt = r.a + d;
  if (tap >= p.taps || (unsigned)tt >= (unsigned)r.b) return g_zero_page;
  return A + (r.base + d) * p.lda + (int64_t)g * p.a_goff + c;
}

// ---- hand-off between workgroups of ONE launch (the split-K reduce below).  MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement &
// inter-workgroup visibility", first row of the table of measured forms: every byte stored write-through (sc1), every storing wave drains
// its stores, one lane per workgroup adds to the tile's agent-scope counter behind a workgroup barrier, the workgroup whose add came
// last -- told by the value the add returned -- reads the bytes with sc1 loads (L1 bypassed) after a barrier that lane has joined.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t raw_rsrc2(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, byte_off, 0, 16);      // aux bit 4 = sc1 on gfx950
}
__device__ __forceinline__ f32x4 ld16_sc1f(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16));
}

__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---- split-K reduce of ONE 64-column tile strip: rows [m0, m0 + NI * nthr / 4) x columns [n0, n0 + 64) of C.  A thread owns NI segments of
//      16 columns (segment w = tid + i * nthr -> row w / 4, columns 16 (w % 4)); everything they need is requested before the first use --
//      the residual rows, then the slabs of ALL its segments four at a time, so a slab group costs one memory round trip -- the slabs are
//      added in slab order (the result does not depend on who reduces, nor on the row count), then the epilogue: row factor (rs[row in
//      strip], from the deferred RMSNorm's sums of squares), col_scale / bias, residual, activation, store; optionally the consumer's
//      RMSNorm operands (GemmArgs::ng_*): T(x * gain) and the row's sum of squares over the tile (16 columns per lane in column order, then a
//      fixed butterfly over the row's 4 lanes).  Called by the tile's last-arriving workgroup inside the GEMM launch (gemm_dma_kernel) or by
//      the stand-alone reduce kernel below: the same code, hence the same bits.
template <typename T, int NI>
__device__ __forceinline__ void splitk_reduce_tile(const GemmArgs& p, int m0, int n0, int nt_, int tid, int nthr, const float* rs) {
  constexpr int VEC = TypeInfo<T>::kVec;
  constexpr int SEG = 4;                                     // 16-column segments of a 64-column tile row: adjacent lanes
  const __amdgpu_buffer_rsrc_t prs = raw_rsrc2(p.partial);
  const unsigned KS = (unsigned)p.ksplit, sstride = (unsigned)((int64_t)p.M * p.N * 4);
  const T* Rr = reinterpret_cast<const T*>(p.residual);
  const bool want_ng = p.ng_out != nullptr;
  int mI[NI], ncI[NI], rI[NI];
  bool liveI[NI];
  unsigned baseI[NI];
  Chunk16 rraw[NI][16 / VEC];
  float y[NI][16];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int w = tid + i * nthr;
    rI[i] = w / SEG;
    mI[i] = m0 + rI[i]; ncI[i] = n0 + (w % SEG) * 16;
    liveI[i] = mI[i] < p.M && ncI[i] < p.N;
    baseI[i] = liveI[i] ? (unsigned)(((int64_t)mI[i] * p.N + ncI[i]) * 4) : 0u;
    if (liveI[i] && Rr) {
#pragma unroll
      for (int e = 0; e < 16; e += VEC) rraw[i][e / VEC] = *reinterpret_cast<const Chunk16*>(Rr + (int64_t)mI[i] * p.ldr + ncI[i] + e);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) y[i][e] = 0.f;
  }
  for (unsigned s = 0; s < KS; s += 4) {                  // four slabs' loads of every segment in flight, added in slab order
    f32x4 v[NI][4][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned sj = s + j < KS ? s + j : KS - 1;      // (past the last slab: a repeated load, not added)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[i][j][q] = liveI[i] ? ld16_sc1f(prs, baseI[i] + sj * sstride + q * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (s + j < KS) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { y[i][4 * q] += v[i][j][q][0]; y[i][4 * q + 1] += v[i][j][q][1]; y[i][4 * q + 2] += v[i][j][q][2]; y[i][4 * q + 3] += v[i][j][q][3]; }
        }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) segment_finish<T>(p, mI[i], ncI[i], nt_, liveI[i], y[i], rraw[i], p.row_sumsq ? rs[rI[i]] : 1.f, tid);
}

// The same reduce as a launch of its own (GemmArgs::sync == nullptr): one workgroup of 128 threads per (column tile, 32 rows).  Above a few
// dozen rows a tile's last-arriving workgroup becomes a serial tail (8 slabs x 97 rows x 64 columns: 200 KB through one workgroup), which
// this spreads over ~4 x as many workgroups; model.hip chooses by the row count -- speed only: the arithmetic is splitk_reduce_tile's.
template <typename T>
__global__ __launch_bounds__(128) void splitk_reduce_kernel(const GemmArgs p) {
  __shared__ float rs[32];
  const int nt_ = blockIdx.x, m0 = blockIdx.y * 32, n0 = nt_ * 64, tid = threadIdx.x;
  if (p.row_sumsq && tid < 32) rs[tid] = row_factor(p, m0 + tid);
  __syncthreads();
  splitk_reduce_tile<T, 1>(p, m0, n0, nt_, tid, 128, rs);
}

// BM x BN tile, kStages LDS stages (kStages - 1 k-tiles in flight), NWAVES waves as WAVES_M x (NWAVES / WAVES_M).
// DW: waves that issue the LDS-DMA loads (segments are dealt over them); the rest only compute (12-wave 256 x 192 tile: 56 segments = 8 x 7)
template <typename T, int AMODE, int BM, int BN, int kStages, int WAVES_M, int NWAVES, int DW = NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void gemm_dma_kernel(const GemmArgs p) {
  constexpr int kThreads = NWAVES * 64;
  constexpr int VEC = TypeInfo<T>::kVec;
  constexpr int BKE = kRowBytes / (int)sizeof(T);
  constexpr int WAVES_N = NWAVES / WAVES_M;
  constexpr int TM = BM / (WAVES_M * 16), TN = BN / (WAVES_N * 16);
  constexpr int kStageBytes = (BM + BN) * kRowBytes;
  static_assert((BM + BN) * 8 % (DW * 64) == 0 && BM * 8 % (DW * 64) == 0, "segments must deal evenly over the DMA waves");
  constexpr int NLD = (BM + BN) * 8 / (DW * 64);          // DMA instructions per issuing thread per k-tile
  constexpr int NLA = BM * 8 / (DW * 64);                 // ... of which fetch A rows
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations go to M0 without a VGPR detour
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const void* zp = g_zero_page;                          // source of out-of-range rows
  asm volatile("" : "+s"(zp));                           // opaque: held in SGPRs, not re-fetched from the GOT (s_load + s_waitcnt) for every DMA group
  const int ntn = (p.N + BN - 1) / BN;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so give every XCD a CONTIGUOUS
  // range of logical tiles (= whole M-tile rows): the A panel of a row is then fetched by one L2 instead of all eight.
  // Bijective for any grid size; affects speed only, never results.
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  // An XCD's contiguous range covers whole M-tile rows (A panel read once, W by every XCD) or, when M < N, whole N-tile
  // columns (W slice read once, A by every XCD): memory-side reads are A + 8 W vs 8 A + W bytes.
  const int ntm = (p.M + BM - 1) / BM;
  const bool nmajor = p.M < p.N;
  int mt = nmajor ? bid % ntm : bid / ntn, nt_ = nmajor ? bid / ntm : bid - (bid / ntn) * ntn;
  if (p.tile_order == 1) blocked_tile(bid, ntm, ntn, mt, nt_);
  const int m0 = mt * BM, n0 = nt_ * BN;
  const int g = blockIdx.z;

  // DMA instruction j of this wave fills the 1-KiB LDS segment s = j*8 + wave = rows s*8 .. s*8+7 of the [A | W] image;
  // lane i lands at physical chunk (i & 7) of row s*8 + (i >> 3) and therefore fetches logical chunk (i & 7) ^ (i >> 3).
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;
  ARow2 arow[NLA];
#pragma unroll
  for (int j = 0; j < NLA; ++j) init_arow2<T, AMODE>(p, m0 + (j * DW + wave) * 8 + lrow, g, arow[j]);
  int64_t wbase[NLD - NLA]; int wvalid[NLD - NLA];
#pragma unroll
  for (int j = 0; j < NLD - NLA; ++j) {
    int n = n0 + ((j + NLA) * DW + wave) * 8 + lrow - BM;
    wvalid[j] = n < p.N;
    wbase[j] = (int64_t)(g * p.N + n);
  }
  const T* W1 = reinterpret_cast<const T*>(p.W);
  const T* W2 = reinterpret_cast<const T*>(p.W2);
  const T* A2 = reinterpret_cast<const T*>(p.A2);
  const int nt1 = p.K / BKE, nall = nt1 + p.K2 / BKE;
  // split-K (gridDim.y > 1): this workgroup owns k-tiles [tbeg, tbeg + ntot) and stores raw fp32 partial sums
  const bool split = gridDim.y > 1;
  const int per = (nall + (int)gridDim.y - 1) / (int)gridDim.y;
  const int tbeg = split ? (int)blockIdx.y * per : 0;
  const int ntot = split ? (tbeg + per < nall ? per : (nall - tbeg > 0 ? nall - tbeg : 0)) : nall;

  const bool tap_uniform = AMODE != A_DENSE && p.Cin % BKE == 0;
  const char* tap_ptr[NLA];
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const T* A = reinterpret_cast<const T*>(p.A);
    if (AMODE == A_CONV2D) tap_ptr[j] = reinterpret_cast<const char*>(A + arow[j].base + ((int64_t)arow[j].a * p.Wd + arow[j].b) * p.Cin + lchunk * VEC);
    else if (AMODE == A_CONV1D) tap_ptr[j] = reinterpret_cast<const char*>(A + arow[j].base * p.lda + (int64_t)g * p.a_goff + lchunk * VEC);
    else tap_ptr[j] = nullptr;
  }
  auto issue = [&](int tl) {
    if (DW < NWAVES && wave >= DW) return;            // compute-only waves
    char* st = smem + (tl % kStages) * kStageBytes;
    const int t = tbeg + tl;
    if (t < nt1) {
      const int k = t * BKE + lchunk * VEC;
      if ((AMODE == A_CONV2D || AMODE == A_CONV1D) && tap_uniform) {
        // the whole k-tile lies inside ONE tap: (ky, kx) / the time offset and the tap's address delta are wave-uniform
        // scalars, computed once per k-tile; per row only the bounds test and one 64-bit add remain (tap_ptr: the row's
        // address at tap (0, 0) / time offset 0, hoisted out of the loop)
        const int kt0 = t * BKE, tap = kt0 / p.Cin, c0 = kt0 - tap * p.Cin;
        if (AMODE == A_CONV2D) {
          const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
          const int64_t soff = ((int64_t)(ky * p.Wd + kx) * p.Cin + c0) * (int64_t)sizeof(T);
#pragma unroll
          for (int j = 0; j < NLA; ++j) {
            const bool ok = arow[j].valid && (unsigned)(arow[j].a + ky) < (unsigned)p.H && (unsigned)(arow[j].b + kx) < (unsigned)p.Wd;
            dma16(ok ? (const void*)(tap_ptr[j] + soff) : zp, st + (j * DW + wave) * 1024);
          }
        } else {
          const int dlt = tap - p.pad;
          const int64_t soff = ((int64_t)dlt * p.lda + c0) * (int64_t)sizeof(T);
          const bool tap_ok = tap < p.taps;
#pragma unroll
          for (int j = 0; j < NLA; ++j) {
            const bool ok = arow[j].valid && tap_ok && (unsigned)(arow[j].a + dlt) < (unsigned)arow[j].b;
            dma16(ok ? (const void*)(tap_ptr[j] + soff) : zp, st + (j * DW + wave) * 1024);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NLA; ++j) dma16(a_src<T, AMODE>(p, arow[j], k, g, zp), st + (j * DW + wave) * 1024);
      }
#pragma unroll
      for (int j = 0; j < NLD - NLA; ++j)
        dma16(wvalid[j] ? (const void*)(W1 + wbase[j] * p.ldw + k) : zp, st + ((j + NLA) * DW + wave) * 1024);
    } else {
      const int k = (t - nt1) * BKE + lchunk * VEC;
#pragma unroll
      for (int j = 0; j < NLA; ++j) {
        const int m = m0 + (j * DW + wave) * 8 + lrow;
        dma16(m < p.M ? (const void*)(A2 + (int64_t)m * p.lda2 + k) : zp, st + (j * DW + wave) * 1024);
      }
#pragma unroll
      for (int j = 0; j < NLD - NLA; ++j)
        dma16(wvalid[j] ? (const void*)(W2 + wbase[j] * p.ldw2 + k) : zp, st + ((j + NLA) * DW + wave) * 1024);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < kStages - 1; ++t)
    if (t < ntot) issue(t);
  // Decode-shaped launches (the 128-row tiles): the rows' 1 / rms from the producer's partial sums of squares (GemmArgs::row_sumsq), one
  // row per thread, into LDS behind the k-tile stages -- the loads go out BEHIND the first k-tiles' DMA (already in flight), so waiting for
  // them costs nothing the first k-tile would not have waited for anyway; the epilogues read the factors after the k loop's barriers.
  constexpr bool kRowScale = AMODE == A_DENSE && BM == 128;
  float* rs_lds = reinterpret_cast<float*>(smem + kStages * kStageBytes);
  if constexpr (kRowScale) {
    if constexpr (BN >= 128) {
      // the vocabulary-sized launch behind a <= 16-row strip producer (lm_head of a small decode step): the rows' sums of squares come per
      // 16-column strip (GemmArgs::row_sumsq_form); all threads form the tile sums (s0 + s1) + (s2 + s3), one thread per row adds the tiles in order
      if (p.row_sumsq && p.row_sumsq_form) {
        float* ts = rs_lds + 256;                            // [M <= 16][tiles <= 128]
        const int nT = p.row_sumsq_tiles, ntq = p.M * nT;
        for (int idx = tid; idx < ntq; idx += kThreads) {
          const int t = idx / p.M, r = idx - t * p.M;
          const float q0 = p.row_sumsq[(int64_t)(4 * t) * p.M + r], q1 = p.row_sumsq[(int64_t)(4 * t + 1) * p.M + r];
          const float q2 = p.row_sumsq[(int64_t)(4 * t + 2) * p.M + r], q3 = p.row_sumsq[(int64_t)(4 * t + 3) * p.M + r];
          ts[r * nT + t] = (q0 + q1) + (q2 + q3);
        }
        __syncthreads();
        if (tid < p.M) {
          float tot = 0.f;
          for (int t = 0; t < nT; ++t) tot += ts[tid * nT + t];
          rs_lds[tid] = rsqrtf(tot / (float)p.row_dim + p.row_eps);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    if (p.row_sumsq && !p.row_sumsq_form && tid < BM) {
      rs_lds[tid] = row_factor(p, m0 + tid);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }

  const int frow = lane & 15, fg = lane >> 4;
  for (int t = 0; t < ntot; ++t) {
    // tile t has landed once at most (kStages - 2) younger tiles (NLD loads each) are still outstanding
    if (kStages >= 3 && t + kStages - 2 < ntot) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kStages - 2) * NLD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // stage (t + kStages - 1) % kStages held tile t-1: every wave finished reading it before the barrier
    if (t + kStages - 1 < ntot) issue(t + kStages - 1);
    const char* sA = smem + (t % kStages) * kStageBytes;
    const char* sB = sA + BM * kRowBytes;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      constexpr int TMH = (TM > 4 && TM % 4 == 0) ? 4 : TM;      // A fragments are consumed in groups of 4 row tiles (register budget)
      Chunk16 bf[TN];
      const int chunk = kb * 4 + fg;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int row = wn * (TN * 16) + j * 16 + frow;
        bf[j] = *reinterpret_cast<const Chunk16*>(sB + row * kRowBytes + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int ih = 0; ih < TM / TMH; ++ih) {
        Chunk16 af[TMH];
#pragma unroll
        for (int i = 0; i < TMH; ++i) {
          int row = wm * (TM * 16) + (ih * TMH + i) * 16 + frow;
          af[i] = *reinterpret_cast<const Chunk16*>(sA + row * kRowBytes + ((chunk ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TMH; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma2<T>::run(af[i], bf[j], acc[ih * TMH + i][j]);
      }
    }
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: each 16-row slab of the wave tile is transposed through a wave-private fp32 LDS patch so that a lane
  //      owns 16 consecutive columns of one row: residual / bias loads and the C stores are full 16-byte vectors and
  //      every output row segment is written as whole 128-byte lines (the MFMA C layout alone gives 2-byte stores).
  static_assert(TN == 4, "wave tile is 64 columns wide");
  __builtin_amdgcn_s_barrier();                      // every wave is done reading the last k-tile
  asm volatile("" ::: "memory");
  constexpr int PW = 68;                             // padded row (floats)
  float* sw = reinterpret_cast<float*>(smem) + wave * (16 * PW);
  const T* R = split ? nullptr : reinterpret_cast<const T*>(p.residual);
  const float* e_scale = split ? nullptr : p.col_scale;
  const float* e_bias = split ? nullptr : p.col_bias;
  const int e_act = split ? (int)ACT_NONE : p.act;
  const bool e_f32 = split || p.out_f32;
  void* e_C = split ? (void*)(p.partial + (int64_t)blockIdx.y * p.M * p.N) : p.C;
  const int64_t e_ldc = split ? (int64_t)p.N : p.ldc;
  // the tiles the decode-shaped GEMMs run on (<= 128 rows per launch: 128 x 64 / 128 x 128 / 128 x 256): row_scale, and -- on the one that
  // serves split-K launches -- the reduce inside the launch.  Compiled out of every other instantiation (registers: the 16-wave tile has none to spare).
  constexpr bool kFusedReduce = AMODE == A_DENSE && BM == 128 && BN == 64;
  const bool fused = kFusedReduce && split && p.sync != nullptr;      // the reduce happens below, in this launch: slabs go out write-through
  const __amdgpu_buffer_rsrc_t prs = raw_rsrc2(p.partial);
  const int rr = lane >> 2, c0 = (lane & 3) * 16;
  const int nb = n0 + wn * 64 + c0;                  // first of this lane's 16 columns
  const int ngb = g * p.N + nb;
  const bool vec_ok = nb + 16 <= p.N && (e_ldc % VEC) == 0 && ((uintptr_t)e_C & 15) == 0 && (ngb % 16) == 0 &&
                      (!R || ((p.ldr % VEC) == 0 && ((uintptr_t)R & 15) == 0));
  // (round 5) tiles with registers to spare (<= 8 waves, 2-byte elements) request what the slabs need from global memory -- the residual
  // segments of ALL their rows, the lane's 16 per-column scales / biases -- BEFORE the first slab instead of inside each (one exposed
  // load latency per tile instead of one per slab and array)
  constexpr bool kPre = sizeof(T) == 2 && NWAVES <= 8;
  const bool pfull = nb + 16 <= p.N;
  auto load4 = [&](const float* arr, int e, float dflt) -> f32x4 {
    if (!arr) return f32x4{dflt, dflt, dflt, dflt};
    if (pfull && (((uintptr_t)(arr + ngb)) & 15) == 0) return *reinterpret_cast<const f32x4*>(arr + ngb + e);
    f32x4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = nb + e + q < p.N ? arr[ngb + e + q] : dflt;
    return t;
  };
  Chunk16 rpre[kPre ? TM : 1][2];
  f32x4 cpre[kPre ? 4 : 1][2];
  const bool pre_r = kPre && R && vec_ok && !(kFusedReduce && fused);
  if constexpr (kPre) {
    if (pre_r) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mi = m0 + wm * (TM * 16) + i * 16 + rr;
        if (mi < p.M) {
          rpre[i][0] = *reinterpret_cast<const Chunk16*>(R + (int64_t)mi * p.ldr + ngb);
          rpre[i][1] = *reinterpret_cast<const Chunk16*>(R + (int64_t)mi * p.ldr + ngb + 8);
        }
      }
    }
    if (!(kFusedReduce && fused)) {
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) { cpre[e4][0] = load4(e_scale, e4 * 4, 1.f); cpre[e4][1] = load4(e_bias, e4 * 4, 0.f); }
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sw[(fg * 4 + r) * PW + j * 16 + frow] = acc[i][j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float y[16];
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(sw + rr * PW + c0 + e);
      y[e] = v[0]; y[e + 1] = v[1]; y[e + 2] = v[2]; y[e + 3] = v[3];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int m = m0 + wm * (TM * 16) + i * 16 + rr;
    if (m >= p.M) continue;
    if constexpr (kFusedReduce) if (fused) {      // raw fp32 partial sums, 16-byte write-through stores (N % 16 == 0, checked by the launcher)
      if (nb < p.N) {
        const unsigned off = (unsigned)((((int64_t)blockIdx.y * p.M + m) * p.N + nb) * 4);
#pragma unroll
        for (int e = 0; e < 16; e += 4) st16_sc1(prs, off + e * 4, f32x4{y[e], y[e + 1], y[e + 2], y[e + 3]});
      }
      continue;
    }
    if constexpr (kRowScale) if (!split && p.row_sumsq) {
      const float rs = rs_lds[m - m0];
#pragma unroll
      for (int e = 0; e < 16; ++e) y[e] *= rs;
    }
    float res[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) res[e] = 0.f;
    if (R) {
      if (kPre && pre_r) {
        chunk_to_f32<T>(rpre[kPre ? i : 0][0], res);
        chunk_to_f32<T>(rpre[kPre ? i : 0][1], res + 8);
      } else if (vec_ok) {
#pragma unroll
        for (int e = 0; e < 16; e += VEC) chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(R + (int64_t)m * p.ldr + ngb + e), res + e);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) if (nb + e < p.N) res[e] = to_f32(R[(int64_t)m * p.ldr + ngb + e]);
      }
    }
    // per-column parameters: held in registers by the tiles that prefetched them, otherwise re-read per slab (L1 hits), four columns at a
    // time to keep registers free: one 16-byte vector per array when the lane's 16 columns are all inside N and the array is 16-byte
    // aligned there, scalar loads on the ragged edge.
    // No activation (or ReLU after the residual): one path without the run-time selections below -- with nothing between them the order
    // of residual and activation does not matter, fma(y, scale, bias) + res is what both orders compute (res = 0.f without a residual:
    // the `+ 0.f` of the generic form) -- 2-3 VALU instructions per element instead of ~8
    const bool relu_after_res = e_act == ACT_RELU && !p.res_after_act;
    if (kPre && (e_act == ACT_NONE || relu_after_res)) {      // (the 16-wave tile has no registers for a second path)
#pragma unroll
      for (int e4 = 0; e4 < 16; e4 += 4) {
        const f32x4 csc = kPre ? cpre[kPre ? e4 / 4 : 0][0] : load4(e_scale, e4, 1.f), cbi = kPre ? cpre[kPre ? e4 / 4 : 0][1] : load4(e_bias, e4, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) y[e4 + q] = fmaf(y[e4 + q], csc[q], cbi[q]) + res[e4 + q];
      }
      if (relu_after_res) {
#pragma unroll
        for (int e = 0; e < 16; ++e) y[e] = fmaxf(y[e], 0.f);
      }
    } else {
#pragma unroll
      for (int e4 = 0; e4 < 16; e4 += 4) {
        const f32x4 csc = kPre ? cpre[kPre ? e4 / 4 : 0][0] : load4(e_scale, e4, 1.f), cbi = kPre ? cpre[kPre ? e4 / 4 : 0][1] : load4(e_bias, e4, 0.f);
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // one explicit FMA, identical in every GEMM variant, so an element has the same bits whichever tile shape computed it
          v[q] = fmaf(y[e4 + q], csc[q], cbi[q]);
          if (!p.res_after_act) v[q] += res[e4 + q];
        }
        switch (e_act) {                            // one uniform branch per four columns
          case ACT_GELU:
            gelu_act4<T>(v);
            break;
          case ACT_RELU:
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            break;
          case ACT_PRELU: {
            const f32x4 csl = load4(p.prelu, e4, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = v[q] >= 0.f ? v[q] : csl[q] * v[q];
            break;
          }
          case ACT_SILU:
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = silu(v[q]);
            break;
          default: break;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) y[e4 + q] = p.res_after_act ? v[q] + res[e4 + q] : v[q];
      }
    }
    if (p.swiglu_pairs && !split) {
      // gate / up weight rows are interleaved in blocks of 16: the lanes with an even (lane & 3) hold a gate block, their right
      // neighbour the matching up block of the same row.  Both are rounded to T first (what the unfused GEMM -> swiglu_kernel
      // sequence reads back), so the fused and the two-launch path give the same bits.
      float o[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float up = __shfl_down(y[e], 1, 64);
        o[e] = silu(to_f32(from_f32<T>(y[e]))) * to_f32(from_f32<T>(up));
      }
      if ((lane & 1) == 0 && nb + 32 <= p.N) {
        const int64_t oc = (int64_t)m * p.ldc + (ngb >> 5) * 16;
        if (p.out_f32) {
          float* Cf = reinterpret_cast<float*>(p.C) + oc;
#pragma unroll
          for (int e = 0; e < 16; e += 4) *reinterpret_cast<f32x4*>(Cf + e) = f32x4{o[e], o[e + 1], o[e + 2], o[e + 3]};
        } else {
          T* Ct = reinterpret_cast<T*>(p.C) + oc;
#pragma unroll
          for (int e = 0; e < 16; e += VEC) *reinterpret_cast<Chunk16*>(Ct + e) = chunk_from_f32<T>(o + e);
        }
      }
      continue;
    }
    if (e_f32) {
      float* Cf = reinterpret_cast<float*>(e_C) + (int64_t)m * e_ldc + ngb;
      if (nb + 16 <= p.N && (e_ldc % 4) == 0 && ((uintptr_t)e_C & 15) == 0 && (ngb % 4) == 0) {
#pragma unroll
        for (int e = 0; e < 16; e += 4) *reinterpret_cast<f32x4*>(Cf + e) = f32x4{y[e], y[e + 1], y[e + 2], y[e + 3]};
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) if (nb + e < p.N) Cf[e] = y[e];
      }
    } else {
      T* Ct = reinterpret_cast<T*>(e_C) + (int64_t)m * e_ldc + ngb;
      if (vec_ok) {
#pragma unroll
        for (int e = 0; e < 16; e += VEC) *reinterpret_cast<Chunk16*>(Ct + e) = chunk_from_f32<T>(y + e);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) if (nb + e < p.N) Ct[e] = from_f32<T>(y[e]);
      }
    }
  }

  // ---- split-K reduce inside the launch: the LAST workgroup to arrive at this output tile sums the slabs in slab order (fixed: the
  //      result does not depend on which workgroup came last, nor on the row count) and applies the epilogue; optionally it also prepares
  //      the consumer's RMSNorm (GemmArgs::ng_*).  Nobody ever waits: no assumption about co-residency or dispatch order.
  if constexpr (kFusedReduce) {
    if (fused) {
      int* sflag = reinterpret_cast<int*>(smem + NWAVES * (16 * PW) * 4);      // behind the waves' patches
      const int tile_id = mt * ntn + nt_;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's slab stores have left the chip's caches
      __syncthreads();
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(p.sync + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == (int)gridDim.y - 1;
        // The measured sc1-only hand-off (guide, first table row) holds at ONE workgroup per CU; this tile shares a CU with a second
        // workgroup and with other streams' kernels, so the last arriver also takes the agent-scope acquire (invalidate this CU's L1,
        // wait for it) before the barrier that releases the other waves' loads: ~1.7 us on one workgroup per tile.
        if (last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        *sflag = last;
      }
      __syncthreads();
      if (!*sflag) return;
      splitk_reduce_tile<T, BM * (BN / 16) / kThreads>(p, m0, n0, nt_, tid, kThreads, rs_lds);
      if (tid == 0) __hip_atomic_store(p.sync + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
  }
}

template <typename T, int AMODE, int BM, int BN, int kStages, int WAVES_M, int NWAVES, int DW = NWAVES>
int launch_dma(const GemmArgs& a, hipStream_t st) {
  auto kern = gemm_dma_kernel<T, AMODE, BM, BN, kStages, WAVES_M, NWAVES, DW>;
  // + the rows' 1 / rms (decode-shaped launches); the wide tiles (lm_head) also stage the strip layout's tile sums: 16 rows x 128 tiles
  constexpr int smem = kStages * (BM + BN) * kRowBytes + ((AMODE == A_DENSE && BM == 128) ? (BN >= 128 ? 1024 + 8192 : 1024) : 0);
  static bool attr_set = false;
  if (!attr_set) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  int64_t nblk = (int64_t)ceil_div(a.M, BM) * ceil_div(a.N, BN);
  dim3 grid((unsigned)nblk, a.ksplit > 1 ? a.ksplit : 1, a.groups);
  // row-major: the blocked order only pays on deep-K shapes (fc2 126 -> 118 us), which the ping-pong kernels take anyway
  GemmArgs b = a;
  b.tile_order = 0;
  hipLaunchKernelGGL(kern, grid, dim3(NWAVES * 64), smem, st, b);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

// Small dense GEMMs -- fewer than 144 tiles of 256 x 128: the LoRA down projections (N = 128 / 256), AV-HuBERT's out-projection, the
// projector layers -- run on the 128 x 64 tile (4 waves, 3 LDS stages = 72 KiB: two workgroups per CU) WITHOUT split-K: 2-8 x the
// workgroups of the big tiles, no fp32 slabs, no reduce launch (round 2 ran them as 256 x 128 tiles with a static split: 20-33 us at
// 40-200 TFLOP/s).  The tile shape never changes a bit (same k order, same epilogue expressions).
static bool small_tile_rule(const GemmArgs& a) {
  if (a.amode != A_DENSE || a.groups != 1 || a.ksplit > 1 || a.swiglu_pairs || a.N < 64 || a.M <= 128) return false;
  return (int64_t)ceil_div(a.M, 256) * ceil_div(a.N, 128) < 144;
}

static int dma_tile_choice(const GemmArgs& a) {
  // 0: not eligible, 1: 256x128, 2: 256x256, 4: the <= 128-row / small-GEMM tiles, 5: grouped conv1d on 128x64 tiles
  // <= 128 rows (the decode-shaped GEMMs, tiny prefills): 128 x 64 tiles stream the weights with 4 x more workgroups per N
  if (a.amode == A_DENSE && a.groups == 1 && a.M <= 128 && (a.M > 32 || a.decode_tiled || a.ksplit > 1) && a.N >= 64) return 4;
  // grouped token-major conv1d with 64 output channels per group and whole k-tiles per tap (AV-HuBERT's positional convolution,
  // wav2vec2.py:825-838: 16 groups of 64 channels, 128 taps): 128 x 64 tiles, one column tile per group (round 5; the classic kernel ran it at 0.12 of peak)
  if (a.amode == A_CONV1D && a.N == 64 && a.groups > 1 && a.ksplit <= 1 && a.M >= 256 && a.Cin % (128 / (a.dtype == DT_F32 ? 4 : 2)) == 0) return 5;
  if (a.ksplit > 1) return 0;      // split-K: decode-shaped launches only (launch_gemm checks)
  if (small_tile_rule(a)) return 4;
  if (a.amode == A_STEM || a.N < 128) return 0;
  const int64_t n128 = (int64_t)ceil_div(a.M, 256) * ceil_div(a.N, 128) * a.groups;
  const int64_t n256 = (int64_t)ceil_div(a.M, 256) * ceil_div(a.N, 256) * a.groups;
  // The 256 x 256 tile (persistent ping-pong kernel where its forms exist) from 64 tiles on.  Alone on the chip it wins only once ~60 % of
  // the CUs get a tile (144; tools/bench_kernels.py) -- the LLM prefill's o_proj / down_proj (80 tiles) and QKV (120) finish sooner on finer
  // tiles that cover more CUs -- but the pipelined job is bound by CU-TIME, not by any one launch's latency: the CUs a coarse launch leaves
  // free run the other calls' kernels and the decode steps, and the 8-wave 256 x 256 tile does the most work per CU-microsecond.  Same box,
  // alternating runs (round 5): threshold 144 with round 4's 160 x 128 tile (4 waves of 80 x 64) for the 80-tile launches 319.2 utt/s, 144 without that tile 323.3,
  // **64: 326.7** -- while the "MFMA side, one call at a time" got 0.8 ms per call WORSE; 48 / 24 / 12 (AV-HuBERT's 52-tile o_proj / fc2 on
  // 52 CUs as well): 332.6 / 332.9 / 333.3 against 335.3 for 64 on a second box -- there the launch is too long a link in its call's chain.
  // The 160 x 128 tile is gone with its last caller.
  // (256 x 192 / 192 x 256 tiles against the ragged last round were built in round 1 and dropped in round 4 for the same reason: alone
  // they won 10 %, with several calls in flight they cost 2 %; round 5's 192-row form of the persistent kernel repeated the lesson.)
  if (a.amode == A_DENSE && a.N >= 256 && n256 >= (launch_alone() ? 144 : 64)) return 2;      // (one call at a time: the latency rule)
  if (n128 >= 144) return 1;
  return 0;
}

template <typename T>
int launch_dma_mode(const GemmArgs& a, hipStream_t st) {
  // 256x256: 16 waves (4 x 4) of 64 x 64, 2 stages; dense only.  256x128: 8 waves (4 x 2) of 64 x 64, 3 stages.
  if (a.force_tile == 2 && a.amode == A_DENSE && a.groups == 1 && a.ksplit <= 1 && a.N >= 256) return launch_dma<T, A_DENSE, 256, 256, 2, 4, 16>(a, st);
  if (a.force_tile == 8 && a.amode == A_DENSE && a.groups == 1 && a.ksplit <= 1) return launch_dma<T, A_DENSE, 256, 128, 3, 4, 8>(a, st);
  if (dma_tile_choice(a) == 2) {
    if (gemm_8pp_eligible(a)) return launch_gemm_8pp(a, st);      // more tiles than CUs: persistent ping-pong kernel (gemm4.hip), bf16
    if (gemm_8p_eligible(a)) return launch_gemm_8p(a, st);        // 8-wave ping-pong schedule (gemm3.hip), bf16
    return launch_dma<T, A_DENSE, 256, 256, 2, 4, 16>(a, st);
  }
  if (dma_tile_choice(a) == 4) {
    // every workgroup re-reads the A rows of its k range: 64-column tiles move 2 B of A per byte of W, 128-column tiles 1 B --
    // worth it as soon as N / 128 still gives every CU several workgroups (lm_head)
    if (a.N >= 256 * 256 && a.ksplit <= 1) return launch_dma<T, A_DENSE, 128, 256, 3, 2, 8>(a, st);   // 8 waves (2 x 4) of 64 x 64: 0.5 B of A per byte of W
    if (a.N >= 256 * 128 && a.ksplit <= 1) return launch_dma<T, A_DENSE, 128, 128, 3, 2, 4>(a, st);   // 4 waves (2 x 2) of 64 x 64
    // 4 waves (4 x 1) of 32 x 64.  Three LDS stages of 24 KiB (72 KiB: two workgroups share a CU): with the whole k range of a split-K
    // slice resident (4 stages, one workgroup per CU) a launch is 8 % faster alone, but beam search keeps several decode chains in flight
    // and what counts there is how many of their workgroups fit beside each other (120.6 -> 125.1 utt/s, round 3).
    return launch_dma<T, A_DENSE, 128, 64, 3, 4, 4>(a, st);
  }
  if (dma_tile_choice(a) == 5) return launch_dma<T, A_CONV1D, 128, 64, 3, 4, 4>(a, st);
  // implicit-GEMM convolutions with >= 256 output channels (trunk layer3 / layer4, resnet.py:61-74): the 16-wave 256 x 256 tile gathers every
  // input row once per 256 columns instead of once per 128 -- the gather (9 taps re-read from L2) is what bounds these launches.  Same k order and
  // epilogue expressions as the 256 x 128 tile: same bits.
  static const bool conv256 = knob("OAVSR_CONV_TILE256") != 0;
  if (conv256 && a.amode == A_CONV2D && a.N >= 256 && a.N % 256 == 0 && a.groups == 1 && a.ksplit <= 1 && sizeof(T) == 2)
    return launch_dma<T, A_CONV2D, 256, 256, 2, 4, 16>(a, st);
  switch (a.amode) {
    case A_DENSE: return launch_dma<T, A_DENSE, 256, 128, 3, 4, 8>(a, st);
    case A_CONV2D: return launch_dma<T, A_CONV2D, 256, 128, 3, 4, 8>(a, st);
    case A_CONV1D: return launch_dma<T, A_CONV1D, 256, 128, 3, 4, 8>(a, st);
  }
  return -1;
}

}  // namespace

bool gemm_dma_eligible(const GemmArgs& a) { return dma_tile_choice(a) != 0; }
bool gemm_small_tile(const GemmArgs& a) { return small_tile_rule(a); }
// fused SwiGLU epilogue (gate / up rows interleaved by 16): LDS-DMA kernels only, whole 32-row blocks, plain epilogue, 16-byte rows
bool gemm_dma_swiglu_ok(const GemmArgs& a) {
  const int vec = a.dtype == DT_F32 ? 4 : 8;
  return dma_tile_choice(a) != 0 && a.ksplit <= 1 && a.groups == 1 && a.N % 32 == 0 && !a.residual && a.act == ACT_NONE &&
         a.ldc % vec == 0 && ((uintptr_t)a.C & 15) == 0;
}
const char* gemm_dma_variant(const GemmArgs& a) {
  const int c = dma_tile_choice(a);
  if (c == 4 || c == 5) return a.ksplit > 1 ? "dma_splitk_128x64" : "dma_128x64";
  if (c == 2 && gemm_8pp_eligible(a)) return "8pp_256x256";
  if (c == 2 && gemm_8p_eligible(a)) return "8p_256x256";
  return a.ksplit > 1 ? "dma_splitk_256x128" : c == 2 ? "dma_256x256" : "dma_256x128";
}

// Fraction of the chip's CUs a launch can occupy (profiling only: duration x share = the launch's CU-time, what a pipelined job pays for
// it while other calls' kernels run on the rest): workgroups / (workgroups that fit a CU x CUs), by the tile the dispatcher will pick.
double gemm_cu_share(const GemmArgs& a) {
  static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  const int c = dma_tile_choice(a);
  const int g = a.groups < 1 ? 1 : a.groups, ks = a.ksplit < 1 ? 1 : a.ksplit;
  double wgs = 0, per_cu = 1;
  if (c == 2) {
    const int64_t n256 = (int64_t)ceil_div(a.M, 256) * ceil_div(a.N, 256);
    wgs = gemm_8pp_eligible(a) ? (double)gemm_8pp_grid(n256) : (double)n256;
  } else if (c == 4 || c == 5) {
    const int bn = (c == 4 && ks <= 1 && a.N >= 256 * 256) ? 256 : (c == 4 && ks <= 1 && a.N >= 256 * 128) ? 128 : 64;
    wgs = (double)ceil_div(a.M, 128) * ceil_div(a.N, bn) * ks * g;
    per_cu = bn == 64 ? 2 : 1;
  } else if (c == 1) {
    wgs = (double)ceil_div(a.M, 256) * ceil_div(a.N, 128) * g;
  } else {
    wgs = (double)ceil_div(a.M, 128) * ceil_div(a.N, 128) * g;      // classic kernel: 128 x 128 tiles, four waves
    per_cu = 2;
  }
  const double share = wgs / (per_cu * ncu);
  return share > 1.0 ? 1.0 : share;
}

int launch_gemm_dma(const GemmArgs& a, hipStream_t st) {
  OAVSR_TRY(a.dtype == DT_F32 ? launch_dma_mode<float>(a, st) : launch_dma_mode<bf16_t>(a, st));
  if (a.ksplit > 1 && !a.sync) {      // split-K without arrival counters: the slabs were stored plainly; the reduce is a launch of its own
    const dim3 grid(ceil_div(a.N, 64), ceil_div(a.M, 32));
    if (a.dtype == DT_F32) hipLaunchKernelGGL(splitk_reduce_kernel<float>, grid, dim3(128), 0, st, a);
    else hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, grid, dim3(128), 0, st, a);
    OAVSR_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

}  // namespace oavsr
