// 256 x 256 bf16 GEMM with an 8-wave "ping-pong" schedule for gfx950:  C[M,N] = epi(A1.W1^T + A2.W2^T), dense operands.
//
// Same math, operand layouts and fused epilogue as gemm2.hip (one accumulation order per output element: k ascending in 32-wide
// MFMA steps, so an element has the same bits whichever tile variant computed it); what differs is the schedule of the k loop:
//
//   * 8 waves; wave (R, Cw) = (wave >> 1, wave & 1) owns FOUR 32 x 64 quadrants of the 256 x 256 tile, at rows mh*128 + R*32 and
//     columns nh*128 + Cw*64 (mh, nh in {0, 1}): its A rows come half from the tile's upper 128 rows (half-tile A0), half from the
//     lower (A1), likewise B0 / B1 for the W rows -- so within a k-tile a wave needs A0 and B0 first, then A1, then B1.  (64-column
//     quadrants: every row segment the epilogue stores is one whole 128-byte line.)
//   * a k-tile (64 bf16 = 128-byte LDS rows) is four PHASES; a phase = [LDS fragment reads + one 16 KiB half-tile of LDS-DMA
//     (2 x global_load_lds_dwordx4 per thread) + counted vmcnt] s_barrier [16 MFMA 16x16x32 = one quadrant x K 64] s_barrier:
//        p0: read A0, B0 (12 ds_read_b128)   MFMA Q00      DMA: A1 of tile t+1
//        p1: read A1     (4)                  MFMA Q10      DMA: B1 of tile t+1
//        p2: read B1     (8)                  MFMA Q11      DMA: B0 of tile t+2
//        p3: --                               MFMA Q01      DMA: A0 of tile t+2      (both A fragment sets stay in registers)
//     Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA section while its partner reads LDS
//     and issues DMA: the matrix pipe never waits for a fragment read.
//   * two 64 KiB k-tile buffers; a half-tile slot is re-staged two phases after its last read (the earliest that is safe with the
//     one-barrier stagger), which puts every DMA five phases (~1.25 k-tiles) ahead of its first read.  `s_waitcnt vmcnt(8)` after
//     each phase's issue keeps the four youngest half-tiles (64 KiB per CU) in flight across the barriers -- never vmcnt(0) in
//     the loop.  Past the last k-tile the issue slots carry dummy loads (into a spare LDS region) so the count stays
//     uniform and the loop needs no tail variant (each lane re-reads its own last chunk: distinct, L2-resident addresses).
//   * LDS image and XOR swizzle (applied to the DMA's per-lane SOURCE address and again on the read) as in gemm2.hip.
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"

namespace oavsr {
namespace {

__device__ __attribute__((aligned(256))) unsigned int g_zero_page3[64];

constexpr int kHalf = 16384;              // one half-tile: 128 rows x 128 B
constexpr int kBuf = 4 * kHalf;           // A0 | A1 | B0 | B1
constexpr int kSpare = 2 * kBuf;          // dummy-DMA landing zone (8 waves x 2 KiB)
constexpr int kSmem8p = 2 * kBuf + 8 * 2048;

__device__ __forceinline__ void dma16_3(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void mma(const Chunk16& a, const Chunk16& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
}

#define OAVSR_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define OAVSR_BARRIER() do { OAVSR_FENCE(); __builtin_amdgcn_s_barrier(); OAVSR_FENCE(); } while (0)

__global__ __launch_bounds__(512, 2) void gemm_8p_kernel(const GemmArgs p) {
  typedef bf16_t T;
  constexpr int BKE = 64, VEC = 8;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R = wave >> 1, Cw = wave & 1;
  const void* zp = g_zero_page3;
  asm volatile("" : "+s"(zp));

  // XCD-aware tile order (as gemm2.hip): an XCD's contiguous range of logical tiles follows the cheaper operand
  const int ntn = (p.N + 255) / 256, ntm = (p.M + 255) / 256;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const bool nmajor = p.M < p.N;
  int mt = nmajor ? bid % ntm : bid / ntn, nt_ = nmajor ? bid / ntm : bid - (bid / ntn) * ntn;
  if (p.tile_order == 1) blocked_tile(bid, ntm, ntn, mt, nt_);
  const int m0 = mt * 256, n0 = nt_ * 256;

  // DMA: half-tile h = 16 segments of 1 KiB (8 rows x 128 B); this wave fills segments `wave` and `wave + 8`.  Lane i lands at
  // physical chunk (i & 7) of row seg*8 + (i >> 3) and therefore fetches logical chunk (i & 7) ^ (i >> 3).  Rows past M / N are
  // clamped to the last valid row: they produce finite values that the epilogue never stores.
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  const T* A1p = reinterpret_cast<const T*>(p.A);
  const T* W1p = reinterpret_cast<const T*>(p.W);
  const char* src[4][2];                         // [A0, A1, B0, B1][segment]; advanced by one k-tile (128 B) per use
  int arow[2][2], wrow[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = h * 128 + (wave + 8 * j) * 8 + lrow;
      arow[h][j] = min(m0 + r, p.M - 1);
      wrow[h][j] = min(n0 + r, p.N - 1);
      src[h][j] = reinterpret_cast<const char*>(A1p + (int64_t)arow[h][j] * p.lda + lchunk * VEC);
      src[2 + h][j] = reinterpret_cast<const char*>(W1p + (int64_t)wrow[h][j] * p.ldw + lchunk * VEC);
    }
  const int nt1 = p.K / BKE, nall = nt1 + p.K2 / BKE;
  auto issue = [&](auto hc, int tt) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;       // compile-time: src[h] stays in registers
    // half-tile h of k-tile tt -> buffer tt & 1; dummy when the k loop has run out of tiles
    if (tt < nall) {
      if (p.K2 && tt == nt1) {                   // switch this half-tile kind to the second K segment (LoRA term)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (h < 2) src[h][j] = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.A2) + (int64_t)arow[h][j] * p.lda2 + lchunk * VEC);
          else src[h][j] = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.W2) + (int64_t)wrow[h - 2][j] * p.ldw2 + lchunk * VEC);
        }
      }
      char* dst = smem + (tt & 1) * kBuf + h * kHalf;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        dma16_3(src[h][j], dst + (wave + 8 * j) * 1024);
        src[h][j] += 128;
      }
    } else {
      // dummy: re-read this lane's last real chunk (an L2 hit at its own address -- one shared dummy address would make every wave of
      // the chip queue on a single L2 line: measured 10+ us per tile) into the spare region
#pragma unroll
      for (int j = 0; j < 2; ++j) dma16_3(nall > 0 ? (const void*)(src[h][j] - 128) : zp, smem + kSpare + wave * 2048 + j * 1024);
    }
  };
  const std::integral_constant<int, 0> kA0; const std::integral_constant<int, 1> kA1;
  const std::integral_constant<int, 2> kB0; const std::integral_constant<int, 3> kB1;

  // fragment read addresses: row r of a half-tile lives at r*128 + ((chunk ^ (r & 7)) << 4); r & 7 == frow & 7 for every row this lane reads
  const int frow = lane & 15, fg = lane >> 4;
  const int sw0 = ((fg ^ (frow & 7)) << 4), sw1 = (((4 + fg) ^ (frow & 7)) << 4);
  const int a_off = (R * 32 + frow) * 128, b_off = 2 * kHalf + (Cw * 64 + frow) * 128;

  f32x4 acc[2][2][2][4];                           // [mh][nh][m-tile][n-tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  Chunk16 af0[2][2], af1[2][2], bf[4][2];

  auto load_a = [&](const char* buf, int mh, Chunk16 (&af)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      af[i][0] = *reinterpret_cast<const Chunk16*>(buf + mh * kHalf + a_off + i * 2048 + sw0);
      af[i][1] = *reinterpret_cast<const Chunk16*>(buf + mh * kHalf + a_off + i * 2048 + sw1);
    }
  };
  auto load_b = [&](const char* buf, int nh) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf[j][0] = *reinterpret_cast<const Chunk16*>(buf + nh * kHalf + b_off + j * 2048 + sw0);
      bf[j][1] = *reinterpret_cast<const Chunk16*>(buf + nh * kHalf + b_off + j * 2048 + sw1);
    }
  };
  auto quad = [&](f32x4 (&c)[2][4], const Chunk16 (&af)[2][2]) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma(af[i][kb], bf[j][kb], c[i][j]);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: the six half-tiles the periodic schedule would have issued in phases -6 .. -1 ----
  issue(kB0, 0); issue(kA0, 0); issue(kA1, 0); issue(kB1, 0); issue(kB0, 1); issue(kA0, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // B0(0), A0(0) have landed (this wave's share)
  OAVSR_BARRIER();
  const bool late = wave >= 4;                            // waves 4-7 run one barrier behind waves 0-3
  if (late) OAVSR_BARRIER();

  for (int t = 0; t < nall; ++t) {
    const char* buf = smem + (t & 1) * kBuf;
    // p0
    load_a(buf, 0, af0);
    load_b(buf, 0);
    issue(kA1, t + 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    OAVSR_BARRIER();
    quad(acc[0][0], af0);
    OAVSR_BARRIER();
    // p1
    load_a(buf, 1, af1);
    issue(kB1, t + 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    OAVSR_BARRIER();
    quad(acc[1][0], af1);
    OAVSR_BARRIER();
    // p2
    load_b(buf, 1);
    issue(kB0, t + 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    OAVSR_BARRIER();
    quad(acc[1][1], af1);
    OAVSR_BARRIER();
    // p3
    issue(kA0, t + 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    OAVSR_BARRIER();
    quad(acc[0][1], af0);
    OAVSR_BARRIER();
  }
  if (!late) OAVSR_BARRIER();                             // balance the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dummy loads of the last phases
  OAVSR_BARRIER();

  // ---- epilogue: every 16 x 64 slab of a quadrant goes through a wave-private fp32 LDS patch so that a lane owns 16 consecutive
  //      columns of one row: residual loads and C stores are 16-byte vectors, four lanes write one whole 128-byte line.
  //      Per-element arithmetic is gemm2.hip's, expression for expression (same bits whichever kernel ran).
  constexpr int PW = 68;                                  // padded patch row (floats)
  float* sw = reinterpret_cast<float*>(smem) + wave * (16 * PW);
  const T* Rp = reinterpret_cast<const T*>(p.residual);
  const int rr = lane >> 2, c0 = (lane & 3) * 16;
  auto epi = [&](f32x4 (&cq)[2][4], int mh, int nh) __attribute__((always_inline)) {
    const int nb = n0 + nh * 128 + Cw * 64 + c0;          // first of this lane's 16 columns
    const bool full = nb + 16 <= p.N;
    const bool vec_ok = full && (p.ldc % VEC) == 0 && ((uintptr_t)p.C & 15) == 0 && (nb % 16) == 0 &&
                        (!Rp || ((p.ldr % VEC) == 0 && ((uintptr_t)Rp & 15) == 0));
    // Everything the quadrant's two slabs read from global memory is requested up front -- the per-column parameters (they depend
    // on the lane's 16 columns only) and the residual row segments -- so the slab loop has no dependent global-load latency in
    // it (with 2 waves per SIMD nothing else would hide it).
    auto load4 = [&](const float* arr, int e, float dflt) -> f32x4 {
      if (!arr) return f32x4{dflt, dflt, dflt, dflt};
      if (full && (((uintptr_t)(arr + nb)) & 15) == 0) return *reinterpret_cast<const f32x4*>(arr + nb + e);
      f32x4 t;
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = nb + e + q < p.N ? arr[nb + e + q] : dflt;
      return t;
    };
    f32x4 pcs[4], pcb[4], pcl[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      pcs[h] = load4(p.col_scale, 4 * h, 1.f); pcb[h] = load4(p.col_bias, 4 * h, 0.f);
      pcl[h] = p.act == ACT_PRELU ? load4(p.prelu, 4 * h, 0.f) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float resq[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + mh * 128 + R * 32 + i * 16 + rr;
#pragma unroll
      for (int e = 0; e < 16; ++e) resq[i][e] = 0.f;
      if (Rp && m < p.M && nb < p.N) {
        if (vec_ok) {
#pragma unroll
          for (int e = 0; e < 16; e += VEC) chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(Rp + (int64_t)m * p.ldr + nb + e), resq[i] + e);
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) if (nb + e < p.N) resq[i][e] = to_f32(Rp[(int64_t)m * p.ldr + nb + e]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sw[(fg * 4 + r) * PW + j * 16 + frow] = cq[i][j][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float y[16];
#pragma unroll
      for (int e = 0; e < 16; e += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(sw + rr * PW + c0 + e);
        y[e] = v[0]; y[e + 1] = v[1]; y[e + 2] = v[2]; y[e + 3] = v[3];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int m = m0 + mh * 128 + R * 32 + i * 16 + rr;
      const float* res = resq[i];
#pragma unroll
      for (int e4 = 0; e4 < 16; e4 += 4) {
        const f32x4 csc = pcs[e4 >> 2], cbi = pcb[e4 >> 2];
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = fmaf(y[e4 + q], csc[q], cbi[q]);
          if (!p.res_after_act) v[q] += res[e4 + q];
        }
        switch (p.act) {
          case ACT_GELU:
            gelu_act4<T>(v);
            break;
          case ACT_RELU:
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            break;
          case ACT_PRELU: {
            const f32x4 csl = pcl[e4 >> 2];
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
      if (p.swiglu_pairs) {
        // gate / up weight rows are interleaved in blocks of 16: lanes with an even (lane & 3) hold a gate block, their right neighbour the
        // matching up block of the same row; both rounded to T first (= GEMM -> swiglu kernel, bit for bit).  All lanes take part in the shuffle.
        float o[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float up = __shfl_down(y[e], 1, 64);
          o[e] = silu(to_f32(from_f32<T>(y[e]))) * to_f32(from_f32<T>(up));
        }
        if (m < p.M && (lane & 1) == 0 && nb + 32 <= p.N) {
          const int64_t oc = (int64_t)m * p.ldc + (nb >> 5) * 16;
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
      } else if (m < p.M && nb < p.N) {
        if (p.out_f32) {
          float* Cf = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + nb;
          if (full && (p.ldc % 4) == 0 && ((uintptr_t)p.C & 15) == 0 && (nb % 4) == 0) {
#pragma unroll
            for (int e = 0; e < 16; e += 4) *reinterpret_cast<f32x4*>(Cf + e) = f32x4{y[e], y[e + 1], y[e + 2], y[e + 3]};
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) if (nb + e < p.N) Cf[e] = y[e];
          }
        } else {
          T* Ct = reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + nb;
          if (vec_ok) {
#pragma unroll
            for (int e = 0; e < 16; e += VEC) *reinterpret_cast<Chunk16*>(Ct + e) = chunk_from_f32<T>(y + e);
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) if (nb + e < p.N) Ct[e] = from_f32<T>(y[e]);
          }
        }
      }
    }
  };
  epi(acc[0][0], 0, 0); epi(acc[0][1], 0, 1); epi(acc[1][0], 1, 0); epi(acc[1][1], 1, 1);
}

}  // namespace

// dense bf16, no groups / split-K; K (and K2) multiples of 64
bool gemm_8p_supported(const GemmArgs& a) {
  return a.dtype == DT_BF16 && a.amode == A_DENSE && a.groups == 1 && a.ksplit <= 1 && a.K % 64 == 0 && a.K2 % 64 == 0 && a.K >= 64 &&
         a.lda % 8 == 0 && a.ldw % 8 == 0 && (((uintptr_t)a.A | (uintptr_t)a.W) & 15) == 0 &&
         (!a.K2 || (a.lda2 % 8 == 0 && a.ldw2 % 8 == 0 && (((uintptr_t)a.A2 | (uintptr_t)a.W2) & 15) == 0));
}
// Measured on MI355X (tools/bench_kernels.py, tools/gemm_kscan.py): per k-tile the ping-pong loop is 10-20 % faster than the 16-wave
// kernel (1.44 vs 1.78 us at 4096 x 4096), but its start-up (six half-tiles of prologue, the half-filled pipeline of the first
// k-tiles) and its 2-waves-per-SIMD epilogue cost ~4-6 us more per tile: it wins from K = 2048 up (Whisper fc2 126 -> 115 us, LLM
// down_proj 135 -> 126 us) and loses at K = 1024 (QKV 97 -> 124 us).  (Tests reach either kernel through oavsr_op_gemm's kernel selector.)
bool gemm_8p_eligible(const GemmArgs& a) { return gemm_8p_supported(a) && a.K + a.K2 >= 2048; }

int launch_gemm_8p(const GemmArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_8p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem8p));
    attr_set = true;
  }
  const int64_t nblk = (int64_t)ceil_div(a.M, 256) * ceil_div(a.N, 256);
  // 8 x 4 blocks of tiles per XCD (kernels.h): +3..5 % on the deep-K shapes this kernel serves
  GemmArgs b = a;
  b.tile_order = 1;
  hipLaunchKernelGGL(gemm_8p_kernel, dim3((unsigned)nblk), dim3(512), kSmem8p, st, b);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace oavsr
