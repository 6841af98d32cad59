// PERSISTENT 256 x 256 bf16 GEMM with the 8-wave ping-pong k loop of gemm3.hip:  C[M,N] = epi(A1.W1^T + A2.W2^T), dense operands.
//
// What a non-persistent tile pays besides its k loop (profiles/r03: 12 us per tile for the 16-wave kernel, 16-18 us for the ping-pong kernel,
// against 23-28 us of k loop at K = 1024): dispatch + address set-up, a cold prologue (six half-tiles through an empty pipeline), the
// transposing epilogue, and -- because every CU of a round reaches its epilogue at the same time -- 128 KiB of C per CU leaving in one
// chip-wide burst, with the workgroup's successor unable to start before the last store is acknowledged.  Here ONE workgroup per CU walks a
// list of output tiles (round j: logical tile j*grid + its position; an XCD's workgroups always hold a contiguous range of logical tiles, in
// the 8 x 4 blocked order of kernels.h) and the k-tile stream simply continues across tiles:
//   * LDS-DMA never drains: the half-tiles issued in the last two k-tiles of tile i are the first 1.5 k-tiles of tile i+1 (issuing A1 / B1
//     of its second k-tile as well before the epilogue's stores -- two more phases of slack for them -- was measured: no gain, removed);
//   * the epilogue's C stores are NOT waited for: vector-memory operations retire in issue order, so the k loop's counted `s_waitcnt
//     vmcnt(N)` of the next tile's first four phases allow S more outstanding operations -- the tile's stores, a compile-time count on full
//     tiles (2 x 16-byte stores per 16-row slab: 16 per wave; fused SwiGLU: 8) -- and the stores drain behind ~2 us of MFMA work.  Edge tiles
//     (ragged M / N: row-guarded stores, unknown count) fall back to the plain counts, i.e. they wait for their stores: slower, never wrong
//     (assuming FEWER outstanding operations than there are is the safe direction);
//   * (round 5) the MFMAs run with the operands swapped -- weights as the A operand, their rows permuted on the way into LDS (free: the
//     DMA's scalar / per-lane row offsets) -- so a lane's accumulators of a 16-row slab ARE 16 consecutive output columns of one row: bias,
//     activation, SwiGLU (gate and up of a column in the same lane), residual and rounding happen in registers and go out as 16-byte row
//     segments, without the LDS transposition of round 4 (measured equal alone; it frees 32 KiB of LDS and half the epilogue's code).  The
//     bias arrives by LDS-DMA during the tile's last k-tile (no ordinary global load anywhere in the loop: the compiler would drain vmcnt);
//   * the epilogue is instantiated per activation (template ACT): with the run-time selections of the generic form (activation switch,
//     residual-before / after-activation, the `+ 0.f` of the other kernels) a plain bias epilogue cost ~6 VALU instructions per element --
//     2.6 us per tile on the two waves of a SIMD; specialised: fc1 (GELU) 114 -> 102 us, gate / up (SwiGLU) 156 -> 145 us, AV-HuBERT fc1
//     40 -> 36 us, same bits (profiles/r05_gemm_epilogue_probe.txt: what the time between two k loops consists of);
//   * the two wave groups re-synchronise for the epilogue (waves 4-7 run one barrier behind in the k loop, see gemm3.hip) and re-stagger
//     after.  Nothing in the register-only epilogue needs that -- letting the stagger run through it (kSyncEpi = false: each group's epilogue
//     under ONE quadrant of the other's MFMAs, the two epilogues one after the other) measured 3-6 % SLOWER on fc1 / gate-up: two waves of a
//     SIMD hide each other's latencies better than 16 MFMAs hide an epilogue.
// Same accumulation order (k ascending, 32 per MFMA) and the same epilogue expressions as gemm2.hip / gemm3.hip: bit-identical to them
// (tests/test_kernels_gpu.py).  No second K segment, no per-column scale / PReLU, bf16 output.  The dispatcher takes it for every dense
// 256 x 256-tile launch it supports: with one tile per workgroup it still beats the 16-wave and the ping-pong kernel (leaner prologue and
// epilogue; RES variant for o_proj / fc2 / down_proj); the LoRA-segment GEMMs stay on gemm2.hip / gemm3.hip.
#include <stdlib.h>
#include <mutex>
#include <set>
#include <type_traits>
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kHalf4 = 16384;                 // one half-tile: 128 rows x 128 B
constexpr int kBuf4 = 4 * kHalf4;             // A0 | A1 | B0 | B1
constexpr int kPatch4 = 2 * kBuf4;            // 8 waves x 4 KiB: where the dummy DMAs of a workgroup's LAST tile land (lower half) and the tile's bias (upper half)
constexpr int kSmem8pp = 2 * kBuf4 + 8 * 4096;   // 160 KiB: the whole LDS of a CU

__device__ __forceinline__ void mma4(const Chunk16& a, const Chunk16& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr bool kSyncEpi = true;
#define PP_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_BARRIER() do { PP_FENCE(); __builtin_amdgcn_s_barrier(); PP_FENCE(); } while (0)

// SWIGLU: fused SwiGLU epilogue (gate / up rows interleaved by 16).  RES: residual rows (o_proj, fc2, down_proj), with the bias and ALL
// residual rows of the tile requested before the first store (ordinary loads: the compiler waits for everything in flight at their first
// use, which costs nothing in the single-round launches this variant serves).  ACT: -1 = generic epilogue (run-time activation /
// residual order), ACT_NONE / ACT_GELU = the specialised forms the launcher picks when they apply (same bits).
// BM: rows of a tile, 256 or 192.  The 192-row form keeps the whole schedule -- eight waves, four phases per k-tile, the same half-tiles of
// W -- and shortens the SECOND half-tile of A to 64 rows (one DMA piece per wave instead of two; a wave's quadrants over it are 16 x 64 instead
// of 32 x 64: 8 MFMAs instead of 16 in phases p1 / p2), so a k-tile is 48 MFMAs per wave instead of 64 and every counted wait sees 7 instead of
// 8 younger operations.  It exists for launches whose 256-row tiles leave CUs idle (Whisper fc2 / o_proj: 188 tiles -> 252); same bits.
template <bool SWIGLU, bool RES, int ACT, int BM = 256>
__global__ __launch_bounds__(512, 2) void gemm_8pp_kernel(const GemmArgs p, const int ntiles) {
  typedef bf16_t T;
  constexpr int BKE = 64, VEC = 8;
  constexpr bool kSpec = ACT >= 0;              // epilogue specialised for a compile-time activation (launcher: ACT_NONE, or ACT_GELU without a residual)
  static_assert(BM == 256 || BM == 192, "tile rows");
  constexpr int M1 = BM == 256 ? 2 : 1;          // 16-row m-tiles of a wave in the quadrants over the second A half-tile
  constexpr int NSLAB = 2 * (2 + M1);            // 16-row slabs a wave stores per tile
  constexpr int S = SWIGLU ? NSLAB : 2 * NSLAB;  // store instructions a wave issues for one FULL tile: 2 per slab (SwiGLU: 1)
  constexpr int kW = BM == 256 ? 8 : 7;          // vector-memory operations of the four half-tile issues younger than the one a phase waits for
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R = wave >> 1, Cw = wave & 1;
  const bool late = wave >= 4;                  // waves 4-7 run one barrier behind waves 0-3 inside the k loop

  // ---- tile list of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (private L2s): in every round the workgroups of
  //      one XCD hold a CONTIGUOUS range of logical tiles (gridDim.x is a multiple of 8; the last, partial round is dealt bijectively).
  const int ntn = (p.N + 255) / 256, ntm = (p.M + BM - 1) / BM;
  const int G = gridDim.x, xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
  const bool nmajor = p.M < p.N;
  auto tile_at = [&](int round, int& m0o, int& n0o) -> bool {
    const int base = round * G, rem = ntiles - base;
    if (rem <= 0) return false;
    int pos;
    if (rem >= G) {
      pos = xcd * (G >> 3) + wi;
    } else {
      const int q = rem >> 3, r = rem & 7;
      if (wi >= q + (xcd < r ? 1 : 0)) return false;
      pos = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + wi;
    }
    const int L = base + pos;
    int mt = nmajor ? L % ntm : L / ntn, nt = nmajor ? L / ntm : L - (L / ntn) * ntn;
    if (p.tile_order == 1) blocked_tile(L, ntm, ntn, mt, nt);
    m0o = mt * BM; n0o = nt * 256;
    return true;
  };
  int m0 = 0, n0 = 0, m0n = 0, n0n = 0;
  if (!tile_at(0, m0, n0)) return;              // workgroup-uniform, before any barrier
  int round = 0;
  bool has_next = tile_at(1, m0n, n0n);

  // ---- DMA sources.  Half-tile kind h in {A0, A1, B0, B1} = 16 segments of 1 KiB (8 rows x 128 B); this wave fills segments `wave` and
  //      `wave + 8`.  Lane i lands at physical chunk (i & 7) of row seg*8 + (i >> 3) and fetches logical chunk (i & 7) ^ (i >> 3).  A and W
  //      are buffer resources that end with their last valid row: a lane's byte offset inside an 8-row segment is ONE constant VGPR per
  //      operand, everything that moves (tile, half, segment, k-tile) is a scalar offset, and rows past M / N arrive as zeros from the
  //      hardware range check (they feed accumulators the epilogue never stores).  Every kind walks the k-tiles of a tile in order and moves
  //      on to the next tile of the list by itself (A0 / B0 in the second-to-last k-tile, A1 / B1 in the last).
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)(((int64_t)p.M - 1) * p.lda * 2 + (int64_t)p.K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (int)(((int64_t)p.N - 1) * p.ldw * 2 + (int64_t)p.K * 2), 0x00020000);
  // Weight rows are PERMUTED on their way into LDS (free: a segment's 8 rows are separate 128-byte pieces either way) so that, with the
  // weights as the MFMA's A operand, a lane's accumulators are consecutive output columns (see the epilogue): LDS row j*16 + fg*4 + r of a
  // wave's 64 columns holds global row fg*16 + j*4 + r -- SwiGLU (gate / up interleaved in blocks of 16): (fg>>1)*32 + (j&1)*16 + (fg&1)*8 +
  // (j>>1)*4 + r.  Segment `wave` = LDS rows wave*8 + lrow: j = wave >> 1, fg = (wave & 1)*2 + (lrow >> 2), r = lrow & 3.
  const int wrow_v = SWIGLU ? (lrow >> 2) * 8 + (lrow & 3) : (lrow >> 2) * 16 + (lrow & 3);
  const int wrow_s = SWIGLU ? (wave & 1) * 32 + ((wave >> 1) & 1) * 16 + (wave >> 2) * 4 : (wave & 1) * 32 + (wave >> 1) * 4;
  const int avo = (lrow * (int)p.lda + lchunk * VEC) * 2, wvo = (wrow_v * (int)p.ldw + lchunk * VEC) * 2;
  const int a64 = 64 * (int)p.lda * 2, w64 = 64 * (int)p.ldw * 2;      // segment `wave + 8` starts 64 rows below segment `wave`
  int so[4];                                    // scalar byte offset of segment `wave` of the k-tile each kind issues next
  auto set_src = [&](auto hc, int m0x, int n0x) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
    if (h < 2) so[h] = (m0x + (h & 1) * 128 + wave * 8) * (int)p.lda * 2;
    else so[h] = (n0x + (h & 1) * 128 + wrow_s) * (int)p.ldw * 2;
  };
  const std::integral_constant<int, 0> kA0; const std::integral_constant<int, 1> kA1;
  const std::integral_constant<int, 2> kB0; const std::integral_constant<int, 3> kB1;
  set_src(kA0, m0, n0); set_src(kA1, m0, n0); set_src(kB0, m0, n0); set_src(kB1, m0, n0);

  const int nall = p.K / BKE;                   // >= 3 (launcher)
  int par = 0;                                  // buffer of local k-tile t of the current tile: (t + par) & 1 (the stream never restarts)
  auto dma2 = [&](auto hc, int soff, char* dst0, char* dst1) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
    if (h < 2) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)dst0, 16, avo, soff, 0, 0);
      if (h == 0 || BM == 256) __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)dst1, 16, avo, soff + a64, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst0, 16, wvo, soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst1, 16, wvo, soff + w64, 0, 0);
    }
  };
  // half-tile h of k-tile tt, counted from the start of the tile being COMPUTED (tt >= nall: k-tile tt - nall of the next tile)
  auto issue = [&](auto hc, int tt) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
    char* dst = smem + ((tt + par) & 1) * kBuf4 + h * kHalf4 + wave * 1024;
    dma2(hc, so[h], dst, dst + 8192);
    so[h] += 128;
  };
  // k-tile lt (0 / 1) of the NEXT tile of the list; without one, dummies that keep the vmcnt arithmetic uniform: each lane re-reads its
  // last real chunk -- an L2 hit at an address of its own -- into this wave's patch, which nobody reads before the drain in front of the
  // workgroup's last epilogue
  auto issue_next = [&](auto hc, int lt) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
    if (has_next) {
      if (lt == 0) set_src(hc, m0n, n0n);
      issue(hc, nall + lt);
    } else {
      char* dst = smem + kPatch4 + wave * 4096;
      dma2(hc, so[h] - 128, dst, dst + 1024);
    }
  };

  // The tile's bias goes HBM -> LDS by DMA as well (an ordinary load in the epilogue would make the compiler drain vmcnt -- the next tile's
  // half-tiles in flight -- at its first use): in the tile's LAST k-tile every wave fetches the 2 x 64 bias values of its own column halves
  // into the spare half of its patch, two 4-byte-per-lane DMAs ahead of that phase's half-tile; they are older than everything the next
  // three waits leave in flight, so they have landed when the epilogue starts (same wave: no barrier needed), and past N they read as
  // zeros by the range check.
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.col_bias), 0, p.col_bias ? p.N * 4 : 0, 0x00020000);
  float* sbias = reinterpret_cast<float*>(smem + kPatch4 + wave * 4096 + 2048);      // [2][64]
  auto issue_bias = [&]() __attribute__((always_inline)) {
    if (!p.col_bias) return;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (__attribute__((address_space(3))) void*)(sbias + nh * 64), 4, lane * 4,
                                               (n0 + nh * 128 + Cw * 64) * 4, 0, 0);
  };

  // fragment read addresses (gemm3.hip): row r of a half-tile lives at r*128 + ((chunk ^ (r & 7)) << 4)
  const int frow = lane & 15, fg = lane >> 4;
  const int sw0 = ((fg ^ (frow & 7)) << 4), sw1 = (((4 + fg) ^ (frow & 7)) << 4);
  const int a_off = (R * 32 + frow) * 128, b_off = 2 * kHalf4 + (Cw * 64 + frow) * 128;
  const int a_off1 = BM == 256 ? a_off : (R * 16 + frow) * 128;      // second A half-tile: 16 rows per wave in the 192-row form

  f32x4 acc[2][2][2][4];                        // [mh][nh][m-tile][n-tile]
  Chunk16 af0[2][2], af1[2][2], bf[4][2];
  auto load_a = [&](const char* buf, auto mhc, Chunk16 (&af)[2][2]) __attribute__((always_inline)) {
    constexpr int mh = decltype(mhc)::value;
#pragma unroll
    for (int i = 0; i < (mh ? M1 : 2); ++i) {
      af[i][0] = *reinterpret_cast<const Chunk16*>(buf + mh * kHalf4 + (mh ? a_off1 : a_off) + i * 2048 + sw0);
      af[i][1] = *reinterpret_cast<const Chunk16*>(buf + mh * kHalf4 + (mh ? a_off1 : a_off) + i * 2048 + sw1);
    }
  };
  auto load_b = [&](const char* buf, int nh) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf[j][0] = *reinterpret_cast<const Chunk16*>(buf + nh * kHalf4 + b_off + j * 2048 + sw0);
      bf[j][1] = *reinterpret_cast<const Chunk16*>(buf + nh * kHalf4 + b_off + j * 2048 + sw1);
    }
  };
  auto quad = [&](f32x4 (&c)[2][4], const Chunk16 (&af)[2][2], auto mic) __attribute__((always_inline)) {
    constexpr int MI = decltype(mic)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma4(bf[j][kb], af[i][kb], c[i][j]);      // weights = A operand: D rows = weight rows
    __builtin_amdgcn_s_setprio(0);
  };
  // counted wait of a tile's first phases: `se` != 0 -> the S stores of the previous (full) tile may still be outstanding
  auto wait_se = [&](auto cc, int se) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value;
    if (se) wait_vm<S + C>(); else wait_vm<C>();
  };
  const std::integral_constant<int, kW> c8;
  const std::integral_constant<int, 0> kMh0; const std::integral_constant<int, 1> kMh1;
  const std::integral_constant<int, 2> kMi2; const std::integral_constant<int, M1> kMi1;

  // One k-tile = four phases (gemm3.hip).  KT 0: steady state.  KT 1: first k-tile of a tile -- the queue still holds the previous tile's
  // stores behind the four half-tiles issued before them, so the counts are larger.  KT 3 / 4: second-to-last / last k-tile -- the
  // half-tiles issued are the next tile's.  What every wait guarantees is unchanged: p0 -> A1(t), p1 -> B1(t), p2 -> B0(t+1), p3 -> A0(t+1) have landed (this
  // wave's share).
  auto ktile = [&](int t, auto ktc, int se) __attribute__((always_inline)) {
    constexpr int KT = decltype(ktc)::value;
    const char* buf = smem + ((t + par) & 1) * kBuf4;
    // p0
    load_a(buf, kMh0, af0);
    load_b(buf, 0);
    if (KT == 4 && !RES) issue_bias();
    if (KT == 4) issue_next(kA1, 0);
    else issue(kA1, t + 1);
    if (KT == 1) wait_se(c8, se); else wait_vm<kW>();
    PP_BARRIER();
    quad(acc[0][0], af0, kMi2);
    PP_BARRIER();
    // p1
    load_a(buf, kMh1, af1);
    if (KT == 4) issue_next(kB1, 0);
    else issue(kB1, t + 1);
    if (KT == 1) wait_se(c8, se); else wait_vm<kW>();
    PP_BARRIER();
    quad(acc[1][0], af1, kMi1);
    PP_BARRIER();
    // p2
    load_b(buf, 1);
    if (KT == 3) issue_next(kB0, 0);
    else if (KT == 4) issue_next(kB0, 1);
    else issue(kB0, t + 2);
    if (KT == 1) wait_se(c8, se); else wait_vm<kW>();
    PP_BARRIER();
    quad(acc[1][1], af1, kMi1);
    PP_BARRIER();
    // p3
    if (KT == 3) issue_next(kA0, 0);
    else if (KT == 4) issue_next(kA0, 1);
    else issue(kA0, t + 2);
    if (KT == 1) wait_se(c8, se); else wait_vm<kW>();
    PP_BARRIER();
    quad(acc[0][1], af0, kMi2);
    PP_BARRIER();
  };
  const std::integral_constant<int, 0> kt_steady; const std::integral_constant<int, 1> kt_first;
  const std::integral_constant<int, 3> kt_penult; const std::integral_constant<int, 4> kt_last;

  // ---- prologue of the workgroup's first tile: the half-tiles the periodic schedule would have issued before its first phase
  issue(kB0, 0); issue(kA0, 0); issue(kA1, 0); issue(kB1, 0); issue(kB0, 1); issue(kA0, 1);
  wait_vm<kW>();                                // B0(0), A0(0) have landed (this wave's share)
  PP_BARRIER();
  if (late) PP_BARRIER();

  // epilogue geometry.  The MFMAs run with the operands SWAPPED (weights = A operand, activations = B operand), so a lane's four
  // accumulator registers of n-block j are four consecutive weight rows of ONE activation row (lane: m = i*16 + frow, LDS weight rows
  // j*16 + fg*4 + r), and the DMA above placed global weight row `perm` into that LDS row such that a lane's 16 registers of a 16-row slab
  // ARE 16 consecutive output columns fg*16 .. fg*16 + 15 of its row (SwiGLU: gate and up of the 8 output columns fg*8 .. fg*8 + 7).  Bias,
  // activation (SwiGLU), residual and rounding all happen in registers and every store is a 16-byte vector, four lanes per 128-byte
  // (64-byte) row segment: no transposition through LDS, nothing shared between waves -- so nobody synchronises for the epilogue: the
  // stagger of the k loop simply continues (waves 0-3 run their epilogue while waves 4-7 issue their last quadrant's MFMAs, waves 4-7 theirs
  // under the next tile's first quadrant of waves 0-3).
  const bool c_vec = (p.ldc % VEC) == 0 && ((uintptr_t)p.C & 15) == 0;
  // float offset of the bias of registers [j][0..3] inside the wave's 64 columns
  auto boff = [&](int j) __attribute__((always_inline)) -> int {
    return SWIGLU ? 32 * (fg >> 1) + 16 * (j & 1) + 8 * (fg & 1) + 4 * (j >> 1) : fg * 16 + j * 4;
  };

  int se = 0;                                   // S if the previous tile of this workgroup issued exactly S stores per wave, else 0
  for (;;) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ktile(0, kt_first, se);
    for (int t = 1; t < nall - 2; ++t) ktile(t, kt_steady, 0);
    ktile(nall - 2, kt_penult, 0);
    ktile(nall - 1, kt_last, 0);

    // ---- tile boundary
    if (!has_next) wait_vm<0>();                // the dummy DMAs of the last phases land in this wave's patch
    if (kSyncEpi && !late) PP_BARRIER();        // waves 0-3 wait for the last quadrant of waves 4-7: both groups run the epilogue together
    PP_FENCE();

    const bool full = m0 + BM <= p.M && n0 + 256 <= p.N && c_vec;
    // RES: the residual row segments of the wave's eight slabs (a lane's 16 columns of row frow), raw bf16, requested up front
    Chunk16 rsd[RES ? 2 : 1][RES ? 2 : 1][RES ? 2 : 1][2];
    if constexpr (RES) {
      const T* Rp = reinterpret_cast<const T*>(p.residual);
      const bool r_vec = (p.ldr % VEC) == 0 && ((uintptr_t)Rp & 15) == 0;
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int i = 0; i < (mh ? M1 : 2); ++i) {
            const int m = m0 + mh * 128 + ((mh && BM == 192) ? R * 16 : R * 32) + i * 16 + frow;
            const int nb = n0 + nh * 128 + Cw * 64 + fg * 16;
            if (m < p.M && nb + 16 <= p.N && r_vec) {
              rsd[mh][nh][i][0] = *reinterpret_cast<const Chunk16*>(Rp + (int64_t)m * p.ldr + nb);
              rsd[mh][nh][i][1] = *reinterpret_cast<const Chunk16*>(Rp + (int64_t)m * p.ldr + nb + 8);
            } else {
              float t[16];
#pragma unroll
              for (int e = 0; e < 16; ++e) t[e] = (m < p.M && nb + e < p.N) ? to_f32(Rp[(int64_t)m * p.ldr + nb + e]) : 0.f;
              rsd[mh][nh][i][0] = chunk_from_f32<T>(t);
              rsd[mh][nh][i][1] = chunk_from_f32<T>(t + 8);
            }
          }
    }
    auto epi = [&](f32x4 (&cq)[2][4], int mh, int nh) __attribute__((always_inline)) {
      // bias of the lane's registers [j][r], from the LDS copy the last k-tile's DMA made (RES: from global memory)
      float cb[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (RES) {
          const int n = n0 + nh * 128 + Cw * 64 + boff(j);
#pragma unroll
          for (int r = 0; r < 4; ++r) cb[j][r] = (p.col_bias && n + r < p.N) ? p.col_bias[n + r] : 0.f;
        } else if (p.col_bias) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(sbias + nh * 64 + boff(j));
          cb[j][0] = t4[0]; cb[j][1] = t4[1]; cb[j][2] = t4[2]; cb[j][3] = t4[3];
        } else {
          cb[j][0] = cb[j][1] = cb[j][2] = cb[j][3] = 0.f;
        }
        // specialised forms, no residual: gemm2.hip / gemm3.hip compute (acc + bias) + 0.f (the `+ 0.f` turns -0 into +0); acc + (bias + 0.f)
        // has the same bits for every input -- the sum can only be -0 when BOTH addends are, and bias + 0.f never is -- at one add per
        // element instead of two
        if constexpr (kSpec && !RES) {
#pragma unroll
          for (int r = 0; r < 4; ++r) cb[j][r] += 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (mh && i >= M1) continue;             // 192-row form: one m-tile over the second A half-tile
        const int m = m0 + mh * 128 + ((mh && BM == 192) ? R * 16 : R * 32) + i * 16 + frow;
        float res[16];
        if constexpr (RES) {
          chunk_to_f32<T>(rsd[mh][nh][i][0], res);
          chunk_to_f32<T>(rsd[mh][nh][i][1], res + 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (kSpec) {
            // ACT is a compile-time constant (no activation, or GELU without a residual): the generic form below with its run-time
            // selections folded away -- ~6 VALU instructions per element down to 1 (+ the activation)
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = cq[i][j][r] + cb[j][r];
              if constexpr (RES) v[r] += res[j * 4 + r];
            }
            if constexpr (ACT == ACT_GELU) gelu_act4<T>(v);
#pragma unroll
            for (int r = 0; r < 4; ++r) cq[i][j][r] = v[r];
          } else {
            // per-element arithmetic: gemm2.hip / gemm3.hip expression for expression with col_scale = 1 (without a residual the `+ 0.f`
            // is theirs too)
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = fmaf(cq[i][j][r], 1.f, cb[j][r]);
              if (!p.res_after_act) v[r] += RES ? res[j * 4 + r] : 0.f;
            }
            switch (p.act) {                      // one uniform branch per four elements
              case ACT_GELU:
                gelu_act4<T>(v);
                break;
              case ACT_RELU:
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                break;
              case ACT_SILU:
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
                break;
              default: break;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) cq[i][j][r] = p.res_after_act ? v[r] + (RES ? res[j * 4 + r] : 0.f) : v[r];
          }
        }
        if (SWIGLU) {
          // gate / up weight rows are interleaved in blocks of 16; the DMA permutation put gate (j even) and up (j odd) of the lane's 8 output
          // columns into the same lane.  Both are rounded to T first (what the unfused GEMM -> swiglu_kernel sequence reads back).
          float o[8];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[jj * 4 + r] = silu(to_f32(from_f32<T>(cq[i][2 * jj][r]))) * to_f32(from_f32<T>(cq[i][2 * jj + 1][r]));
          const Chunk16 out = chunk_from_f32<T>(o);
          const int nb = ((n0 + nh * 128 + Cw * 64) >> 1) + fg * 8;     // output column (N / 2 columns in all)
          T* Ct = reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + nb;
          if (full) {
            *reinterpret_cast<Chunk16*>(Ct) = out;
          } else if (m < p.M && nb + 8 <= (p.N >> 1)) {
            if (c_vec) *reinterpret_cast<Chunk16*>(Ct) = out;
            else {
              const bf16x8 ob = __builtin_bit_cast(bf16x8, out.v);
#pragma unroll
              for (int e = 0; e < 8; ++e) Ct[e] = ob[e];
            }
          }
        } else {
          float y[16];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[j * 4 + r] = cq[i][j][r];
          const Chunk16 out0 = chunk_from_f32<T>(y), out1 = chunk_from_f32<T>(y + 8);
          const int nb = n0 + nh * 128 + Cw * 64 + fg * 16;          // first of this lane's 16 columns
          T* Ct = reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + nb;
          if (full) {
            *reinterpret_cast<Chunk16*>(Ct) = out0;
            *reinterpret_cast<Chunk16*>(Ct + 8) = out1;
          } else if (m < p.M && nb < p.N) {
            if (c_vec && nb + 16 <= p.N) {
              *reinterpret_cast<Chunk16*>(Ct) = out0;
              *reinterpret_cast<Chunk16*>(Ct + 8) = out1;
            } else {
#pragma unroll
              for (int e = 0; e < 16; ++e) if (nb + e < p.N) Ct[e] = from_f32<T>(y[e]);
            }
          }
        }
      }
    };
    epi(acc[0][0], 0, 0); epi(acc[1][0], 1, 0); epi(acc[1][1], 1, 1); epi(acc[0][1], 0, 1);
    PP_FENCE();

    if (!has_next) break;
    se = full ? S : 0;
    par = (par + nall) & 1;
    m0 = m0n; n0 = n0n;
    ++round;
    has_next = tile_at(round + 1, m0n, n0n);
    if (kSyncEpi && late) PP_BARRIER();         // re-stagger: waves 4-7 fall one barrier behind again
  }
}

int ncu8() {
  static const int n = [] {
    int d = 0, c = 256;
    (void)hipGetDevice(&d);
    (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d);
    return c & ~7;
  }();
  return n;
}

}  // namespace

// what the kernel can compute at all: the ping-pong kernel's operands (one K segment, at least four k-tiles), bf16 output on 16-byte rows,
// bias + GELU / ReLU / SiLU (or the fused SwiGLU)
bool gemm_8pp_supported(const GemmArgs& a) {
  const int64_t abytes = ((int64_t)a.M - 1) * a.lda * 2 + (int64_t)a.K * 2, wbytes = ((int64_t)a.N - 1) * a.ldw * 2 + (int64_t)a.K * 2;
  if (abytes >= (1ll << 31) || wbytes >= (1ll << 31)) return false;      // 32-bit buffer offsets
  return gemm_8p_supported(a) && a.K2 == 0 && a.K >= 192 && !(a.residual && a.swiglu_pairs) && !a.col_scale && a.act != ACT_PRELU && !a.out_f32 && a.ldc % 8 == 0 &&
         ((uintptr_t)a.C & 15) == 0 && (!a.swiglu_pairs || a.N % 64 == 0);
}
// workgroups for `ntiles` tiles: as few rounds as the chip allows, then the smallest multiple of 8 that still needs that many rounds
// (Whisper QKV: 564 tiles -> 3 rounds -> 192 workgroups; the other 64 CUs stay free for the kernels of other streams)
int gemm_8pp_grid(int64_t ntiles) {
  const int cap = ncu8() < 8 ? 8 : ncu8();
  const int64_t rounds = (ntiles + cap - 1) / cap;
  int64_t g = (ntiles + rounds - 1) / rounds;
  g = (g + 7) / 8 * 8;
  return (int)(g > cap ? cap : g);
}
// the dispatcher's rule: every shape the 256 x 256 tiles serve (>= 144 tiles, launch_dma_mode) that the kernel supports.  With more tiles
// than CUs persistence pays (Whisper QKV 104 -> 85 us); with one tile per workgroup this kernel still beats the 16-wave and the ping-pong
// kernel through its leaner prologue and epilogue (scalar DMA addressing, C-layout bias / activation: Whisper fc2 108 -> 89 us, o_proj 34 ->
// 30 us, 4096^3 1.13 -> 1.27 PFLOP/s; profiles/r04_vendor_yardstick.txt)
bool gemm_8pp_eligible(const GemmArgs& a) { return gemm_8pp_supported(a); }

// Tile rows.  The 192-row form puts an under-filled launch on more CUs (Whisper fc2 / o_proj: 188 -> 252 tiles, QKV: 192 -> 256 workgroups
// for three rounds) and finishes it sooner ALONE -- QKV 83 -> 74 us, fc2 91 -> 84, o_proj 31 -> 29 (profiles/r05_gemm_8pp_rows192.txt) -- but a
// 192-row tile takes 0.89-0.93 of a 256-row tile's time for 0.75 of its work: ~20 % more CU-time per launch.  In the pipelined job the CUs a
// 256-row launch leaves free are not idle (other calls' kernels, the decode steps), and the headline went 321-322 -> 317-318 utt/s with the
// finer tiles while the family's own time fell 10.9 -> 10.6 ms per call.  So the launcher keeps 256 rows while calls share the chip and takes
// the 192-row form where it finishes sooner when ONE call runs at a time (launch_alone(), oavsr_set_launch_objective); OAVSR_GEMM_TILE_ROWS
// forces either.  Never a bit: both forms accumulate every element in the
// same order (tests compare them).
int gemm_8pp_rows(const GemmArgs& a) {
  static const int forced = knob("OAVSR_GEMM_TILE_ROWS");
  const bool has192 = !a.swiglu_pairs && (a.act == ACT_NONE || (a.act == ACT_GELU && !a.res_after_act && !a.residual));
  if (!has192) return 256;
  if (forced == 192) return 192;
  if (forced == 256 || !launch_alone()) return 256;
  // one call at a time (oavsr_set_launch_objective): the form that finishes sooner -- rounds x tile time, a 192-row tile at 0.9
  const int cap = ncu8() < 8 ? 8 : ncu8();
  const int64_t ntn = ceil_div(a.N, 256), t256 = (int64_t)ceil_div(a.M, 256) * ntn, t192 = (int64_t)ceil_div(a.M, 192) * ntn;
  const int64_t r256 = (t256 + cap - 1) / cap, r192 = (t192 + cap - 1) / cap;
  return r192 * 90 < r256 * 100 ? 192 : 256;
}

int launch_gemm_8pp(const GemmArgs& a, hipStream_t st, int grid, int rows) {
  OAVSR_REQUIRE(gemm_8pp_supported(a), "gemm_8pp: shape / epilogue not supported by the persistent ping-pong kernel");
  if (rows == 0) rows = gemm_8pp_rows(a);
  OAVSR_REQUIRE(rows == 256 || rows == 192, "gemm_8pp: tile rows must be 256 or 192 (got %d)", rows);
  const bool none = a.act == ACT_NONE;
  OAVSR_REQUIRE(rows == 256 || (!a.swiglu_pairs && (none || (a.act == ACT_GELU && !a.res_after_act && !a.residual))),
                "gemm_8pp: the 192-row form serves plain / residual epilogues without activation and bias + GELU");
  const int64_t ntiles = (int64_t)ceil_div(a.M, rows) * ceil_div(a.N, 256);
  if (grid <= 0) grid = gemm_8pp_grid(ntiles);
  OAVSR_REQUIRE(grid % 8 == 0 && grid >= 8, "gemm_8pp: the grid must be a multiple of 8 (got %d)", grid);
  if ((int64_t)grid > (ntiles + 7) / 8 * 8) grid = (int)((ntiles + 7) / 8 * 8);     // (a ragged single round is dealt over the 8 XCD groups by tile_at)
  GemmArgs b = a;
  b.tile_order = 1;
  auto go = [&](auto kern) -> int {
    // every instantiation has the same function-pointer type (one lambda instance): the attribute is remembered per kernel POINTER
    static std::mutex mu;
    static std::set<const void*> attr_set;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!attr_set.count(reinterpret_cast<const void*>(kern))) {
        OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem8pp));
        attr_set.insert(reinterpret_cast<const void*>(kern));
      }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), kSmem8pp, st, b, (int)ntiles);
    OAVSR_CHECK_HIP(hipGetLastError());
    return 0;
  };
  // epilogue specialisations (same bits as the generic form, see the kernel): no activation -- whatever res_after_act says, there is
  // nothing to order --, and GELU without a residual (fc1)
  if (rows == 192) {
    if (a.residual) return go(gemm_8pp_kernel<false, true, ACT_NONE, 192>);
    return none ? go(gemm_8pp_kernel<false, false, ACT_NONE, 192>) : go(gemm_8pp_kernel<false, false, ACT_GELU, 192>);
  }
  if (a.residual) return none ? go(gemm_8pp_kernel<false, true, ACT_NONE>) : go(gemm_8pp_kernel<false, true, -1>);
  if (a.swiglu_pairs) return none ? go(gemm_8pp_kernel<true, false, ACT_NONE>) : go(gemm_8pp_kernel<true, false, -1>);
  if (none) return go(gemm_8pp_kernel<false, false, ACT_NONE>);
  if (a.act == ACT_GELU && !a.res_after_act) return go(gemm_8pp_kernel<false, false, ACT_GELU>);
  return go(gemm_8pp_kernel<false, false, -1>);
}

}  // namespace oavsr
