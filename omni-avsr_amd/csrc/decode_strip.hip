// Decode-shaped GEMMs of FEW rows (<= kStripRows = 16: the reference's own protocol -- one utterance, one row per step, 15 with beam
// search, Llama_LoRA.py:246-259,314,372,587-655 -- and every call-private decode loop of <= 16 rows):  C[M,N] = epi(A1.W1^T + A2.W2^T).
//
// The 128-row LDS-DMA tiles (gemm2.hip) serve these launches with the arithmetic of a 128-row batch: 2 / 3 of a k-tile's LDS traffic is
// the zero page, the K split lives in separate workgroups whose fp32 slabs go through memory, an arrival counter and a second pass -- a
// launch costs 12-14 us to move 1-14 MB (profiles/r05_decode_mix_timing.txt), 8-9 us of it fixed.  Here the same arithmetic is laid out for the
// weights, which are all that moves:
//   * a WAVE owns one item = one 16-column strip of W x one k-slice (the slices are the tiled kernels': slice s = k-tiles [s per, (s + 1) per),
//     per = ceil(k-tiles / ksplit), the LoRA segment's k-tiles behind the main segment's) and streams its weights through a wave-PRIVATE FIFO of D
//     k-tiles in LDS (LDS-DMA, non-temporal: every byte is read once per step; lane l reads back the 16 bytes lane l fetched); its <= 16 activation
//     rows come from L2 the same way.  No barrier and no waiting on another wave inside the k loop, counted waits only; the MFMAs are the tiled
//     kernels' (Mma2: activations as the A operand, k-tile by k-tile, chunk 0 then 1), so an item's 16 x 16 partial sums have the bits of the
//     corresponding slab entries
//   * the items of a strip sit in ONE workgroup: their partial sums meet in LDS and are added in slab order (0.f + s0 + s1 + ...: what
//     splitk_reduce_tile does with the slabs), then the shared epilogue (epilogue.h: segment_finish) -- no slabs in memory, no arrival
//     counter, no reduce launch.  ksplit stays what it is on the tiled path: a function of (N, K) that defines the summation order; here it only
//     says how many waves share a strip
//   * the deferred RMSNorm's per-row sums of squares are kept per 16-column strip (GemmArgs::rowsq_strips); a consumer forms the tiled
//     producer's 64-column butterfly (s0 + s1) + (s2 + s3) from them and adds the tiles in order (row_sumsq_form = 1): same bits as the tiled path
// => a row decodes to the same bits at 1 row and inside a 128-row step (tests/test_kernels_gpu.py::test_decode_shaped_gemm_rows_*).
#include "kernels.h"
#include "epilogue.h"

namespace oavsr {
namespace {

constexpr int kStripRowBytes = 128;
constexpr int kStripMaxTiles = 128;      // row_sumsq_tiles the kernel can take (hidden <= 8192)
// one k-tile of a wave's ring: W chunk 0 | W chunk 1 (1 KiB each: 16 weight rows x 64 bytes) | A: MA activation rows x 128 bytes (MA = 8 or 16)
// (a 2-row form for M <= 2 -- 2.25 KiB per k-tile, rings a third deeper -- was measured: no change; a CU's streaming rate does not follow its ring depth)
constexpr int slot_bytes(int MA) { return 2048 + MA * 128; }

template <typename T> __device__ __forceinline__ Chunk16 ld16(const T* p) { return *reinterpret_cast<const Chunk16*>(p); }

// HBM -> LDS without a register destination: lane l's 16 bytes land at lds_wave_base + 16 l.  Nothing the compiler allocates is written
// asynchronously (a first version streamed into registers from inline asm; the compiler split the ring's live ranges between the loop and
// its tail and COPIED registers whose loads were still in flight).  aux 2 = nt: weights are read once per step.
template <int AUX>
__device__ __forceinline__ void strip_dma16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}
// hand-off of 1 KiB between the two workgroups of a PAIR (below): the form of gemm2.hip's in-launch reduce -- write-through (sc1) 16-byte stores, drained,
// an agent-scope add behind the workgroup's barrier; the reader polls with sc1 loads, acquires, and reads with sc1 loads behind a barrier
__device__ __forceinline__ __amdgpu_buffer_rsrc_t strip_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void strip_st16_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, byte_off, 0, 16);      // aux bit 4 = sc1 on gfx950
}
__device__ __forceinline__ f32x4 strip_ld16_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16));
}
template <int N> __device__ __forceinline__ void strip_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// D: k-tiles in flight per wave (ring slots); NW: waves per workgroup; MA: activation rows staged per k-tile (8: M <= 8, one DMA instruction; 16).
// <= 16 rows: one MFMA row block.
//
// CHAIN (decode_chain_kernel below): the launch is one PHASE of several GEMMs that feed each other, its workgroups behind the producer's in
// block order.  A wave requests the weights of its first D k-tiles, THEN waits for the producer phase (wait_ctr == wait_n workgroups done,
// agent-scope acquire), and only then touches what the producer wrote: activations, residual, sums of squares.  At the end every wave
// releases its stores and the workgroup counts itself done (done_ctr).  Same items, same MFMA order, same finish: same bits.
// (a macro, not a lambda: captured by reference, tq[][] stayed in scratch memory)
#define OAVSR_STRIP_FIN_LOADS() \
  do { \
    if (flive && p.residual) { \
      const T* Rr = reinterpret_cast<const T*>(p.residual) + (int64_t)frw * p.ldr + fnc; \
      _Pragma("unroll") for (int e = 0; e < 16; e += VEC) rraw[e / VEC] = ld16(Rr + e); \
    } \
    if (nT > 0) { \
      _Pragma("unroll") for (int j = 0; j < JQ; ++j) { \
        const int idx = tid + j * kThreads; \
        if (idx < ntq) { \
          const int t = idx / p.M, r = idx - t * p.M; \
          if (p.row_sumsq_form) { \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) tq[j][q] = p.row_sumsq[(int64_t)(4 * t + q) * p.M + r]; \
          } else { \
            tq[j][0] = p.row_sumsq[(int64_t)t * p.M + r]; \
          } \
        } \
      } \
    } \
  } while (0)
//
// PAIR: a launch of few strips with long k-slices (down_proj of a 2048-wide model: 128 strips x 8 slices of 1024) fills half the chip, and a CU
// streams ~20 GB/s whatever the depth of its rings (measured: deeper rings change nothing, the per-CU rate of 128- and 256-workgroup launches
// is the same) -- so TWO workgroups share a strip: block 2 i takes the first half of its k-slices, sums them in slab order (0.f + s0 + ...) and
// hands the 16 x 16 fp32 sums over through memory; block 2 i + 1 takes the second half, CONTINUES that sum in slab order (+ s_h + ...) and
// finishes.  The same chain of additions as one workgroup: same bits.  Block 2 i never waits and is dispatched first; the wait is bounded.
template <typename T, int D, int NW, int MA, bool CHAIN, bool PAIR = false>
__device__ __forceinline__ void strip_body(const GemmArgs& p, const int spw, const int wg, char* smem, const int* wait_ctr, const int wait_n,
                                           int* done_ctr, int* err, int* clean_ctr = nullptr, const int clean_n = 0, const int nwg = 0) {
  constexpr int kSlotBytes = slot_bytes(MA);
  constexpr int VEC = TypeInfo<T>::kVec;
  constexpr int BKE = kStripRowBytes / (int)sizeof(T);
  constexpr int kThreads = NW * 64;
  constexpr int kRows = 16;
  constexpr int JQ = 4;                                  // tile sums of the row factor a thread prepares from registers
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;
  const int ks = p.ksplit < 1 ? 1 : p.ksplit;
  const int nt1 = p.K / BKE, nall = nt1 + p.K2 / BKE;
  const int per = (nall + ks - 1) / ks;
  const int nstrips = (p.N + 15) >> 4;
  const int half = PAIR ? (wg & 1) : 0;                   // PAIR: spw == 1, this workgroup's k-slices are [half ksl, (half + 1) ksl)
  const int ksl = PAIR ? ks / 2 : ks;
  const int strip0 = PAIR ? (wg >> 1) : wg * spw;
  const int items = spw * ksl;
  const int nT = p.row_sumsq ? p.row_sumsq_tiles : 0;
  char* ring = smem + wave * (D * kSlotBytes);            // this wave's FIFO
  float* slab = reinterpret_cast<float*>(smem + NW * D * kSlotBytes);      // [items][16][16] partial sums
  float* tsum = slab + (size_t)items * kRows * 16;        // [rows][nT] tile sums of squares (row factor)
  const T* A1 = reinterpret_cast<const T*>(p.A);
  const T* A2 = reinterpret_cast<const T*>(p.A2);
  const T* W1 = reinterpret_cast<const T*>(p.W);
  const T* W2 = reinterpret_cast<const T*>(p.W2);

  // ---- what the finish needs from memory goes out FIRST (the oldest loads: landed long before anything waits for them)
  // finish thread -> (strip sl of the workgroup, row): 16 columns of one row; fused SwiGLU: (pair of strips, row)
  const bool swi = p.swiglu_pairs != 0;
  const int fsl = tid / kRows, frw = tid % kRows;
  const int fstrip = strip0 + (swi ? 2 * fsl : fsl);
  const bool flive = fsl < (swi ? spw / 2 : spw) && frw < p.M && fstrip < nstrips;
  const int fnc = fstrip * 16;
  Chunk16 rraw[16 / VEC];
  float tq[JQ][4];
  const int ntq = p.M * nT;                              // tile sums this workgroup needs: [t][row], row fastest (coalesced)
  if constexpr (!CHAIN) OAVSR_STRIP_FIN_LOADS();
  // CHAIN: the producer phase is complete when its workgroups have all counted themselves done.  Bounded: a phase that never completes (a
  // bug, never a legal schedule: producers sit in lower blocks, which the hardware dispatches first and which wait for nothing younger)
  // flags `err` instead of hanging the device.
  bool waited = !CHAIN;
#define OAVSR_STRIP_CHAIN_WAIT() \
  do { \
    if (wait_ctr) {      /* ONE wave polls and acquires for the workgroup (the acquire invalidates the CU's L1); the others join at the barrier */ \
      if (wave == 0) { \
        int spins = 0; \
        while (__hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_n) { \
          __builtin_amdgcn_s_sleep(4); \
          if (++spins > (1 << 20)) { if (lane == 0) atomicAdd(err, 1); break; } \
        } \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); \
      } \
      __syncthreads(); \
    } \
    waited = true; \
    OAVSR_STRIP_FIN_LOADS(); \
  } while (0)

  // ---- the items of this wave.  Rows past M read row M - 1: an output row depends on its own activation row only and rows >= M are never
  //      stored, so there is no zero fill and no branch between the loads of the ring
  // Activations: lane l of an A instruction fetches chunk (l & 7) ^ (l >> 3) of row 8 j + (l >> 3) (j: the instruction; 8 rows x 128 bytes each), so
  // the k-tile's rows sit in LDS as [row][chunk ^ (row & 7)] -- the MFMA fragment reads (row = lane & 15, chunk = 4 kb + lane / 16) are conflict-free
  const int arow_l = lane >> 3, achunk_l = (lane & 7) ^ (lane >> 3);
  constexpr int AJ = MA / 8;
  const T* a1row[AJ];
  const T* a2row[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int mr = min(j * 8 + arow_l, p.M - 1);
    a1row[j] = A1 + (int64_t)mr * p.lda + achunk_l * VEC;
    a2row[j] = A2 ? A2 + (int64_t)mr * p.lda2 + achunk_l * VEC : nullptr;
  }
  const int fr_a = frow & (MA - 1);                        // (MA = 8: lanes of rows 8..15 re-read rows 0..7 -- rows >= M are never stored)
  const int a_off0 = (fr_a * 8 + (fg ^ (fr_a & 7))) * 16, a_off1 = (fr_a * 8 + ((4 + fg) ^ (fr_a & 7))) * 16;
  for (int it = wave; it < items; it += NW) {
    const int sl = it / ksl, s = it - sl * ksl + half * ksl;
    const int strip = strip0 + sl;
    if (strip >= nstrips) continue;
    const int wr = min(strip * 16 + frow, p.N - 1);       // ragged last strip (lm_head's vocabulary): the row is read, its column never stored
    const T* w1row = W1 + (int64_t)wr * p.ldw + fg * VEC;
    const T* w2row = W2 ? W2 + (int64_t)wr * p.ldw2 + fg * VEC : nullptr;
    const int tb = s * per;
    const int n = tb + per <= nall ? per : (nall - tb > 0 ? nall - tb : 0);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};

    auto loadw = [&](int t, int slot) {                   // the k-tile's two 16-byte chunks of W per lane
      const bool s1 = t < nt1;                            // wave-uniform
      const int k = (s1 ? t : t - nt1) * BKE;
      const T* wp = (s1 ? w1row : w2row) + k;
      char* dst = ring + slot * kSlotBytes;
      strip_dma16<2>(wp, dst);
      strip_dma16<2>(wp + 4 * VEC, dst + 1024);
    };
    auto loada = [&](int t, int slot) {                   // MA rows of A
      const bool s1 = t < nt1;
      const int k = (s1 ? t : t - nt1) * BKE;
      char* dst = ring + slot * kSlotBytes;
#pragma unroll
      for (int j = 0; j < AJ; ++j) strip_dma16<0>((s1 ? a1row[j] : a2row[j]) + k, dst + 2048 + j * 1024);
    };
    auto loadk = [&](int t, int slot) { loadw(t, slot); loada(t, slot); };      // L LDS-DMA instructions
    auto mma = [&](int slot) {                            // W: each lane reads back exactly what it fetched; A: the swizzled row-major tile
      const char* src = ring + slot * kSlotBytes;
      const Chunk16 w0 = *reinterpret_cast<const Chunk16*>(src + lane * 16), w1 = *reinterpret_cast<const Chunk16*>(src + 1024 + lane * 16);
      const Chunk16 a0 = *reinterpret_cast<const Chunk16*>(src + 2048 + a_off0), a1 = *reinterpret_cast<const Chunk16*>(src + 2048 + a_off1);
      Mma2<T>::run(a0, w0, acc);
      Mma2<T>::run(a1, w1, acc);
      asm volatile("" ::: "memory");                      // the slot is read before anything below refills it
    };

    // L DMA instructions per k-tile; a slot's are the oldest outstanding when it is consumed.  Steady state: D L outstanding before a slot is
    // consumed -> vmcnt((D - 1) L); the drain uses counts that are exact or smaller (a smaller count only waits longer).
    // (the previous item's last LDS reads are done: its MFMAs took their operands before its slab entry was written)
    constexpr int L = 2 + AJ;
    if (CHAIN && !waited) {      // the wave's first item: weights, the producer, then what it wrote; everything has landed when the loop starts (its counts hold trivially)
#pragma unroll
      for (int i = 0; i < D; ++i)
        if (i < n) loadw(tb + i, i);
      OAVSR_STRIP_CHAIN_WAIT();
#pragma unroll
      for (int i = 0; i < D; ++i)
        if (i < n) loada(tb + i, i);
      strip_wait<0>();
    } else {
#pragma unroll
      for (int i = 0; i < D; ++i)
        if (i < n) loadk(tb + i, i);
    }
    int done = 0;
    for (; done + 2 * D <= n; done += D) {                // steady state: every slot is consumed and refilled, no branch in the body
      static_for<0, D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        strip_wait<(D - 1) * L>();
        mma(i);
        loadk(tb + done + D + i, i);
      });
    }
    const int rem = n - done;                             // < 2 D k-tiles left, the first min(rem, D) of them in the ring
    if (rem >= D) {
      static_for<0, D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        strip_wait<(D - 1 - i) * L>();                    // the slots behind i are outstanding (+ the refills: waited for early)
        mma(i);
        if (i + D < rem) loadk(tb + done + D + i, i);
      });
      strip_wait<0>();
      static_for<0, D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i + D < rem) mma(i);
      });
    } else {
      strip_wait<0>();
      static_for<0, D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i < rem) mma(i);
      });
    }

    float* sp = slab + (size_t)it * kRows * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) sp[(fg * 4 + r) * 16 + frow] = acc[r];
  }

  if (CHAIN && !waited) OAVSR_STRIP_CHAIN_WAIT();      // (a wave without an item)
  // ---- row factor: the tile sums (the tiled producer's butterfly over a tile's four strips, or the producer's own tile sums) into LDS
  if (nT > 0) {
#pragma unroll
    for (int j = 0; j < JQ; ++j) {
      const int idx = tid + j * kThreads;
      if (idx < ntq) {
        const int t = idx / p.M, r = idx - t * p.M;
        tsum[r * nT + t] = p.row_sumsq_form ? (tq[j][0] + tq[j][1]) + (tq[j][2] + tq[j][3]) : tq[j][0];
      }
    }
    for (int idx = tid + JQ * kThreads; idx < ntq; idx += kThreads) {      // (more tile sums than the register set holds: very wide models)
      const int t = idx / p.M, r = idx - t * p.M;
      float v;
      if (p.row_sumsq_form) {
        const float q0 = p.row_sumsq[(int64_t)(4 * t) * p.M + r], q1 = p.row_sumsq[(int64_t)(4 * t + 1) * p.M + r];
        const float q2 = p.row_sumsq[(int64_t)(4 * t + 2) * p.M + r], q3 = p.row_sumsq[(int64_t)(4 * t + 3) * p.M + r];
        v = (q0 + q1) + (q2 + q3);
      } else {
        v = p.row_sumsq[(int64_t)t * p.M + r];
      }
      tsum[r * nT + t] = v;
    }
  }
  if constexpr (PAIR) {
    if (half == 1 && wave == 0) {      // the first half's sums: polled by ONE wave (the one whose lanes will read them), behind its own items
      int spins = 0;
      while (__hip_atomic_load(p.sync + strip0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 1) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 20)) break;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  auto signal_done = [&]() {      // CHAIN: every wave's stores are released before the workgroup counts itself done
    if constexpr (CHAIN) {
      // this wave's stores have reached the L2; ONE agent-scope release per workgroup then writes the L2's dirty lines back (the add carries it).
      // (Measured and rejected: write-through (sc1) stores + a relaxed add, no write-back -- 0.2 ms per step faster, but a consumer behind an
      // acquire read stale bytes: tests/test_kernels_gpu.py::test_decode_chain_equals_separate_launches failed.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const int before = __hip_atomic_fetch_add(done_ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // the LAST workgroup of the LAST phase: every wait of the launch has been passed -- the counters go back to zero for the next launch
        if (clean_ctr && before == nwg - 1)
          for (int i = 0; i < clean_n; ++i) __hip_atomic_store(clean_ctr + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if constexpr (PAIR) {
    if (half == 0) {      // 0.f + s0 + ... + s_{ksl - 1} of the M rows, handed to block wg + 1: [strip][j][row] x 16 bytes (M x N x 4 bytes: inside any split-K slab buffer)
      if (tid < p.M) {
        float y[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) y[e] = 0.f;
        const float* sp = slab + (size_t)tid * 16;
        for (int s = 0; s < ksl; ++s) {
#pragma unroll
          for (int e = 0; e < 16; e += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sp + (size_t)s * kRows * 16 + e);
            y[e] += v[0]; y[e + 1] += v[1]; y[e + 2] += v[2]; y[e + 3] += v[3];
          }
        }
        const __amdgpu_buffer_rsrc_t prs = strip_rsrc(p.partial + (size_t)strip0 * (p.M * 16));
#pragma unroll
        for (int j = 0; j < 4; ++j) strip_st16_sc1(prs, (unsigned)((j * p.M + tid) * 16), f32x4{y[4 * j], y[4 * j + 1], y[4 * j + 2], y[4 * j + 3]});
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(p.sync + strip0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  // ---- finish: the strip's slices in slab order, then the shared epilogue
  if (flive) do {
  float rs = 1.f;
  if (nT > 0) {
    float tot = 0.f;
    for (int t = 0; t < nT; ++t) tot += tsum[frw * nT + t];
    rs = rsqrtf(tot / (float)p.row_dim + p.row_eps);
  }
  auto strip_sum = [&](int sl, float (&y)[16]) {
    const float* sp = slab + ((size_t)sl * ksl * kRows + frw) * 16;
    if constexpr (PAIR) {      // continue the first half's chain of additions
      const __amdgpu_buffer_rsrc_t prs = strip_rsrc(p.partial + (size_t)strip0 * (p.M * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = strip_ld16_sc1(prs, (unsigned)((j * p.M + frw) * 16));
        y[4 * j] = v[0]; y[4 * j + 1] = v[1]; y[4 * j + 2] = v[2]; y[4 * j + 3] = v[3];
      }
      for (int s = 0; s < ksl; ++s) {
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sp + (size_t)s * kRows * 16 + e);
          y[e] += v[0]; y[e + 1] += v[1]; y[e + 2] += v[2]; y[e + 3] += v[3];
        }
      }
      return;
    }
    if (ks == 1) {
#pragma unroll
      for (int e = 0; e < 16; e += 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(sp + e); y[e] = v[0]; y[e + 1] = v[1]; y[e + 2] = v[2]; y[e + 3] = v[3]; }
      return;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) y[e] = 0.f;
    for (int s = 0; s < ks; ++s) {
#pragma unroll
      for (int e = 0; e < 16; e += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sp + (size_t)s * kRows * 16 + e);
        y[e] += v[0]; y[e + 1] += v[1]; y[e + 2] += v[2]; y[e + 3] += v[3];
      }
    }
  };
  float y[16];
  if (!swi) {
    strip_sum(fsl, y);
    segment_finish<T>(p, frw, fnc, fnc >> 6, true, y, rraw, rs, tid);
    break;
  }
  // fused SwiGLU (gate / up weight rows interleaved in blocks of 16: strip 2j = gate block j, strip 2j + 1 = its up block): the un-split
  // epilogue of the tiled kernel -- row factor, fma(y, 1, 0) + 0, both halves rounded to T, silu(gate) * up
  float u[16], o[16];
  strip_sum(2 * fsl, y);
  strip_sum(2 * fsl + 1, u);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float g = y[e], v = u[e];
    if (nT > 0) { g *= rs; v *= rs; }
    g = fmaf(g, 1.f, 0.f) + 0.f; v = fmaf(v, 1.f, 0.f) + 0.f;
    o[e] = silu(to_f32(from_f32<T>(g))) * to_f32(from_f32<T>(v));
  }
  const int64_t oc = (int64_t)frw * p.ldc + (fnc >> 5) * 16;
  if (p.out_f32) {
    float* Cf = reinterpret_cast<float*>(p.C) + oc;
#pragma unroll
    for (int e = 0; e < 16; e += 4) *reinterpret_cast<f32x4*>(Cf + e) = f32x4{o[e], o[e + 1], o[e + 2], o[e + 3]};
  } else {
    T* Ct = reinterpret_cast<T*>(p.C) + oc;
#pragma unroll
    for (int e = 0; e < 16; e += VEC) *reinterpret_cast<Chunk16*>(Ct + e) = chunk_from_f32<T>(o + e);
  }
  } while (0);
  if constexpr (PAIR) {
    if (tid == 0) __hip_atomic_store(p.sync + strip0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (its only reader is done: ready for the next launch)
  }
  signal_done();      // (every thread of the workgroup: one barrier)
}

template <typename T, int D, int NW, int MA>
__global__ __launch_bounds__(NW * 64) void decode_strip_kernel(const GemmArgs p, const int spw) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  strip_body<T, D, NW, MA, false>(p, spw, blockIdx.x, smem, nullptr, 0, nullptr, nullptr);
}

template <typename T, int D, int MA>
__global__ __launch_bounds__(256) void decode_strip_pair_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  strip_body<T, D, 4, MA, false, true>(p, 1, blockIdx.x, smem, nullptr, 0, nullptr, nullptr);
}

// Several decode-shaped GEMMs that feed each other (o_proj -> gate / up -> down_proj -> the next layer's LoRA-down -> QKV) as the phases of
// ONE launch: phase i's workgroups are blocks wg0[i] .. wg0[i + 1]; each waits for phase i - 1 as described above.  What a launch boundary
// costs these GEMMs is not the launch but the empty pipe behind it -- the first weights of a kernel are requested when its predecessor has
// drained; here the waiting workgroups of the next phases sit on the CUs with their rings full.
constexpr int kChainMax = 6;
struct StripChain {
  int n; int wg0[kChainMax + 1]; int spw[kChainMax];
  int* ctr;      // [n] workgroups done per phase: zero at launch, zeroed again by the launch's last workgroup
  int* err;      // counts waits given up (never, see strip_body)
  GemmArgs g[kChainMax];
};
template <typename T, int D, int NW, int MA>
__global__ __launch_bounds__(NW * 64) void decode_chain_kernel(const StripChain c) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  int ph = 0;
#pragma unroll
  for (int i = 1; i < kChainMax; ++i)
    if (i < c.n && (int)blockIdx.x >= c.wg0[i]) ph = i;
  ph = __builtin_amdgcn_readfirstlane(ph);
  strip_body<T, D, NW, MA, true>(c.g[ph], c.spw[ph], (int)blockIdx.x - c.wg0[ph], smem, ph > 0 ? c.ctr + ph - 1 : nullptr,
                                 ph > 0 ? c.wg0[ph] - c.wg0[ph - 1] : 0, c.ctr + ph, c.err, ph == c.n - 1 ? c.ctr : nullptr, c.n, c.wg0[ph + 1] - c.wg0[ph]);
}

template <typename T, int D, int NW, int MA>
int launch_strip(const GemmArgs& a, int spw, size_t smem, hipStream_t st) {
  auto kern = decode_strip_kernel<T, D, NW, MA>;
  static bool attr_set = false;
  if (!attr_set) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int nstrips = ceil_div(a.N, 16);
  hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(nstrips, spw)), dim3(NW * 64), smem, st, a, spw);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int D, int MA>
int launch_strip_pair(const GemmArgs& a, size_t smem, hipStream_t st) {
  auto kern = decode_strip_pair_kernel<T, D, MA>;
  static bool attr_set = false;
  if (!attr_set) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(2 * ceil_div(a.N, 16))), dim3(256), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

// Ring depth: what the CU's 160 KiB of LDS hold.  Launches of at most one workgroup per CU (every GEMM of a layer) get 32 KiB of ring per
// wave at 4 waves -- the short k-slices of the narrow projections (8-9 k-tiles) are then ONE round trip to memory, and gate / up keeps 64 KiB
// of weights in flight per CU; more workgroups than CUs (two per CU) and the 8-wave form: 16 KiB per wave.
template <typename T>
int launch_strip_t(const GemmArgs& a, hipStream_t st) {
  static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  const int ks = a.ksplit < 1 ? 1 : a.ksplit;
  const int nw = ks >= 8 ? 8 : 4;
  int spw = nw / ks < 1 ? 1 : nw / ks;
  if (a.swiglu_pairs && (spw & 1)) spw += 1;
  const int ma = a.M <= 8 ? 8 : 16;
  // two workgroups per strip (strip_body: PAIR) where one per strip leaves half the chip idle and the strips are long: >= 128 KiB of weights each
  const int nstrips = ceil_div(a.N, 16);
  static const bool pairs_on = knob("OAVSR_STRIP_PAIRS") != 0;
  if (pairs_on && a.partial && a.sync && !a.swiglu_pairs && ks >= 2 && ks % 2 == 0 && 2 * nstrips <= ncu && nstrips <= kSyncInts && a.N % 16 == 0 &&
      (size_t)16 * (a.K + a.K2) * sizeof(T) >= 128 * 1024) {
    const int d = ma == 8 ? 10 : 8;
    const size_t smem = (size_t)4 * d * slot_bytes(ma) + (size_t)(ks / 2) * 16 * 16 * 4 + (size_t)a.M * (a.row_sumsq ? a.row_sumsq_tiles : 0) * 4 + 16;
    OAVSR_REQUIRE(smem <= 160 * 1024, "decode strips: %zu bytes of LDS (ksplit %d, pairs)", smem, ks);
    return ma == 8 ? launch_strip_pair<T, 10, 8>(a, smem, st) : launch_strip_pair<T, 8, 16>(a, smem, st);
  }
  const bool deep = nw == 4 && ceil_div(ceil_div(a.N, 16), spw) <= ncu;
  const int d = deep ? (ma == 8 ? 10 : 8) : (ma == 8 ? 5 : 4);
  const size_t smem = (size_t)nw * d * slot_bytes(ma) + (size_t)spw * ks * 16 * 16 * 4 + (size_t)a.M * (a.row_sumsq ? a.row_sumsq_tiles : 0) * 4 + 16;
  OAVSR_REQUIRE(smem <= 160 * 1024, "decode strips: %zu bytes of LDS (ksplit %d)", smem, ks);
  if (nw == 8) return ma == 8 ? launch_strip<T, 5, 8, 8>(a, spw, smem, st) : launch_strip<T, 4, 8, 16>(a, spw, smem, st);
  if (deep) return ma == 8 ? launch_strip<T, 10, 4, 8>(a, spw, smem, st) : launch_strip<T, 8, 4, 16>(a, spw, smem, st);
  return ma == 8 ? launch_strip<T, 5, 4, 8>(a, spw, smem, st) : launch_strip<T, 4, 4, 16>(a, spw, smem, st);
}

// The chain's form: 4 waves per workgroup for every phase (an item's bits do not depend on the wave that computes it), ring depth by the
// knob: 2 = two workgroups per CU (half the ring each: more of the next phases resident and requesting), 1 = one.
template <typename T, int D, int MA>
int launch_chain(const StripChain& c, size_t smem, hipStream_t st) {
  auto kern = decode_chain_kernel<T, D, 4, MA>;
  static bool attr_set = false;
  if (!attr_set) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)c.wg0[c.n]), dim3(256), smem, st, c);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int prepare_strip(const GemmArgs& a, GemmArgs& b);

template <typename T>
int launch_chain_t(const GemmArgs* g, int n, int* ctr, int* err, hipStream_t st) {
  const int per_cu = knob("OAVSR_DECODE_CHAIN_PER_CU");
  StripChain c;
  c.n = n; c.ctr = ctr; c.err = err; c.wg0[0] = 0;
  const int ma = g[0].M <= 8 ? 8 : 16;
  const int d = per_cu >= 2 ? (ma == 8 ? 5 : 4) : (ma == 8 ? 10 : 8);
  size_t smem = 0;
  for (int i = 0; i < n; ++i) {
    OAVSR_TRY(prepare_strip(g[i], c.g[i]));
    const GemmArgs& a = c.g[i];
    const int ks = a.ksplit < 1 ? 1 : a.ksplit;
    int spw = 4 / ks < 1 ? 1 : 4 / ks;
    if (a.swiglu_pairs && (spw & 1)) spw += 1;
    c.spw[i] = spw;
    c.wg0[i + 1] = c.wg0[i] + ceil_div(ceil_div(a.N, 16), spw);
    smem = std::max(smem, (size_t)4 * d * slot_bytes(ma) + (size_t)spw * ks * 16 * 16 * 4 + (size_t)a.M * (a.row_sumsq ? a.row_sumsq_tiles : 0) * 4 + 16);
  }
  for (int i = n; i < kChainMax; ++i) { c.spw[i] = 1; c.wg0[i + 1] = c.wg0[n]; }
  OAVSR_REQUIRE(smem <= 160 * 1024, "decode chain: %zu bytes of LDS", smem);
  if (per_cu >= 2) return ma == 8 ? launch_chain<T, 5, 8>(c, smem, st) : launch_chain<T, 4, 16>(c, smem, st);
  return ma == 8 ? launch_chain<T, 10, 8>(c, smem, st) : launch_chain<T, 8, 16>(c, smem, st);
}

}  // namespace

// What the strips can serve: dense decode-shaped launches of <= kStripRows rows.  Next-norm outputs only in the strip layout
// (GemmArgs::rowsq_strips) -- a caller that asks for the tiled layout gets the tiled kernel.
bool gemm_strip_eligible(const GemmArgs& a) {
  if (a.amode != A_DENSE || a.groups != 1 || a.M < 1 || a.M > kStripRows || a.force_tile) return false;
  if (!(a.decode_tiled || a.ksplit > 1 || a.row_sumsq)) return false;
  if (a.ng_out && !a.rowsq_strips) return false;
  if (a.row_sumsq && a.row_sumsq_tiles > kStripMaxTiles) return false;
  if (a.ksplit > 64) return false;
  // the vocabulary-sized launch (lm_head): at 1-2 rows the strips stream it as fast as the 128 x 256 tile (step 1.13 vs 1.18 ms at one row), from a few
  // rows on the tile shares the activation rows between 256 columns and wins (8 rows: 104 vs 129 us, 16 rows: 121 vs 153; profiles/r06_decode_sites_by_rows.txt)
  // -- same bits, so the faster form
  if (a.N >= 65536 && a.M > 2) return false;
  const int vec = a.dtype == DT_F32 ? 4 : 8;
  if (a.swiglu_pairs && (a.ksplit > 1 || a.N % 32 != 0 || a.residual || a.act != ACT_NONE || a.col_scale || a.col_bias || a.ldc % vec != 0 || ((uintptr_t)a.C & 15) != 0))
    return false;
  return true;
}

namespace {
int prepare_strip(const GemmArgs& a, GemmArgs& b) {
  OAVSR_REQUIRE(gemm_strip_eligible(a), "decode strips: launch %d x %d x %d not eligible", a.M, a.N, a.K + a.K2);
  const int vec_t = a.dtype == DT_F32 ? 4 : 8;
  if (a.ksplit > 1 || a.ng_out)
    OAVSR_REQUIRE(a.N % 16 == 0 && ((uintptr_t)a.C & 15) == 0 && a.ldc % (a.out_f32 ? 4 : vec_t) == 0 &&
                  (!a.residual || (a.ldr % vec_t == 0 && ((uintptr_t)a.residual & 15) == 0)), "decode strips: a split launch needs N %% 16 == 0 and 16-byte rows");
  if (a.residual) OAVSR_REQUIRE(a.N % 16 == 0 && a.ldr % vec_t == 0 && ((uintptr_t)a.residual & 15) == 0, "decode strips: the residual needs N %% 16 == 0 and 16-byte rows");
  if (a.ng_out)
    OAVSR_REQUIRE(a.ng_gain && a.ng_rowsq && !a.out_f32 && a.act == ACT_NONE && a.N % 64 == 0 && a.ng_ld % vec_t == 0 && ((uintptr_t)a.ng_out & 15) == 0,
                  "decode strips: next-norm outputs need a plain epilogue, whole column tiles and 16-byte rows");
  OAVSR_REQUIRE(!a.row_sumsq || (a.row_sumsq_tiles >= 1 && a.row_dim > 0), "decode strips: the row factor needs row_sumsq_tiles >= 1 and row_dim > 0");
  b = a;
  b.rowsq_strips = a.ng_out ? 1 : 0;
  return 0;
}
}  // namespace

int launch_gemm_strip(const GemmArgs& a, hipStream_t st) {
  GemmArgs b;
  OAVSR_TRY(prepare_strip(a, b));
  return a.dtype == DT_F32 ? launch_strip_t<float>(b, st) : launch_strip_t<bf16_t>(b, st);
}

// n >= 2 strip launches, each consuming what the one before it produced, as ONE launch (decode_chain_kernel).  Same M and element type;
// ctr: n ints that are zero when the launch starts; err: one int that stays zero.
int launch_gemm_strip_chain(const GemmArgs* g, int n, int* ctr, int* err, hipStream_t st) {
  OAVSR_REQUIRE(g && n >= 1 && n <= kChainMax && ctr && err, "decode chain: bad arguments (n = %d)", n);
  for (int i = 1; i < n; ++i) OAVSR_REQUIRE(g[i].M == g[0].M && g[i].dtype == g[0].dtype, "decode chain: phases of different shapes");
  return g[0].dtype == DT_F32 ? launch_chain_t<float>(g, n, ctr, err, st) : launch_chain_t<bf16_t>(g, n, ctr, err, st);
}

}  // namespace oavsr
