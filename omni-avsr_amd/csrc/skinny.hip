// Skinny GEMM for the decode loop (M <= 128 rows: B utterances x K beams), gfx950.
//
// Weight-streaming regime: every weight byte is read exactly once per step, straight from HBM into the MFMA
// B-operand registers (lane l holds W[n0 + (l&15)][k + 8*(l>>4) ..] = one 16-byte global load, no LDS round
// trip); the (tiny, L2-resident) activations are the A operand.  A workgroup is 8 waves that interleave the
// 64-byte k blocks of the same NT*16 output columns (8 x U x NT KiB of weights in flight per workgroup) and combine
// their partial sums through LDS, then apply the fused epilogue.
//
// Fusions (decode only, all optional):
//   * second K segment = LoRA term (x A^T)(s B)^T  (Llama_LoRA.py:250-259)
//   * RMSNorm prologue: A1 rows are multiplied by the norm gain on the fly, sum(x^2) is accumulated from the very
//     same fragments, and rstd scales the segment-1 accumulator in the epilogue:  y = rstd * ((x*g) W^T) + t B^T
//     (LlamaRMSNorm -> q/k/v_proj, Llama_LoRA.py:624-645 via transformers)
//   * SwiGLU epilogue on a gate/up weight whose rows are interleaved in blocks of 16: out = silu(gate) * up
//
// HALF mode (M <= 8 rows, few column tiles: o_proj / down_proj / qkv of a B=8 greedy step): a workgroup owns 8 columns
// instead of 16, so N/8 workgroups cover the chip where N/16 would leave half the CUs idle.  The 16x16 MFMA is fed
// [8 columns x even k-blocks | the same 8 columns x odd k-blocks] on the weight side and [8 rows x even | 8 rows x odd] on
// the activation side; the two diagonal 8x8 blocks of the result are the even / odd partial sums (the off-diagonal blocks
// are discarded).  One wave load then covers 8 weight rows x one whole 128-byte line.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kMaxWaves = 8;       // waves per workgroup: 8, or 4 for the many-round wide kernels (template parameter W)

__device__ __forceinline__ Chunk16 zchunk() { Chunk16 c; c.v = u32x4{0u, 0u, 0u, 0u}; return c; }

template <typename T> struct MmaS;
template <> struct MmaS<bf16_t> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
  }
};
template <> struct MmaS<float> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    f32x4 fa = __builtin_bit_cast(f32x4, a.v), fb = __builtin_bit_cast(f32x4, b.v);
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb[s], c, 0, 0, 0);
  }
};

// 16 bytes through L2 (sc1): the only legal way to read what another workgroup of the SAME launch has stored (a CU's vector L1 is never
// refreshed by other CUs' stores; MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility")
__device__ __forceinline__ Chunk16 ld16_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  Chunk16 c;
  c.v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16);     // buffer_load_dwordx4 ... sc1 (aux bit 4 = sc1 on gfx950)
  return c;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t raw_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void st4_sc1(void* ptr, unsigned v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(ptr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One workgroup's tile of the weight-streaming GEMM.  CHAIN: the tile is a stage of a fused launch (skinny_chain2_kernel below):
//   * cs.wait_ctr: its activations are produced by lower-numbered workgroups of the same launch -- the weight loads go out first, then
//     one lane polls the stage's top counter (target = *cs.step x cs.wait_n: the counters only ever grow, decode step s expects s x n arrivals),
//     the workgroup barrier follows and the activations are read with sc1 loads;
//   * cs.signal_ctr: the outputs are stored write-through (sc1), every wave drains its stores, the workgroup meets at a barrier and
//     one lane bumps the counter (the MI355X hand-off recipe with sc1 stores / an agent-scope atomic add / sc1 polls and loads).
// Workgroups are dispatched in block-id order, so a consumer only ever waits for workgroups that are already resident or done; the
// spin is bounded and raises *cs.err instead of hanging if that assumption should ever fail.
template <typename T, int MT, int NT, bool NORM, int U, bool HALF, int W, bool CHAIN>
__device__ __forceinline__ void skinny_body(const GemmArgs& p, const int bid, char* smem, const SkinnySync& cs) {
  constexpr int kWaves = W, kThreads = W * 64;
  static_assert(!HALF || MT == 1, "HALF mode: at most 8 rows");
  constexpr int KBE = 64 / (int)sizeof(T);       // elements per 64-byte k block
  constexpr int VEC = TypeInfo<T>::kVec;
  // U = k blocks in flight per wave
  constexpr int MR = MT * 16, NC = NT * 16;
  float* sred = reinterpret_cast<float*>(smem);   // [8][MR][NC]
  float* sred2 = sred + kWaves * MR * NC;         // [8][MR][NC]   (NORM && K2)
  float* sssq = sred2 + ((NORM && p.K2) ? kWaves * MR * NC : 0);   // [8][MR]   (the second image exists only with a second K segment)
  float* sgain = sssq + (NORM ? kWaves * MR : 0);        // [K]  norm gain, staged once per workgroup (small tiles only)
  // staged norm gain: pays for single-round launches (8-wave workgroups); the many-round lm_head tiles (4 waves) read it from L2 instead --
  // no staging pass and no barrier in front of every tile's stream (109 -> 105 us).  Same values either way.
  constexpr bool GAIN_LDS = NORM && W == kMaxWaves && (kWaves * MR * NC * 8 + kWaves * MR * 4) <= 96 * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int hi = HALF ? (r16 >> 3) : 0, rw = HALF ? (r16 & 7) : r16;   // k-block parity, row within the operand tile
  constexpr int TW = HALF ? 8 : 16;                                    // output columns per MFMA tile
  const int n0 = bid * (NT * TW);
  const T* A1 = reinterpret_cast<const T*>(p.A);
  const T* A2 = reinterpret_cast<const T*>(p.A2);
  const T* W1 = reinterpret_cast<const T*>(p.W);
  const T* W2 = reinterpret_cast<const T*>(p.W2);
  const int nkb1 = p.K / KBE, nkb = nkb1 + p.K2 / KBE;

  f32x4 acc[MT][NT], acc2[NORM ? MT : 1][NORM ? NT : 1];
  float ssq[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    ssq[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (NORM) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // HALF: the loop walks PAIRS of k blocks (nkb1 is even there, so a pair never straddles the two K segments)
  const int nsteps = HALF ? (nkb + 1) / 2 : nkb;
  Chunk16 bf[U][NT], af[U][MT];
  // The weight stream does not depend on anything this kernel waits for, so the first trip's loads go out before the gain is staged
  // and before the epilogue operands / activations are fetched: their HBM latency runs under those L2 round trips.
  auto load_w = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = HALF ? 2 * (kb0 + kWaves * u) + hi : kb0 + kWaves * u;
      const bool in = kb < nkb;
      const bool s1 = kb < nkb1;
      const T* Wb = s1 ? W1 : W2;
      const int64_t ldw = s1 ? p.ldw : p.ldw2;
      const int k = (s1 ? kb : kb - nkb1) * KBE + g * VEC;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * TW + rw;
        if (in && n < p.N) bf[u][j].v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Wb + (int64_t)n * ldw + k));   // streamed once
        else bf[u][j] = zchunk();
      }
    }
  };
  if (wave < nsteps) load_w(wave);
  if (CHAIN && cs.wait_ctr) {
    // The weights are in flight; lane 0 first lets its own share land (by then the producers are usually done: the counter is not
    // hammered by hundreds of early pollers, whose requests would queue in front of the producers' atomic adds), then polls the top
    // counter with a back-off.
    if (tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int target = *cs.step * cs.wait_n;
      int spins = 0;
      while (__hip_atomic_load(cs.wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1 << 20)) { if (cs.err) *cs.err = 1; break; }
      }
    }
    if (!GAIN_LDS) __syncthreads();                      // (with a staged gain the barrier below does it)
  }
  if (GAIN_LDS) {                                        // K floats, 16 B per thread and trip; read back after the first barrier-free
    for (int i = tid * 4; i < p.K; i += kThreads * 4)   // use is safe: every wave waits on its own operands, then on this barrier
      *reinterpret_cast<f32x4*>(sgain + i) = *reinterpret_cast<const f32x4*>(p.norm_gain + i);
    __syncthreads();
  }
  // epilogue operands of this thread's first output element are fetched now, under the weight stream
  constexpr int OR = HALF ? 8 : MR, OC = NT * TW;         // output tile of the workgroup
  const T* R = reinterpret_cast<const T*>(p.residual);
  float pre_res = 0.f, pre_scale = 1.f, pre_bias = 0.f;
  if (!p.swiglu_pairs && tid < OR * OC) {
    const int m = tid / OC, n = n0 + (tid - m * OC);
    if (m < p.M && n < p.N) {
      if (R) pre_res = to_f32(R[(int64_t)m * p.ldr + n]);
      if (p.col_scale) pre_scale = p.col_scale[n];
      if (p.col_bias) pre_bias = p.col_bias[n];
    }
  }
  for (int kb0 = wave; kb0 < nsteps; kb0 += kWaves * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = HALF ? 2 * (kb0 + kWaves * u) + hi : kb0 + kWaves * u;
      const bool in = kb < nkb;
      const bool s1 = kb < nkb1;
      const T* Ab = s1 ? A1 : A2;
      const int64_t lda = s1 ? p.lda : p.lda2;
      const int k = (s1 ? kb : kb - nkb1) * KBE + g * VEC;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = i * 16 + rw;
        if (CHAIN && (cs.a1_sc1 | cs.a2_sc1)) {
          // produced inside this launch: 16-byte sc1 buffer loads.  The segment test is made provably wave-uniform (a scalar branch
          // between the two kernel-argument descriptors; a per-lane descriptor would be loaded in a waterfall loop).
          const bool s1u = __builtin_amdgcn_readfirstlane((int)((HALF ? 2 * (kb0 + kWaves * u) : kb0 + kWaves * u) < nkb1)) != 0;
          const unsigned off = (unsigned)(((int64_t)m * lda + k) * (int64_t)sizeof(T));
          if (!(in && m < p.M)) af[u][i] = zchunk();
          else if (s1u) af[u][i] = cs.a1_sc1 ? ld16_sc1(raw_rsrc(p.A), off) : *reinterpret_cast<const Chunk16*>(A1 + (int64_t)m * lda + k);
          else af[u][i] = cs.a2_sc1 ? ld16_sc1(raw_rsrc(p.A2), off) : *reinterpret_cast<const Chunk16*>(A2 + (int64_t)m * lda + k);
        } else {
          af[u][i] = (in && m < p.M) ? *reinterpret_cast<const Chunk16*>(Ab + (int64_t)m * lda + k) : zchunk();
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = HALF ? 2 * (kb0 + kWaves * u) + hi : kb0 + kWaves * u;
      const bool s1 = (HALF ? 2 * (kb0 + kWaves * u) : kb) < nkb1;      // wave-uniform
      if (NORM && s1 && kb < nkb) {
        const int k = kb * KBE + g * VEC;
        float gn[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
          f32x4 t = *reinterpret_cast<const f32x4*>((GAIN_LDS ? sgain : p.norm_gain) + k + e);
          gn[e] = t[0]; gn[e + 1] = t[1]; gn[e + 2] = t[2]; gn[e + 3] = t[3];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          float f[VEC];
          chunk_to_f32<T>(af[u][i], f);
#pragma unroll
          for (int e = 0; e < VEC; ++e) { ssq[i] = fmaf(f[e], f[e], ssq[i]); f[e] *= gn[e]; }   // explicit FMA: same bits in every variant
          af[u][i] = chunk_from_f32<T>(f);
        }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (NORM && !s1) MmaS<T>::run(af[u][i], bf[u][j], acc2[i][j]);
          else MmaS<T>::run(af[u][i], bf[u][j], acc[i][j]);
        }
    }
    if (kb0 + kWaves * U < nsteps) load_w(kb0 + kWaves * U);      // next trip's weights: requested as soon as the registers are free
  }
  // partial sums -> LDS (C layout: col = lane&15, row = 4*(lane>>4)+r)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = (wave * MR + i * 16 + g * 4 + r) * NC + j * 16 + r16;
        sred[o] = acc[i][j][r];
        if (NORM && p.K2) sred2[o] = acc2[i][j][r];
      }
  if (NORM) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float v = ssq[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) sssq[wave * MR + i * 16 + r16] = v;
    }
  }
  __syncthreads();
  auto raw = [&](int m, int nn) -> float {
    float y = 0.f;
    const int lc = HALF ? ((nn >> 3) * 16 + (nn & 7)) : nn;      // column inside the LDS image (16 per MFMA tile)
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      y += sred[(w * MR + m) * NC + lc];
      if (HALF) y += sred[(w * MR + m + 8) * NC + lc + 8];
    }
    if (NORM) {
      float s = 0.f, y2 = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        s += sssq[w * MR + m];
        if (HALF) s += sssq[w * MR + m + 8];
      }
      if (p.K2) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
          y2 += sred2[(w * MR + m) * NC + lc];
          if (HALF) y2 += sred2[(w * MR + m + 8) * NC + lc + 8];
        }
      }
      y = fmaf(y, rsqrtf(s / (float)p.K + p.norm_eps), y2);
    }
    return y;
  };
  auto value = [&](int m, int nn) -> float {
    const int n = n0 + nn;
    return fmaf(raw(m, nn), p.col_scale ? p.col_scale[n] : 1.f, p.col_bias ? p.col_bias[n] : 0.f);
  };
  if (CHAIN && cs.signal_ctr) {
    // producer stage: same values as the plain epilogues below, stored as 4-byte write-through words (two bf16 columns or one f32), then
    // drain -> barrier -> one agent-scope add
    constexpr int CPW = 4 / (int)sizeof(T);              // output columns per stored word
    const int ocols = p.swiglu_pairs ? OC / 2 : OC;
    for (int idx = tid; idx < OR * (ocols / CPW); idx += kThreads) {
      const int m = idx / (ocols / CPW), q0 = (idx - m * (ocols / CPW)) * CPW;
      float y[CPW];
      int ncol0 = 0;
      bool ok = m < p.M;
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int q = q0 + c;
        if (p.swiglu_pairs) {
          const int pblk = q >> 4, i = q & 15;
          if (c == 0) ncol0 = (n0 >> 1) + q;
          ok = ok && (n0 + pblk * 32 + i < p.N);
          y[c] = ok ? silu(value(m, pblk * 32 + i)) * value(m, pblk * 32 + 16 + i) : 0.f;
        } else {
          const int n = n0 + q;
          if (c == 0) ncol0 = n;
          ok = ok && n < p.N;
          float v = ok ? value(m, q) : 0.f;
          const float res = (ok && R) ? to_f32(R[(int64_t)m * p.ldr + n]) : 0.f;
          if (!p.res_after_act) v += res;
          switch (p.act) {
            case ACT_GELU: v = gelu_act<T>(v); break;
            case ACT_RELU: v = fmaxf(v, 0.f); break;
            case ACT_PRELU: v = v >= 0.f ? v : p.prelu[ok ? n : 0] * v; break;
            case ACT_SILU: v = silu(v); break;
            default: break;
          }
          if (p.res_after_act) v += res;
          y[c] = v;
        }
      }
      if (!ok) continue;                                 // (N and the tile width are even: a word never straddles the edge)
      if (p.out_f32 || sizeof(T) == 4) st4_sc1(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + ncol0, __float_as_uint(y[0]));
      else {
        const bf16x2 h = {(bf16_t)y[0], (bf16_t)y[CPW - 1]};
        st4_sc1(reinterpret_cast<bf16_t*>(p.C) + (int64_t)m * p.ldc + ncol0, __builtin_bit_cast(unsigned, h));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores ...
    __syncthreads();                                     // ... before the lane that signals for all of them
    // fan-in in two levels (one counter would serialise hundreds of arrivals at ~12 ns each): eight shard counters on lines of their
    // own, picked by bid % 8 (= the XCD under round-robin placement; speed only); the last arriver of a shard bumps the top counter
    if (tid == 0) {
      const int shard = bid & 7, step = *cs.step;
      const int mine = (cs.signal_n + 7 - shard) >> 3;                 // stage-0 workgroups in this shard
      const int old = __hip_atomic_fetch_add(cs.signal_ctr + 32 * (1 + shard), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == step * mine) __hip_atomic_fetch_add(cs.signal_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (p.swiglu_pairs) {
    // weight rows interleaved [gate block b | up block b] in 16-row blocks; out column = (n0/32 + pblk)*16 + i
    for (int idx = tid; idx < OR * (OC / 2); idx += kThreads) {
      const int m = idx / (OC / 2), q = idx - m * (OC / 2);
      const int pblk = q >> 4, i = q & 15;
      const int ncol = (n0 >> 1) + q;
      if (m >= p.M || n0 + pblk * 32 + i >= p.N) continue;
      float y = silu(value(m, pblk * 32 + i)) * value(m, pblk * 32 + 16 + i);
      if (p.out_f32) reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + ncol] = y;
      else reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + ncol] = from_f32<T>(y);
    }
    return;
  }
  for (int idx = tid; idx < OR * OC; idx += kThreads) {
    const int m = idx / OC, nn = idx - m * OC;
    const int n = n0 + nn;
    if (m >= p.M || n >= p.N) continue;
    const bool first = idx == tid;                 // prefetched operands
    float y = first ? fmaf(raw(m, nn), pre_scale, pre_bias) : value(m, nn);
    float res = first ? pre_res : (R ? to_f32(R[(int64_t)m * p.ldr + n]) : 0.f);
    if (!p.res_after_act) y += res;
    switch (p.act) {
      case ACT_GELU: y = gelu_act<T>(y); break;
      case ACT_RELU: y = fmaxf(y, 0.f); break;
      case ACT_PRELU: y = y >= 0.f ? y : p.prelu[n] * y; break;
      case ACT_SILU: y = silu(y); break;
      default: break;
    }
    if (p.res_after_act) y += res;
    if (p.out_f32) reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n] = y;
    else reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + n] = from_f32<T>(y);
  }
}

// 8-column tiles for 9..64 rows (MT = 1: up to 16 rows, MT = 2: up to 32, MT = 4: up to 64 -- round 4, the decode scheduler's 64-row
// steps: the weight stream of a step is what costs, twice the rows share it).  The weight fragment is HALF mode's -- MFMA rows 0-7 hold the
// tile's 8 columns for the EVEN k block of a pair, rows 8-15 the same columns for the ODD block, so a wave load is 8 weight rows x one whole
// 128-byte line and N / 8 workgroups fill the chip where 16-column tiles left half the CUs idle -- but the activation operand keeps all
// 16 rows: two MFMAs per pair, one with the even-block activations (its result columns 0-7 are the even partial sums), one with the
// odd-block activations (columns 8-15 = the odd partial sums); the other halves are discarded.  MFMA time is irrelevant here, the
// weight stream is what costs.  The partial sums are combined exactly as HALF mode combines them (per wave: even, then odd), so a row
// gets the same bits from this kernel as from the <= 8-row kernel: greedy decode does not depend on the batch up to 32 rows.
// SEG2 = false: the launch has no second K segment (K2 == 0; launcher-checked): with the fused norm the LoRA segment's accumulators -- as many
// registers again -- then do not exist (the 64-row x 32-column forms only fit the register file that way).
template <typename T, int MT, int NT, bool NORM, int U, int W, bool SEG2 = true>
__global__ __launch_bounds__(W * 64, (NT <= 2 && MT == 1) ? 4 : 2) void skinny_r16_kernel(const GemmArgs p) {
  constexpr int kWaves = W, kThreads = W * 64;
  constexpr int KBE = 64 / (int)sizeof(T);
  constexpr int VEC = TypeInfo<T>::kVec;
  constexpr int MR = MT * 16, NC = NT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sred = reinterpret_cast<float*>(smem);                   // [8][MR][NC]   columns 0-7 of a 16-block: even partials, 8-15: odd
  float* sred2 = sred + kWaves * MR * NC;                         // [8][MR][NC]   (NORM && K2: the LoRA segment's partial sums, kept apart from the normalised ones)
  float* sssq = sred2 + ((NORM && p.K2) ? kWaves * MR * NC : 0);  // [2][8][MR]    sum of squares per (parity, wave, row)
  float* sgain = sssq + (NORM ? 2 * kWaves * MR : 0);             // [K]
  constexpr bool GAIN_LDS = NORM && (kWaves * MR * NC * 8 + 2 * kWaves * MR * 4) <= 96 * 1024;      // (sized for the two-image case: a compile-time property)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int par = r16 >> 3, cw = r16 & 7;                         // weight operand: k-block parity, column within the tile
  const int n0 = blockIdx.x * (NT * 8);
  const T* A1 = reinterpret_cast<const T*>(p.A);
  const T* A2 = reinterpret_cast<const T*>(p.A2);
  const T* W1 = reinterpret_cast<const T*>(p.W);
  const T* W2 = reinterpret_cast<const T*>(p.W2);
  const int nkb1 = p.K / KBE, nkb = nkb1 + p.K2 / KBE;
  const int nsteps = (nkb + 1) / 2;                               // pairs of k blocks (nkb1 even: a pair never straddles the segments)

  f32x4 acc[MT][NT][2], acc2[(NORM && SEG2) ? MT : 1][(NORM && SEG2) ? NT : 1][2];
  float ssq[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      ssq[i][q] = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (NORM && SEG2) acc2[i][j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  Chunk16 bf[U][NT], af[U][MT][2];
  auto load_w = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = 2 * (kb0 + kWaves * u) + par;
      const bool in = kb < nkb, s1 = kb < nkb1;
      const T* Wb = s1 ? W1 : W2;
      const int64_t ldw = s1 ? p.ldw : p.ldw2;
      const int k = (s1 ? kb : kb - nkb1) * KBE + g * VEC;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 8 + cw;
        if (in && n < p.N) bf[u][j].v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Wb + (int64_t)n * ldw + k));
        else bf[u][j] = zchunk();
      }
    }
  };
  if (wave < nsteps) load_w(wave);
  if (GAIN_LDS) {
    for (int i = tid * 4; i < p.K; i += kThreads * 4)
      *reinterpret_cast<f32x4*>(sgain + i) = *reinterpret_cast<const f32x4*>(p.norm_gain + i);
    __syncthreads();
  }
  for (int kb0 = wave; kb0 < nsteps; kb0 += kWaves * U) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int kb = 2 * (kb0 + kWaves * u) + q;
        const bool in = kb < nkb, s1 = kb < nkb1;
        const T* Ab = s1 ? A1 : A2;
        const int64_t lda = s1 ? p.lda : p.lda2;
        const int k = (s1 ? kb : kb - nkb1) * KBE + g * VEC;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int m = i * 16 + r16;
          af[u][i][q] = (in && m < p.M) ? *reinterpret_cast<const Chunk16*>(Ab + (int64_t)m * lda + k) : zchunk();
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool s1 = 2 * (kb0 + kWaves * u) < nkb1;               // wave-uniform
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int kb = 2 * (kb0 + kWaves * u) + q;
        if (NORM && s1 && kb < nkb) {
          const int k = kb * KBE + g * VEC;
          float gn[VEC];
#pragma unroll
          for (int e = 0; e < VEC; e += 4) {
            f32x4 t = *reinterpret_cast<const f32x4*>((GAIN_LDS ? sgain : p.norm_gain) + k + e);
            gn[e] = t[0]; gn[e + 1] = t[1]; gn[e + 2] = t[2]; gn[e + 3] = t[3];
          }
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            float f[VEC];
            chunk_to_f32<T>(af[u][i][q], f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { ssq[i][q] = fmaf(f[e], f[e], ssq[i][q]); f[e] *= gn[e]; }
            af[u][i][q] = chunk_from_f32<T>(f);
          }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (NORM && SEG2 && !s1) MmaS<T>::run(af[u][i][q], bf[u][j], acc2[i][j][q]);
            else MmaS<T>::run(af[u][i][q], bf[u][j], acc[i][j][q]);
          }
      }
    }
    if (kb0 + kWaves * U < nsteps) load_w(kb0 + kWaves * U);
  }
  // partial sums -> LDS (C layout: col = lane & 15, row = 4 * (lane >> 4) + r): columns 0-7 from the even-block MFMA, 8-15 from the odd one
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = (wave * MR + i * 16 + g * 4 + r) * NC + j * 16 + r16;
        sred[o] = par ? acc[i][j][1][r] : acc[i][j][0][r];
        if (NORM && SEG2 && p.K2) sred2[o] = par ? acc2[i][j][1][r] : acc2[i][j][0][r];
      }
  if (NORM) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v = ssq[i][q];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) sssq[(q * kWaves + wave) * MR + i * 16 + r16] = v;
      }
  }
  __syncthreads();
  constexpr int OC = NT * 8;
  auto raw = [&](int m, int nn) -> float {
    float y = 0.f;
    const int lc = (nn >> 3) * 16 + (nn & 7);
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      y += sred[(w * MR + m) * NC + lc];
      y += sred[(w * MR + m) * NC + lc + 8];
    }
    if (NORM) {
      float s = 0.f, y2 = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        s += sssq[w * MR + m];
        s += sssq[(kWaves + w) * MR + m];
      }
      if (p.K2) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
          y2 += sred2[(w * MR + m) * NC + lc];
          y2 += sred2[(w * MR + m) * NC + lc + 8];
        }
      }
      y = fmaf(y, rsqrtf(s / (float)p.K + p.norm_eps), y2);
    }
    return y;
  };
  auto value = [&](int m, int nn) -> float {
    const int n = n0 + nn;
    return fmaf(raw(m, nn), p.col_scale ? p.col_scale[n] : 1.f, p.col_bias ? p.col_bias[n] : 0.f);
  };
  const T* R = reinterpret_cast<const T*>(p.residual);
  if (p.swiglu_pairs) {
    for (int idx = tid; idx < MR * (OC / 2); idx += kThreads) {
      const int m = idx / (OC / 2), q = idx - m * (OC / 2);
      const int pblk = q >> 4, i = q & 15;
      const int ncol = (n0 >> 1) + q;
      if (m >= p.M || n0 + pblk * 32 + i >= p.N) continue;
      float y = silu(value(m, pblk * 32 + i)) * value(m, pblk * 32 + 16 + i);
      if (p.out_f32) reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + ncol] = y;
      else reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + ncol] = from_f32<T>(y);
    }
    return;
  }
  for (int idx = tid; idx < MR * OC; idx += kThreads) {
    const int m = idx / OC, nn = idx - m * OC;
    const int n = n0 + nn;
    if (m >= p.M || n >= p.N) continue;
    float y = value(m, nn);
    const float res = R ? to_f32(R[(int64_t)m * p.ldr + n]) : 0.f;
    if (!p.res_after_act) y += res;
    switch (p.act) {
      case ACT_GELU: y = gelu_act<T>(y); break;
      case ACT_RELU: y = fmaxf(y, 0.f); break;
      case ACT_PRELU: y = y >= 0.f ? y : p.prelu[n] * y; break;
      case ACT_SILU: y = silu(y); break;
      default: break;
    }
    if (p.res_after_act) y += res;
    if (p.out_f32) reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n] = y;
    else reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + n] = from_f32<T>(y);
  }
}

template <typename T, int MT, int NT, bool NORM, int U, int W = kMaxWaves, bool SEG2 = true>
int launch_r16(const GemmArgs& a, hipStream_t st) {
  OAVSR_REQUIRE(SEG2 || a.K2 == 0, "skinny gemm: this form has no second K segment");
  constexpr int img = W * MT * 16 * NT * 16 * 4;
  constexpr int smem_two = img * (NORM ? 2 : 1) + (NORM ? 2 * W * MT * 16 * 4 : 0);
  constexpr bool gain_lds = NORM && smem_two <= 96 * 1024;                        // the kernel's GAIN_LDS
  const int smem0 = img * ((NORM && a.K2) ? 2 : 1) + (NORM ? 2 * W * MT * 16 * 4 : 0);      // the second image only exists with a LoRA segment
  const int smem = smem0 + (gain_lds ? a.K * 4 : 0);
  OAVSR_REQUIRE(smem <= 160 * 1024, "skinny gemm: %d rows x %d columns x K=%d do not fit the LDS of a CU (%d bytes)", MT * 16, NT * 8, a.K, smem);
  auto kern = skinny_r16_kernel<T, MT, NT, NORM, U, W, SEG2>;
  static bool attr_set = false;
  if (!attr_set) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(ceil_div(a.N, NT * 8)), dim3(W * 64), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// lm_head-shaped GEMV (<= 8 rows, one K segment, thousands of 32-column tiles, plain epilogue): the SAME tile arithmetic as skinny_body's
// HALF / NT = 4 / 4-wave variant -- same lane <-> element mapping, same k order per wave, same cross-wave sums, hence the same bits --
// run by PERSISTENT workgroups that walk tiles bid, bid + grid, ...:
//   * the activation fragments of a wave (its k blocks of the <= 8 rows, times the norm gain) and the rows' 1 / rms do not depend on the
//     tile: loaded, normalised and reduced ONCE per workgroup instead of once per tile;
//   * the weight stream never drains: the first trip of the next tile is requested before the last trip of the current one is consumed,
//     so a wave that waits at the tile's barrier (partial sums -> LDS -> 256 outputs) already has 8 KiB in flight again; the LDS image
//     is double-buffered, one barrier per tile.
// AS = k steps (pairs of 64-byte k blocks) per wave: 8 covers K <= 2048 (bf16), 16 K <= 4096.
// W = 4: the lm_head class; W = 8 with the SwiGLU epilogue: gate / up (the wave count is part of the summation order: each form keeps the
// count of the per-tile kernel it replaces; with 8 waves only the first 256 threads own an output element).
template <typename T, bool NORM, int AS, int W = 4, bool SWIGLU = false>
__global__ __launch_bounds__(W * 64, W == 8 ? 1 : (AS <= 8 ? 3 : 2)) void skinny_persist_kernel(const GemmArgs p, const int ntiles) {
  constexpr int NT = 4, U = 2, TPT = AS / U, MR = 16, NC = 64;
  constexpr int KBE = 64 / (int)sizeof(T), VEC = TypeInfo<T>::kVec;
  static_assert(TPT % 2 == 0, "the weight double buffer alternates per trip");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sred = reinterpret_cast<float*>(smem);            // [2][W][MR][NC]
  float* sssq = sred + 2 * W * MR * NC;                    // [W][MR]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4, hi = r16 >> 3, rw = r16 & 7;
  const T* A1 = reinterpret_cast<const T*>(p.A);
  const T* W1 = reinterpret_cast<const T*>(p.W);
  const int nsteps = (p.K / KBE) / 2;                      // K / KBE is even (launcher)

  Chunk16 bf[2][U][NT];
  auto issue = [&](const int tile, const int trip, auto buf_c) __attribute__((always_inline)) {
    constexpr int B = decltype(buf_c)::value;
    const int n0 = tile * (NT * 8);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int st = wave + W * (trip * U + u);
      const int k = (2 * st + hi) * KBE + g * VEC;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 8 + rw;
        if (st < nsteps && n < p.N) bf[B][u][j].v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W1 + (int64_t)n * p.ldw + k));
        else bf[B][u][j] = zchunk();
      }
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) issue(tile, 0, std::integral_constant<int, 0>{});     // the stream starts before anything else is fetched

  // ---- once per workgroup: this wave's activation fragments (x gain), the rows' sums of squares
  Chunk16 afn[AS];
  float ssq = 0.f;
#pragma unroll
  for (int i = 0; i < AS; ++i) {
    const int st = wave + W * i;
    const int k = (2 * st + hi) * KBE + g * VEC;
    afn[i] = (st < nsteps && rw < p.M) ? *reinterpret_cast<const Chunk16*>(A1 + (int64_t)rw * p.lda + k) : zchunk();
  }
  if (NORM) {
#pragma unroll
    for (int i = 0; i < AS; ++i) {
      const int st = wave + W * i;
      if (st < nsteps) {
        const int k = (2 * st + hi) * KBE + g * VEC;
        float gn[VEC], f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(p.norm_gain + k + e);
          gn[e] = t[0]; gn[e + 1] = t[1]; gn[e + 2] = t[2]; gn[e + 3] = t[3];
        }
        chunk_to_f32<T>(afn[i], f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { ssq = fmaf(f[e], f[e], ssq); f[e] *= gn[e]; }   // the expressions of skinny_body, in its order
        afn[i] = chunk_from_f32<T>(f);
      }
    }
    float v = ssq;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (g == 0) sssq[wave * MR + r16] = v;
    __syncthreads();
  }
  // one output element per thread (of the first 256) and tile: row m, column nn of the 8 x 32 tile
  const int m = (tid >> 5) & 7, nn = tid & 31;
  const bool owner = tid < 256;
  const int lc = (nn >> 3) * 16 + (nn & 7);
  float rs = 1.f;
  if (NORM) {
    float sq = 0.f;
#pragma unroll
    for (int w = 0; w < W; ++w) { sq += sssq[w * MR + m]; sq += sssq[w * MR + m + 8]; }
    rs = rsqrtf(sq / (float)p.K + p.norm_eps);
  }

  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int par = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    static_for<0, TPT>([&](auto TR) {
      constexpr int trip = decltype(TR)::value;
      if constexpr (trip + 1 < TPT) issue(tile, trip + 1, std::integral_constant<int, (trip + 1) & 1>{});
      else if (tile + (int)gridDim.x < ntiles) issue(tile + (int)gridDim.x, 0, std::integral_constant<int, 0>{});
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NT; ++j) MmaS<T>::run(afn[trip * U + u], bf[trip & 1][u][j], acc[j]);
    });
    float* sr = sred + par * (W * MR * NC);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sr[(wave * MR + g * 4 + r) * NC + j * 16 + r16] = acc[j][r];
      acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int n = tile * (NT * 8) + nn;
    float y = 0.f;
    if (owner && n < p.N) {
#pragma unroll
      for (int w = 0; w < W; ++w) { y += sr[(w * MR + m) * NC + lc]; y += sr[(w * MR + m + 8) * NC + lc + 8]; }
      if (NORM) y = fmaf(y, rs, 0.f);
      y = fmaf(y, p.col_scale ? p.col_scale[n] : 1.f, p.col_bias ? p.col_bias[n] : 0.f);
    }
    if (SWIGLU) {
      // weight rows interleaved [gate block | up block] in 16-row blocks: column nn < 16 is a gate, nn + 16 (16 lanes up) its up value
      const float up = __shfl_down(y, 16, 64);
      if (owner && nn < 16 && m < p.M && n + 16 < p.N) {
        const float o = silu(y) * up;
        const int ncol = tile * (NT * 4) + nn;
        if (p.out_f32) reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + ncol] = o;
        else reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + ncol] = from_f32<T>(o);
      }
    } else if (owner && m < p.M && n < p.N) {
      if (p.out_f32) reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n] = y;
      else reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + n] = from_f32<T>(y);
    }
    par ^= 1;
  }
}

// The same for 9..32 rows: skinny_r16_kernel's tile arithmetic (16-row activation operands, one MFMA per k-block parity; MT = 1: <= 16 rows,
// MT = 2: <= 32) on persistent 4-wave workgroups.  Here the activation fragments are as many bytes per tile as the weights (32 rows x K
// against 32 columns x K), so fetching and normalising them once per workgroup instead of once per tile halves what a CU pulls in.
// W = 4 for the lm_head class (N >= 65536), W = 8 for the other wide GEMVs (gate / up with the SwiGLU epilogue): the wave count is part of
// the summation order, so each form keeps the count of the per-tile kernel it replaces.
template <typename T, int MT, bool NORM, int AS, int W = 4, bool SWIGLU = false>
__global__ __launch_bounds__(W * 64, (W == 4 && MT == 1 && AS <= 8) ? 2 : 1) void skinny_persist_r16_kernel(const GemmArgs p, const int ntiles) {
  constexpr int NT = 4, U = 2, TPT = AS / U, MR = MT * 16, NC = 64, kThreads = W * 64;
  constexpr int KBE = 64 / (int)sizeof(T), VEC = TypeInfo<T>::kVec;
  static_assert(TPT % 2 == 0, "the weight double buffer alternates per trip");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sred = reinterpret_cast<float*>(smem);            // [2][W][MR][NC]
  float* sssq = sred + 2 * W * MR * NC;                    // [2][W][MR]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4, par = r16 >> 3, cw = r16 & 7;
  const T* A1 = reinterpret_cast<const T*>(p.A);
  const T* W1 = reinterpret_cast<const T*>(p.W);
  const int nsteps = (p.K / KBE) / 2;

  Chunk16 bf[2][U][NT];
  auto issue = [&](const int tile, const int trip, auto buf_c) __attribute__((always_inline)) {
    constexpr int B = decltype(buf_c)::value;
    const int n0 = tile * (NT * 8);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int st = wave + W * (trip * U + u);
      const int k = (2 * st + par) * KBE + g * VEC;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 8 + cw;
        if (st < nsteps && n < p.N) bf[B][u][j].v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W1 + (int64_t)n * p.ldw + k));
        else bf[B][u][j] = zchunk();
      }
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) issue(tile, 0, std::integral_constant<int, 0>{});

  // ---- once per workgroup: activation fragments (both k-block parities of every step, x gain), sums of squares per (row, parity)
  Chunk16 afn[AS][MT][2];
  float ssq[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i) { ssq[i][0] = 0.f; ssq[i][1] = 0.f; }
#pragma unroll
  for (int s_ = 0; s_ < AS; ++s_) {
    const int st = wave + W * s_;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = (2 * st + q) * KBE + g * VEC;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = i * 16 + r16;
        afn[s_][i][q] = (st < nsteps && m < p.M) ? *reinterpret_cast<const Chunk16*>(A1 + (int64_t)m * p.lda + k) : zchunk();
      }
    }
  }
  if (NORM) {
#pragma unroll
    for (int s_ = 0; s_ < AS; ++s_) {
      const int st = wave + W * s_;
      if (st < nsteps) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int k = (2 * st + q) * KBE + g * VEC;
          float gn[VEC];
#pragma unroll
          for (int e = 0; e < VEC; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p.norm_gain + k + e);
            gn[e] = t[0]; gn[e + 1] = t[1]; gn[e + 2] = t[2]; gn[e + 3] = t[3];
          }
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            float f[VEC];
            chunk_to_f32<T>(afn[s_][i][q], f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { ssq[i][q] = fmaf(f[e], f[e], ssq[i][q]); f[e] *= gn[e]; }   // skinny_r16_kernel's expressions and order
            afn[s_][i][q] = chunk_from_f32<T>(f);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v = ssq[i][q];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) sssq[(q * W + wave) * MR + i * 16 + r16] = v;
      }
    __syncthreads();
  }
  // MR * 32 / threads output elements per thread and tile: idx = tid + t * threads -> row idx / 32, column idx % 32
  static_assert(MR * 32 % kThreads == 0, "outputs per thread");
  constexpr int OPT = MR * 32 / kThreads;
  float rs[OPT];
#pragma unroll
  for (int t = 0; t < OPT; ++t) {
    rs[t] = 1.f;
    if (NORM) {
      const int m = (tid + t * kThreads) >> 5;
      float sq = 0.f;
#pragma unroll
      for (int w = 0; w < W; ++w) { sq += sssq[w * MR + m]; sq += sssq[(W + w) * MR + m]; }
      rs[t] = rsqrtf(sq / (float)p.K + p.norm_eps);
    }
  }

  f32x4 acc[MT][NT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) { acc[i][j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  int parity = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    static_for<0, TPT>([&](auto TR) {
      constexpr int trip = decltype(TR)::value;
      if constexpr (trip + 1 < TPT) issue(tile, trip + 1, std::integral_constant<int, (trip + 1) & 1>{});
      else if (tile + (int)gridDim.x < ntiles) issue(tile + (int)gridDim.x, 0, std::integral_constant<int, 0>{});
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) MmaS<T>::run(afn[trip * U + u][i][q], bf[trip & 1][u][j], acc[i][j][q]);
    });
    float* sr = sred + parity * (W * MR * NC);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sr[(wave * MR + i * 16 + g * 4 + r) * NC + j * 16 + r16] = par ? acc[i][j][1][r] : acc[i][j][0][r];
        acc[i][j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < OPT; ++t) {
      const int idx = tid + t * kThreads, m = idx >> 5, nn = idx & 31;
      const int n = tile * (NT * 8) + nn;
      float y = 0.f;
      if (n < p.N) {
        const int lc = (nn >> 3) * 16 + (nn & 7);
#pragma unroll
        for (int w = 0; w < W; ++w) { y += sr[(w * MR + m) * NC + lc]; y += sr[(w * MR + m) * NC + lc + 8]; }
        if (NORM) y = fmaf(y, rs[t], 0.f);
        y = fmaf(y, p.col_scale ? p.col_scale[n] : 1.f, p.col_bias ? p.col_bias[n] : 0.f);
      }
      if (SWIGLU) {
        // weight rows interleaved [gate block | up block] in 16-row blocks: column nn < 16 is a gate, nn + 16 (16 lanes up) its up value
        const float up = __shfl_down(y, 16, 64);
        if (nn < 16 && m < p.M && n + 16 < p.N) {
          const float o = silu(y) * up;
          const int ncol = tile * (NT * 4) + nn;
          if (p.out_f32) reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + ncol] = o;
          else reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + ncol] = from_f32<T>(o);
        }
      } else if (m < p.M && n < p.N) {
        if (p.out_f32) reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n] = y;
        else reinterpret_cast<T*>(p.C)[(int64_t)m * p.ldc + n] = from_f32<T>(y);
      }
    }
    parity ^= 1;
  }
}

template <typename T, int MT, bool NORM, int W = 4, bool SWIGLU = false>
int launch_persist_r16(const GemmArgs& a, hipStream_t st) {
  constexpr int kbe = 64 / (int)sizeof(T);
  const int nsteps = (a.K / kbe) / 2;
  const int ntiles = ceil_div(a.N, 32);
  static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  constexpr int smem = (2 * W * MT * 16 * 64 + 2 * W * MT * 16) * 4;
  constexpr int AS0 = W == 4 ? 8 : 4;                       // k steps per wave of the shallow variant; the deep one holds twice as many
  const int max_per_cu = (W == 4 && MT == 1 && nsteps <= W * AS0) ? 2 : 1;
  int per_cu = max_per_cu, best = INT32_MAX;
  for (int c = max_per_cu; c >= 1; --c) {
    const int load = c * ceil_div(ntiles, c * ncu);
    if (load < best) { best = load; per_cu = c; }
  }
  // shared chip (decode scheduler): half as many persistent workgroups
  int grid = ntiles < per_cu * ncu ? ntiles : per_cu * ncu;
  const int div = a.shared_chip ? 2 : 1;
  if (div > 1) grid = std::max(1, grid / div);
  auto kern = nsteps <= W * AS0 ? skinny_persist_r16_kernel<T, MT, NORM, AS0, W, SWIGLU> : skinny_persist_r16_kernel<T, MT, NORM, 2 * AS0, W, SWIGLU>;
  if (smem > 64 * 1024) {
    static bool attr_set[2] = {false, false};
    const int which = nsteps <= W * AS0 ? 0 : 1;
    if (!attr_set[which]) {
      OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      attr_set[which] = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(W * 64), smem, st, a, ntiles);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, bool NORM, int W = 4, bool SWIGLU = false>
int launch_persist(const GemmArgs& a, hipStream_t st) {
  constexpr int kbe = 64 / (int)sizeof(T);
  const int nsteps = (a.K / kbe) / 2;
  const int ntiles = ceil_div(a.N, 32);
  static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  constexpr int smem = (2 * W * 16 * 64 + W * 16) * 4;
  constexpr int AS0 = W == 4 ? 8 : 4;
  // Tiles are dealt statically (tile = bid + i * grid), so the launch ends with its most loaded CU: take the workgroups-per-CU count
  // (up to what the registers allow) for which a CU walks the fewest tiles -- lm_head, 4008 tiles on 256 CUs: 3 x 6 = 18 with three
  // workgroups per CU, 2 x 8 = 16 with two.
  const int max_per_cu = W == 8 ? 1 : (nsteps <= W * AS0 ? 3 : 2);
  int per_cu = max_per_cu, best = INT32_MAX;
  for (int c = max_per_cu; c >= 1; --c) {
    const int load = c * ceil_div(ntiles, c * ncu);
    if (load < best) { best = load; per_cu = c; }
  }
  const int grid = ntiles < per_cu * ncu ? ntiles : per_cu * ncu;
  auto kern = nsteps <= W * AS0 ? skinny_persist_kernel<T, NORM, AS0, W, SWIGLU> : skinny_persist_kernel<T, NORM, 2 * AS0, W, SWIGLU>;
  if (smem > 64 * 1024) {
    static bool attr_set[2] = {false, false};
    const int which = nsteps <= W * AS0 ? 0 : 1;
    if (!attr_set[which]) {
      OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      attr_set[which] = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(W * 64), smem, st, a, ntiles);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

// <= 8 rows, 8-wave form: gate / up (SwiGLU epilogue) with at least two tiles per CU (one persistent workgroup per CU walks them: 18.5 ->
// 17.7 us per launch, +2 % with three batches in flight, neutral alone).  OAVSR_SKINNY_PERSIST_GU=0: the per-tile kernel.
static bool persist_w8_eligible(const GemmArgs& a) {
  static const int on = knob("OAVSR_SKINNY_PERSIST"), gu = knob("OAVSR_SKINNY_PERSIST_GU");
  const int esz = a.dtype == DT_F32 ? 4 : 2, kbe = 64 / esz;
  static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  return on && gu && a.M <= 8 && a.K2 == 0 && a.swiglu_pairs && !a.out_f32 && !a.residual && a.act == ACT_NONE && a.N % 32 == 0 && a.N < 65536 &&
         ceil_div(a.N, 32) >= 2 * ncu && (a.K / kbe) % 2 == 0 && (a.K / kbe) / 2 <= 8 * 8;
}

// 9..32 rows, 8-wave form: the wide GEMVs below the lm_head class -- gate / up (SwiGLU epilogue) with at least two tiles per CU
static bool persist_r16_w8_eligible(const GemmArgs& a) {
  static const int on = knob("OAVSR_SKINNY_PERSIST");
  const int esz = a.dtype == DT_F32 ? 4 : 2, kbe = 64 / esz;
  static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  return on && a.M > 8 && a.M <= 32 && a.K2 == 0 && a.swiglu_pairs && !a.out_f32 && !a.residual && a.act == ACT_NONE && a.N % 32 == 0 &&
         a.N < 65536 && ceil_div(a.N, 32) >= 2 * ncu && (a.K / kbe) % 2 == 0 && (a.K / kbe) / 2 <= 8 * 8 &&
         !(a.M > 16 && (a.K / kbe) / 2 > 8 * 4);          // 32 rows x 8 steps of resident fragments: over the 256 registers of an 8-wave workgroup
}

static bool persist_eligible(const GemmArgs& a, int max_rows = 8) {
  static const int on = knob("OAVSR_SKINNY_PERSIST");
  const int esz = a.dtype == DT_F32 ? 4 : 2, kbe = 64 / esz;
  if (max_rows > 8 && (a.K / kbe) / 2 > 4 * 8 && a.M > 16) return false;     // 32 rows x 16 steps of resident fragments do not fit the register file
  return on && a.M <= max_rows && a.K2 == 0 && !a.swiglu_pairs && !a.residual && a.act == ACT_NONE && a.N >= 65536 && (a.K / kbe) % 2 == 0 &&
         (a.K / kbe) / 2 <= 4 * 16;
}

template <typename T, int MT, int NT, bool NORM, int U, bool HALF, int W>
__global__ __launch_bounds__(W * 64, (MT == 1 && U * NT <= 8) ? 4 : 2) void skinny_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  skinny_body<T, MT, NT, NORM, U, HALF, W, false>(p, blockIdx.x, smem, SkinnySync());
}

// Two dependent weight-streaming GEMMs of a decode step (<= 8 rows, 8-column tiles) in ONE launch: blocks [0, nb0) are the tiles of
// stage 0, the rest those of stage 1, which consume stage 0's output (LoRA down -> QKV; gate/up + SwiGLU -> down_proj).  A kernel
// boundary costs 4-5 us of launch / drain / first-byte latency on this chip -- as much as streaming the smaller projections takes --
// and inside one launch stage 1's weight stream starts while stage 0 is still computing.
template <typename T, bool NORM0, int NT0, int U0, bool NORM1, int NT1, int U1>
__global__ __launch_bounds__(kMaxWaves * 64, 4) void skinny_chain2_kernel(const GemmArgs g0, const GemmArgs g1, const int nb0, const SkinnySync s0,
                                                                           const SkinnySync s1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < nb0) skinny_body<T, 1, NT0, NORM0, U0, true, kMaxWaves, true>(g0, blockIdx.x, smem, s0);
  else skinny_body<T, 1, NT1, NORM1, U1, true, kMaxWaves, true>(g1, (int)blockIdx.x - nb0, smem, s1);
}

template <int MT, int NT, bool NORM, int W>
constexpr int sk_smem0() { return W * MT * 16 * NT * 16 * 4 * (NORM ? 2 : 1) + (NORM ? W * MT * 16 * 4 : 0); }

template <typename T, bool NORM0, int NT0, int U0, bool NORM1, int NT1, int U1>
int launch_chain2(const GemmArgs& a0, const GemmArgs& a1, const SkinnyChain& c, hipStream_t st) {
  constexpr int m0 = sk_smem0<1, NT0, NORM0, kMaxWaves>(), m1 = sk_smem0<1, NT1, NORM1, kMaxWaves>();
  const int smem = std::max(m0 + (NORM0 && m0 <= 96 * 1024 ? a0.K * 4 : 0), m1 + (NORM1 && m1 <= 96 * 1024 ? a1.K * 4 : 0));
  OAVSR_REQUIRE(smem <= 160 * 1024, "skinny chain: K too deep for the fused RMSNorm prologue");
  auto kern = skinny_chain2_kernel<T, NORM0, NT0, U0, NORM1, NT1, U1>;
  static bool attr_set = false;
  if (!attr_set) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int nb0 = ceil_div(a0.N, NT0 * 8), nb1 = ceil_div(a1.N, NT1 * 8);
  SkinnySync s0, s1;
  s0.signal_ctr = c.counter; s0.signal_n = nb0; s0.step = c.step; s0.err = c.err;
  s1.wait_ctr = c.counter; s1.wait_n = nb0 < 8 ? nb0 : 8; s1.step = c.step; s1.err = c.err; s1.a1_sc1 = c.feeds_a2 ? 0 : 1; s1.a2_sc1 = c.feeds_a2 ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(nb0 + nb1), dim3(kMaxWaves * 64), smem, st, a0, a1, nb0, s0, s1);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int MT, int NT, bool NORM, int U, bool HALF = false, int W = kMaxWaves>
int launch_sk_u(const GemmArgs& a, hipStream_t st) {
  constexpr int kWaves = W, kThreads = W * 64;
  constexpr int smem0 = kWaves * MT * 16 * NT * 16 * 4 * (NORM ? 2 : 1) + (NORM ? kWaves * MT * 16 * 4 : 0);
  static_assert(smem0 <= 160 * 1024, "LDS");
  constexpr bool gain_lds = NORM && W == kMaxWaves && smem0 <= 96 * 1024;     // as skinny_body's GAIN_LDS
  // the partial sums of the un-normalised second K segment get their own LDS image only when there is one
  const int smem = smem0 - ((NORM && !a.K2) ? kWaves * MT * 16 * NT * 16 * 4 : 0) + (gain_lds ? a.K * 4 : 0);   // + the staged norm gain
  OAVSR_REQUIRE(smem <= 160 * 1024, "skinny gemm: K=%d too deep for the fused RMSNorm prologue", a.K);
  auto kern = skinny_kernel<T, MT, NT, NORM, U, HALF, W>;
  static bool attr_set = false;
  if (!attr_set) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(ceil_div(a.N, NT * (HALF ? 8 : 16))), dim3(kThreads), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

static bool half_enabled() { return true; }      // (8-column HALF-mode tiles for <= 8 rows: always on since round 1)

template <typename T, int MT, int NT, bool NORM>
int launch_sk(const GemmArgs& a, hipStream_t st) {
  constexpr int U0 = MT >= 8 ? 2 : MT >= 4 ? 4 : 8 / NT;                // weight loads in flight per wave: U0 * NT KiB; sized for <= 128 VGPRs (2 workgroups per CU)
  constexpr int kbe = 64 / (int)sizeof(T);
  if constexpr (MT == 1 && (NT == 1 || NT == 4)) {
    // B <= 8 rows: 8-column MFMA tiles (HALF mode).  NT = 1: doubles the grid where 16-column tiles leave CUs idle;
    // NT = 4 (gate/up, lm_head): same 256+ workgroups x 32 columns, but every wave load is 8 rows x one whole 128-B line.
    const bool want = NT == 1 ? (a.N >= 64 && a.N / 16 < 384) : true;
    if (half_enabled() && want && a.M <= 8 && (!a.swiglu_pairs || a.N % 32 == 0) && (a.K / kbe) % 2 == 0) {   // ragged N: loads and stores are guarded
      // many rounds of wide tiles (lm_head): 4-wave workgroups overlap their reduce / epilogue phases better (110 -> 101 us)
      // (deeper request queues per wave -- U = 4 / 8, the whole k range of a wave in one go -- were measured: 18.4 -> 26.3 us gate/up,
      //  109 -> 148 us lm_head; the 200+ VGPRs they need leave one workgroup per CU)
      if (NT == 4 && persist_eligible(a)) return launch_persist<T, NORM>(a, st);     // same bits, persistent workgroups
      if (NT == 4 && persist_w8_eligible(a)) return launch_persist<T, NORM, 8, true>(a, st);
      if (NT == 4 && a.N >= 65536) return launch_sk_u<T, 1, NT, NORM, U0, true, 4>(a, st);
      return launch_sk_u<T, 1, NT, NORM, U0, true>(a, st);
    }
  }
  // deep K with few column tiles (down_proj): keep twice as many weight loads in flight per wave
  if constexpr (MT == 1 && NT == 1) {
    if ((a.K + a.K2) * (int)sizeof(T) >= 64 * kMaxWaves * 16) return launch_sk_u<T, MT, NT, NORM, 16>(a, st);
  }
  return launch_sk_u<T, MT, NT, NORM, U0>(a, st);
}

template <typename T, bool NORM>
int launch_sk_t(const GemmArgs& a, hipStream_t st) {
  const int mt = ceil_div(a.M, 16);
  // 32-column workgroups once they fill the chip; below that 8-column ones: they re-read the activation rows four times as often per
  // weight byte, but measured with 32-column tiles for N = 2048 / 3072 the decode step is 25 % slower alone and 1.5 % slower with three
  // batches in flight (64 - 96 workgroups per launch: the chain of dependent launches gets longer, which costs more than the CU time saved)
  const bool wide = a.N >= 8192;
  // 9..32 rows: 8-column tiles with 16-row activation operands (skinny_r16_kernel) wherever the <= 8-row path would use its 8-column tiles
  constexpr int r16_on = 1;
  constexpr int kbe = 64 / (int)sizeof(T);
  if (r16_on && half_enabled() && a.M > 32 && mt <= 4 && (a.K / kbe) % 2 == 0) {
    // 33..64 rows (the decode scheduler's 64-row steps): the same 8-column-tile arithmetic with four 16-row activation operands per
    // weight fragment -- a row's bits are those of the <= 8-row kernel (tested), only the weight stream is shared by twice the rows.
    // Tile widths follow the LDS image (waves x 64 rows x columns x 4 B <= 128 KiB per image) and the accumulator registers.
    if (wide && (!a.swiglu_pairs || a.N % 32 == 0)) {
      if (a.K2 == 0) {
        if (a.N >= 65536) return launch_r16<T, 4, 4, NORM, 1, 4, false>(a, st);          // lm_head class: 4 waves, like every other row count
        return launch_r16<T, 4, 4, NORM, 1, kMaxWaves, false>(a, st);                    // gate / up (+ SwiGLU): 32 columns per workgroup (two k steps in flight per wave do not fit the register file with the fused norm)
      }
      return a.N >= 65536 ? launch_r16<T, 4, 2, NORM, 1, 4>(a, st) : launch_r16<T, 4, 2, NORM, 1>(a, st);      // (a wide GEMV with a LoRA segment: no call site today)
    }
    if (!a.swiglu_pairs && a.N >= 64 && a.N / 16 < 384) {
      if constexpr (!NORM) { if (a.N >= 1024) return launch_r16<T, 4, 4, NORM, 2>(a, st); }        // o_proj, down_proj (no norm: one accumulator set, room for two k steps of weights in flight per wave)
      if (a.K2 == 0) return launch_r16<T, 4, 2, NORM, 2, kMaxWaves, false>(a, st);       // LoRA down (no second segment: one accumulator set, two k steps of weights in flight)
      return launch_r16<T, 4, 2, NORM, 1>(a, st);                                        // QKV + LoRA segment: two LDS images, two accumulator sets
    }
  }
  if (r16_on && half_enabled() && a.M > 8 && mt <= 2 && (a.K / kbe) % 2 == 0) {
    if (wide && (!a.swiglu_pairs || a.N % 32 == 0)) {
      // same wave count as the <= 8-row kernel picks for this N (4 waves for the many-round lm_head): the k blocks are dealt over
      // the waves, so the wave count is part of the summation order
      if (a.N >= 65536 && persist_eligible(a, 32)) return mt <= 1 ? launch_persist_r16<T, 1, NORM>(a, st) : launch_persist_r16<T, 2, NORM>(a, st);
      if (a.N >= 65536) return mt <= 1 ? launch_r16<T, 1, 4, NORM, 2, 4>(a, st) : launch_r16<T, 2, 4, NORM, 2, 4>(a, st);
      if (persist_r16_w8_eligible(a)) return mt <= 1 ? launch_persist_r16<T, 1, NORM, 8, true>(a, st) : launch_persist_r16<T, 2, NORM, 8, true>(a, st);
      return mt <= 1 ? launch_r16<T, 1, 4, NORM, 2>(a, st) : launch_r16<T, 2, 4, NORM, 2>(a, st);
    }
    // narrow N: two 8-column tiles per workgroup (N / 16 workgroups, as many as the 16-column kernel launched, and the activation rows --
    // 2 - 4 x the weight bytes of a tile at these row counts -- are fetched once per 16 columns); each tile sums exactly as above
    if (!a.swiglu_pairs && a.N >= 64 && a.N / 16 < 384) {
      // shared chip (decode scheduler), 17..32 rows: eight (o_proj, down_proj) / four (QKV, with the fused norm's second LDS image) 8-column
      // tiles per workgroup instead of two.
      const int want = a.shared_chip ? 8 : 2;
      if constexpr (!NORM) { if (want >= 8 && mt == 2 && a.N >= 1024) return launch_r16<T, 2, 8, NORM, 1>(a, st); }
      if (want >= 4 && mt == 2 && a.N >= (NORM ? 2048 : 512)) return launch_r16<T, 2, 4, NORM, 2>(a, st);      // (the LoRA down projection, N <= 512, keeps its 16 - 32 workgroups)
      return mt <= 1 ? launch_r16<T, 1, 2, NORM, 4>(a, st) : launch_r16<T, 2, 2, NORM, 2>(a, st);
    }
  }
  if (a.swiglu_pairs) {
    OAVSR_REQUIRE(mt <= 4 && a.N % 32 == 0, "skinny gemm: fused SwiGLU needs M <= 64 and N %% 32 == 0");
    if (mt <= 1) return wide ? launch_sk<T, 1, 4, NORM>(a, st) : launch_sk<T, 1, 2, NORM>(a, st);
    if (mt <= 2) return wide ? launch_sk<T, 2, 4, NORM>(a, st) : launch_sk<T, 2, 2, NORM>(a, st);
    return launch_sk<T, 4, 2, NORM>(a, st);       // 33..64 rows below the wide class (small models): the 16-column kernel, as for <= 32 rows
  }
  if (mt <= 1) return wide ? launch_sk<T, 1, 4, NORM>(a, st) : launch_sk<T, 1, 1, NORM>(a, st);
  if (mt <= 2) return wide ? launch_sk<T, 2, 4, NORM>(a, st) : launch_sk<T, 2, 1, NORM>(a, st);
  if (mt <= 4) return launch_sk<T, 4, 1, NORM>(a, st);
  return launch_sk<T, 8, 1, NORM>(a, st);
}

}  // namespace

// HALF (8-column tile) eligibility of one stage, as launch_sk decides it for M <= 8
static bool chain_stage_ok(const GemmArgs& a, int nt) {
  const int esz = a.dtype == DT_F32 ? 4 : 2, kbe = 64 / esz, vec = 16 / esz;
  if (a.M < 1 || a.M > 8 || a.amode != A_DENSE || a.groups != 1 || a.K <= 0 || a.K % kbe || a.K2 % kbe || (a.K / kbe) % 2) return false;
  if (a.lda % vec || a.ldw % vec || (((uintptr_t)a.A | (uintptr_t)a.W) & 15)) return false;
  if (a.K2 && (a.lda2 % vec || a.ldw2 % vec || (((uintptr_t)a.A2 | (uintptr_t)a.W2) & 15))) return false;
  if (a.N % 2 || a.ldc % 2 || ((uintptr_t)a.C & 3)) return false;                      // 4-byte write-through stores
  if (nt == 1) return !a.swiglu_pairs && a.N >= 64 && a.N / 16 < 384 && a.N < 8192;   // launch_sk's NT = 1 HALF rule
  return a.swiglu_pairs && a.N >= 8192 && a.N < 65536 && a.N % 32 == 0;               // NT = 4: gate / up with the SwiGLU epilogue
}

bool skinny_chain_supported(const GemmArgs& a0, const GemmArgs& a1, int kind) {
  // Off by default.  Measured on MI355X (round 2, Llama-3.2-1B, 8 rows): the in-launch seam -- sc1 stores, drain, two-level fan-in, poll,
  // sc1 activation loads -- costs what the kernel boundary it replaces costs (4-5 us): LoRA down -> QKV +1 % sequential / -2 % with three
  // batches in flight, gate/up -> down -3 % / -5 % (its 256 consumers can only start when the 512 producers have left the CUs).
  // OAVSR_SKINNY_CHAIN: 0 off, 1 both fusions, 2 / 3 only LoRA->QKV / gate-up->down.
  static const int on = knob("OAVSR_SKINNY_CHAIN");
  if (!on || (on == 2 && kind != 0) || (on == 3 && kind != 1) || !half_enabled() || a0.dtype != a1.dtype) return false;
  if (kind == 0) return a0.norm_gain && a1.norm_gain && a1.K2 > 0 && chain_stage_ok(a0, 1) && chain_stage_ok(a1, 1);   // LoRA down -> QKV
  return a0.norm_gain && !a1.norm_gain && chain_stage_ok(a0, 4) && chain_stage_ok(a1, 1);                               // gate/up -> down
}

int launch_gemm_skinny_chain2(const GemmArgs& a0, const GemmArgs& a1, const SkinnyChain& c, int kind, hipStream_t st) {
  OAVSR_REQUIRE(skinny_chain_supported(a0, a1, kind) && c.counter && c.step, "skinny chain: unsupported shapes");
  if (kind == 0) return a0.dtype == DT_F32 ? launch_chain2<float, true, 1, 8, true, 1, 8>(a0, a1, c, st) : launch_chain2<bf16_t, true, 1, 8, true, 1, 8>(a0, a1, c, st);
  return a0.dtype == DT_F32 ? launch_chain2<float, true, 4, 2, false, 1, 8>(a0, a1, c, st) : launch_chain2<bf16_t, true, 4, 2, false, 1, 8>(a0, a1, c, st);
}

// which weight-streaming kernel launch_gemm_skinny picks for 9..32 rows (profiling family names follow the kernel that runs)
bool skinny_uses_r16(const GemmArgs& a) {
  constexpr int r16_on = 1;
  const int esz = a.dtype == DT_F32 ? 4 : 2, kbe = 64 / esz, mt = ceil_div(a.M, 16);
  if (!(r16_on && half_enabled() && a.M > 8 && mt <= 4 && (a.K / kbe) % 2 == 0)) return false;
  if (a.N >= 8192 && (!a.swiglu_pairs || a.N % 32 == 0)) return true;
  return !a.swiglu_pairs && a.N >= 64 && a.N / 16 < 384;
}

int launch_gemm_skinny(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return 0;
  const int esz = a.dtype == DT_F32 ? 4 : 2;
  const int kbe = 64 / esz, vec = 16 / esz;
  OAVSR_REQUIRE(a.M <= 128, "skinny gemm: M=%d > 128", a.M);
  OAVSR_REQUIRE(a.amode == A_DENSE && a.groups == 1, "skinny gemm: dense only");
  OAVSR_REQUIRE(a.K > 0 && a.K % kbe == 0 && a.K2 % kbe == 0, "skinny gemm: K=%d/K2=%d must be multiples of %d", a.K, a.K2, kbe);
  OAVSR_REQUIRE(a.lda % vec == 0 && a.ldw % vec == 0 && (((uintptr_t)a.A | (uintptr_t)a.W) & 15) == 0, "skinny gemm: operands must be 16-byte aligned");
  if (a.K2) OAVSR_REQUIRE(a.lda2 % vec == 0 && a.ldw2 % vec == 0 && (((uintptr_t)a.A2 | (uintptr_t)a.W2) & 15) == 0, "skinny gemm: segment-2 operands must be 16-byte aligned");
  OAVSR_REQUIRE(!a.swiglu_pairs || (!a.residual && a.act == ACT_NONE), "skinny gemm: SwiGLU epilogue excludes residual/activation");
  if (a.norm_gain) {
    OAVSR_REQUIRE(((uintptr_t)a.norm_gain & 15) == 0, "skinny gemm: norm gain must be 16-byte aligned");
    return a.dtype == DT_F32 ? launch_sk_t<float, true>(a, st) : launch_sk_t<bf16_t, true>(a, st);
  }
  return a.dtype == DT_F32 ? launch_sk_t<float, false>(a, st) : launch_sk_t<bf16_t, false>(a, st);
}

}  // namespace oavsr
