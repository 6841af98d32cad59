// Attention kernels for gfx950.
//
// (1) attn_varlen_kernel: flash-style attention over variable-length sequences (cu_seqlens), used for the
//     Whisper / AV-HuBERT encoders (bidirectional, multihead_attention.py:511-654) and the LLM prefill
//     (causal GQA, Llama_LoRA.py:284-307).  No padded rows are ever attended: each utterance only sees its
//     own keys, which is what makes a batch equal the reference's batch-1 runs.
//     Structure: workgroup = 4 waves = 64 query rows of one (sequence, head); K/V tiles of 64 keys staged in
//     LDS (16-B chunks, XOR-swizzled).  Each wave computes S^T = K.Q^T with the 16x16 MFMA so that a lane owns
//     one query column (lane & 15) -> softmax statistics are lane-local + two xor-shuffles, and the S^T
//     accumulator registers ARE the B operand of the second product O^T = V^T.P^T (no LDS round trip for P).
//     bf16: V^T fragments come from row-major V in LDS through ds_read_b64_tr_b16 (hardware transpose).
//     f32 : exact-f32 MFMA 16x16x4, V read with ds_read_b32.
// (2) decode_attn_kernel: one new token per row against a KV cache (shared utterance prefix + per-row tail
//     addressed through a beam ancestry table).  HBM-bound, VALU only.
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kThreads = 256;
constexpr int KT = 64;   // keys per LDS tile

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// 8-byte transposing LDS read with an immediate offset; the caller orders it with an explicit s_waitcnt lgkmcnt
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr16_b64(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

template <typename T> struct MmaT;
template <> struct MmaT<bf16_t> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
  }
};
template <> struct MmaT<float> {
  static __device__ __forceinline__ void run(const Chunk16& a, const Chunk16& b, f32x4& c) {
    f32x4 fa = __builtin_bit_cast(f32x4, a.v), fb = __builtin_bit_cast(f32x4, b.v);
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s], fb[s], c, 0, 0, 0);
  }
};

template <typename T, int HD>
__global__ __launch_bounds__(kThreads) void attn_varlen_kernel(const AttnArgs p) {
  constexpr int RB = HD * (int)sizeof(T);        // bytes per K/V row
  constexpr int NCH = RB / 16;                    // 16-B chunks per row
  constexpr int SWZ = (RB == 128) ? 7 : 15;
  constexpr int NBLK = RB / 64;                   // 64-byte k blocks along head_dim
  constexpr int NDT = HD / 16;                    // 16-wide output d tiles
  constexpr int NLD = KT * NCH / kThreads;        // chunks per thread per tile per tensor
  __shared__ __attribute__((aligned(16))) char sK[KT * RB];
  __shared__ __attribute__((aligned(16))) char sV[KT * RB];

  const int seq = blockIdx.z, head = blockIdx.y;
  const int q0 = p.cu_q[seq], qlen = p.cu_q[seq + 1] - q0;
  const int k0 = p.cu_k[seq], klen = p.cu_k[seq + 1] - k0;
  const int qt0 = blockIdx.x * 64;
  if (qt0 >= qlen) return;
  const int kvh = head / (p.n_heads / p.n_kv_heads);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;
  const int qrow = qt0 + wave * 16 + qi;                 // query index inside the sequence (may be >= qlen)
  const int qrow_c = qrow < qlen ? qrow : qlen - 1;
  const T* Q = reinterpret_cast<const T*>(p.Q);
  const T* K = reinterpret_cast<const T*>(p.K);
  const T* V = reinterpret_cast<const T*>(p.V);

  Chunk16 qf[NBLK];
  {
    const char* qp = reinterpret_cast<const char*>(Q + (int64_t)(q0 + qrow_c) * p.ldq + (int64_t)head * HD);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) qf[b] = *reinterpret_cast<const Chunk16*>(qp + b * 64 + g * 16);
  }
  const int shift = klen - qlen;                         // causal: key j visible iff j <= qrow + shift
  int kend = klen;
  if (p.causal) {
    int last = qt0 + 63 < qlen ? qt0 + 63 : qlen - 1;
    kend = last + shift + 1;
    if (kend > klen) kend = klen;
    if (kend < 0) kend = 0;
  }
  const float sc = p.scale * 1.44269504088896340736f;    // softmax in base 2

  f32x4 o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (int kt0 = 0; kt0 < kend; kt0 += KT) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      int id = tid + j * kThreads, row = id / NCH, c = id - row * NCH;
      int key = kt0 + row;
      Chunk16 ck, cv;
      if (key < klen) {
        ck = *reinterpret_cast<const Chunk16*>(reinterpret_cast<const char*>(K + (int64_t)(k0 + key) * p.ldk + (int64_t)kvh * HD) + c * 16);
        cv = *reinterpret_cast<const Chunk16*>(reinterpret_cast<const char*>(V + (int64_t)(k0 + key) * p.ldv + (int64_t)kvh * HD) + c * 16);
      } else {
        ck.v = u32x4{0u, 0u, 0u, 0u}; cv.v = ck.v;
      }
      int off = row * RB + ((c ^ (row & SWZ)) << 4);
      *reinterpret_cast<Chunk16*>(sK + off) = ck;
      *reinterpret_cast<Chunk16*>(sV + off) = cv;
    }
    __syncthreads();

    // ---- S^T[key][q] for 4 sub-tiles of 16 keys ----
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = kt * 16 + qi;                       // A-operand row = key (lane & 15)
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        Chunk16 kf = *reinterpret_cast<const Chunk16*>(sK + row * RB + (((b * 4 + g) ^ (row & SWZ)) << 4));
        MmaT<T>::run(kf, qf[b], s[kt]);
      }
    }
    // ---- mask + online softmax (lane owns query qi; its keys: kt*16 + 4g + r) ----
    float mt = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = kt0 + kt * 16 + 4 * g + r;
        bool vis = key < klen && (!p.causal || key <= qrow + shift);
        float v = vis ? s[kt][r] * sc : -INFINITY;
        s[kt][r] = v;
        mt = fmaxf(mt, v);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_use);
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = exp2f(s[kt][r] - m_use);
        s[kt][r] = e;
        lsum += e;
      }
    l_run = l_run * alpha + lsum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[i] *= alpha;

    // ---- O^T[d][q] += V^T[d][key] . P^T[key][q] ----
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {                       // 32 keys per MFMA: sub-tiles 2u, 2u+1
        bf16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[r] = (bf16_t)s[2 * u][r]; pf[4 + r] = (bf16_t)s[2 * u + 1][r]; }
        const int qq = qi >> 2, pp = qi & 3;             // tr-read: lane 4q+p addresses row q, cols 4p..4p+3
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const int c = dt * 2 + (pp >> 1);
          const int row0 = (2 * u) * 16 + 4 * g + qq, row1 = row0 + 16;
          auto a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(sV + row0 * RB + ((c ^ (row0 & SWZ)) << 4) + (pp & 1) * 8));
          auto a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(sV + row1 * RB + ((c ^ (row1 & SWZ)) << 4) + (pp & 1) * 8));
          bf16x4 v0 = __builtin_bit_cast(bf16x4, a0), v1 = __builtin_bit_cast(bf16x4, a1);
          bf16x8 vf;
#pragma unroll
          for (int r = 0; r < 4; ++r) { vf[r] = v0[r]; vf[4 + r] = v1[r]; }
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const int c = dt * 4 + (qi >> 2);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = kt * 16 + 4 * g + r;
            float vv = *reinterpret_cast<const float*>(sV + row * RB + ((c ^ (row & SWZ)) << 4) + (qi & 3) * 4);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, s[kt][r], o[dt], 0, 0, 0);
          }
        }
    }
  }
  // ---- finalize: l over the 4 lane groups, write O[q][d = dt*16 + 4g + r] ----
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (qrow < qlen) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    T* O = reinterpret_cast<T*>(p.O) + (int64_t)(q0 + qrow) * p.ldo + (int64_t)head * HD;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      if constexpr (sizeof(T) == 2) {
        bf16x4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (bf16_t)(o[dt][r] * inv);
        *reinterpret_cast<bf16x4*>(O + dt * 16 + 4 * g) = w;
      } else {
        f32x4 w = o[dt] * inv;
        *reinterpret_cast<f32x4*>(O + dt * 16 + 4 * g) = w;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// (1b) attn_dma_kernel: the bf16 / head_dim 64 fast path of (1).  Same math and fragment layouts; what changes is the data
//      movement and the instruction count of the soft-max, which is what bounds this kernel (VALU issue, not MFMA):
//      * K/V tiles (64 keys x 128 B each) go HBM -> LDS by LDS-DMA (buffer_load ... lds) through a 3-stage ring, two tiles ahead of
//        their use, with ONE barrier per tile (counted vmcnt, never a drain); addressing is a constant VGPR offset per lane plus a
//        scalar tile offset, rows past the sequence end are zero-filled by the buffer range check;
//      * a wave owns QT query tiles of 16 rows so every K / V^T fragment read from LDS feeds QT MFMAs;
//      * two loop bodies: the leading, fully visible tiles run one with no masking code at all, the edge tiles the masked one;
//      * row max across the lane groups with v_permlane16/32_swap (no LDS round trip); exp2 is the bare v_exp_f32 with the softmax
//        scale folded into one fma; plain (unpacked) fp32 math.
// max over lanes l, l^16, l^32, l^48 with the gfx950 row / half swaps: v_permlane16_swap exchanges rows 1 <-> 0' and 3 <-> 2' of the two
// operands (rows = 16 lanes), v_permlane32_swap the upper half of one with the lower half of the other; with both operands holding x,
// max(a, b) afterwards is the xor-16 (xor-32) butterfly step.  The hardware needs 2 wait states between a VALU write and the swap.
__device__ __forceinline__ float xor_max_16_32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\ts_nop 1\n\t"
               "v_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1"
               : "+v"(a), "+v"(b));
  return a;
}

template <int QT>
__global__ __launch_bounds__(kThreads, 2) void attn_dma_kernel(const AttnArgs p) {
  constexpr int HD = 64, RB = 128, NDT = 4, NSTAGE = 3;
  constexpr int kStage = 2 * KT * RB;                  // K image then V image
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int seq = blockIdx.z, head = blockIdx.y;
  const int q0 = p.cu_q[seq], qlen = p.cu_q[seq + 1] - q0;
  const int k0 = p.cu_k[seq], klen = p.cu_k[seq + 1] - k0;
  const int qt0 = blockIdx.x * (64 * QT);
  if (qt0 >= qlen) return;
  const int kvh = head / (p.n_heads / p.n_kv_heads);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 15, g = lane >> 4;
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.Q);
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.K) + (int64_t)kvh * HD;
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.V) + (int64_t)kvh * HD;

  const int shift = klen - qlen;                         // causal: key j visible iff j <= qrow + shift
  int kend = klen;
  if (p.causal) {
    int last = qt0 + 64 * QT - 1 < qlen ? qt0 + 64 * QT - 1 : qlen - 1;
    kend = last + shift + 1;
    if (kend > klen) kend = klen;
    if (kend < 0) kend = 0;
  }
  const int ntile = (kend + KT - 1) / KT;

  // DMA: wave w fills 1-KiB segments w and w+4 (8 key rows each) of the K and of the V image; lane i lands at physical
  // chunk i&7 of row seg*8 + (i>>3) and therefore fetches logical chunk (i&7) ^ (i>>3)   [chunk ^ (row & 7) swizzle].
  // Addressing is all scalar per tile: K and V of this (sequence, kv head) are buffer resources whose num_records ends with the last valid
  // key row, a lane's byte offset inside the first tile is a constant VGPR, and the tile advance is an SGPR offset (buffer_load ... lds
  // with soffset) -- no per-tile vector address arithmetic, and rows past klen read as zeros by the hardware range check.
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K + (int64_t)k0 * p.ldk), 0,
                                                                       klen > 0 ? (int)(((int64_t)klen - 1) * p.ldk * 2 + RB) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(V + (int64_t)k0 * p.ldv), 0,
                                                                       klen > 0 ? (int)(((int64_t)klen - 1) * p.ldv * 2 + RB) : 0, 0x00020000);
  int kvo[2], vvo[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int key = (wave + 4 * h) * 8 + lrow;
    kvo[h] = (key * (int)p.ldk + lchunk * 8) * 2;
    vvo[h] = (key * (int)p.ldv + lchunk * 8) * 2;
  }
  const int kstep = KT * (int)p.ldk * 2, vstep = KT * (int)p.ldv * 2;
  int ksoff = 0, vsoff = 0;                              // scalar: byte offset of the next tile to issue
  unsigned dslot = 0;                                    // LDS byte offset of the ring slot the next issue() fills
  auto issue = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      char* dst = smem + (dslot + (wave + 4 * h) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)dst, 16, kvo[h], ksoff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)(dst + KT * RB), 16, vvo[h], vsoff, 0, 0);
    }
    ksoff += kstep; vsoff += vstep;
    dslot = dslot == (NSTAGE - 1) * kStage ? 0u : dslot + kStage;
  };

  // Q fragments are fetched between the first two tile DMAs and then pinned behind an explicit counted wait: the compiler
  // must not see them as "maybe still in flight" inside the loop, or it drains vmcnt -- and with it the K/V prefetch -- at
  // their first use in every iteration.
  if (ntile > 0) issue();
  Chunk16 qf[QT][2];
  int qrow[QT];
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    qrow[j] = qt0 + (wave * QT + j) * 16 + qi;           // query index inside the sequence (may be >= qlen)
    const int qc = qrow[j] < qlen ? qrow[j] : qlen - 1;
    const char* qp = reinterpret_cast<const char*>(Q + (int64_t)(q0 + qc) * p.ldq + (int64_t)head * HD);
#pragma unroll
    for (int b = 0; b < 2; ++b) qf[j][b] = *reinterpret_cast<const Chunk16*>(qp + b * 64 + g * 16);
  }
  if (ntile > 1) {
    issue();
    if constexpr (QT == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(qf[0][0].v), "+v"(qf[0][1].v), "+v"(qf[1][0].v), "+v"(qf[1][1].v)::"memory");
    else asm volatile("s_waitcnt vmcnt(4)" : "+v"(qf[0][0].v), "+v"(qf[0][1].v)::"memory");
  } else {
    if constexpr (QT == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0][0].v), "+v"(qf[0][1].v), "+v"(qf[1][0].v), "+v"(qf[1][1].v)::"memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0][0].v), "+v"(qf[0][1].v)::"memory");
  }

  // LDS byte offsets of this lane's fragments inside a stage.  K (A operand): row kt*16 + qi, 16-B chunk (b*4 + g) ^ (qi & 7);
  // V^T (tr-read: lane 4q+p addresses row q, cols 4p..4p+3): row 32u + 4g + qq (+16), chunk (2dt + (pp>>1)) ^ (row & 7).
  // kt / u / +16 rows never touch row & 7, so they are plain immediates on top of 2 + 4 per-lane bases.
  unsigned kbase[2], vbase[NDT];
  {
    const int qq = qi >> 2, pp = qi & 3, vrow = 4 * g + qq;
#pragma unroll
    for (int b = 0; b < 2; ++b) kbase[b] = qi * RB + (((b * 4 + g) ^ (qi & 7)) << 4);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) vbase[dt] = vrow * RB + (((dt * 2 + (pp >> 1)) ^ (vrow & 7)) << 4) + (pp & 1) * 8;
  }

  const float sc = p.scale * 1.44269504088896340736f;    // softmax in base 2
  f32x4 o[QT][NDT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    m_run[j] = -INFINITY; l_run[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  unsigned cslot = 0;                                    // ring slot of the tile being consumed
  // One key tile.  MASKED is a compile-time property of the call site: the leading tiles of a wave are fully visible to all its queries and
  // run a body WITHOUT any masking code -- as a run-time branch inside one body the compiler if-converted the visibility selects into
  // the common path (~150 extra VALU instructions per tile, as many as the softmax itself).
  auto tile = [&](const int t, auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    // tile t has landed once only tile t+1's 4 DMA instructions of this wave are still outstanding
    if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < ntile) issue();                          // tile t+2; its stage held tile t-1: every wave is past it
    const unsigned so = cslot;
    cslot = cslot == (NSTAGE - 1) * kStage ? 0u : cslot + kStage;
    unsigned kaddr[2], vaddr[NDT];                        // lane bases inside this stage; everything else is an immediate offset
#pragma unroll
    for (int b = 0; b < 2; ++b) kaddr[b] = so + kbase[b];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) vaddr[dt] = so + KT * RB + vbase[dt];
    const int kt0 = t * KT;

    // ---- S^T[key][q] for 4 sub-tiles of 16 keys; a K fragment feeds all QT query tiles ----
    f32x4 s[QT][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      Chunk16 kf[2];                                      // A-operand row = key kt*16 + (lane & 15)
#pragma unroll
      for (int b = 0; b < 2; ++b) kf[b] = *reinterpret_cast<const Chunk16*>(smem + (kaddr[b] + kt * 16 * RB));
#pragma unroll
      for (int j = 0; j < QT; ++j) {
        s[j][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 2; ++b) MmaT<bf16_t>::run(kf[b], qf[j][b], s[j][kt]);
      }
    }
    // ---- online softmax (lane owns query qi of each tile; its keys: kt*16 + 4g + r) ----
    bf16x8 pf[QT][2];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      const int qmin = qt0 + (wave * QT + j) * 16;
      const bool need_mask = MASKED && (kt0 + KT > klen || (p.causal && kt0 + KT - 1 > qmin + shift));     // wave-uniform
      if constexpr (MASKED) if (need_mask) {
        int key0 = kt0 + 4 * g;
        asm volatile("" : "+v"(key0));                    // pins the index / compare chain inside this rarely taken branch
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + kt * 16 + r;
            const bool vis = key < klen && (!p.causal || key <= qrow[j] + shift);
            if (!vis) s[j][kt][r] = -INFINITY;
          }
      }
      float mt = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mt = fmaxf(mt, s[j][kt][r]);
      mt = xor_max_16_32(mt);                             // over the 4 lane groups, VALU only (no LDS round trip as with ds_bpermute)
      mt *= sc;                                           // sc > 0: max(s) * sc == max(s * sc)
      const float m_new = fmaxf(m_run[j], mt);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_use);
      // exponent argument and row sum in plain (unpacked) fp32: beside MFMAs a v_pk_fma_f32 / v_pk_add_f32 costs ~20 cycles more than the two
      // scalar instructions it replaces (MI355X_MICROARCH.md, price of one filler beside MFMAs); the file is built with -fno-slp-vectorize
      // so that the compiler does not re-pack them.  Two partial sums keep the add chains short.
      float lsum2[2] = {0.f, 0.f};
      const float nm = -m_use;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(s[j][kt][r], sc, nm));
          s[j][kt][r] = e;
          lsum2[r & 1] += e;
        }
      const float lsum = lsum2[0] + lsum2[1];
      l_run[j] = l_run[j] * alpha + lsum;
      m_run[j] = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {   // the running max moved for some query of this tile (rare after the first tiles)
#pragma unroll
        for (int i = 0; i < NDT; ++i) o[j][i] *= alpha;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[j][u][r] = (bf16_t)s[j][2 * u][r]; pf[j][u][4 + r] = (bf16_t)s[j][2 * u + 1][r]; }
    }
    // ---- O^T[d][q] += V^T[d][key] . P^T[key][q]; a V^T fragment feeds all QT query tiles ----
    // (ds_read_b64_tr_b16 issued as inline asm: the builtin never folds the sub-tile offsets into the instruction's offset field
    //  and would spend 16 address adds per tile; the loads are tied to the s_waitcnt below through its operands)
    u32x2 vt[2 * NDT][2];
    static_for<0, 2 * NDT>([&](auto I) {
      constexpr int u = decltype(I)::value / NDT, dt = decltype(I)::value % NDT;
      vt[u * NDT + dt][0] = lds_tr16_b64<u * 32 * RB>(vaddr[dt]);
      vt[u * NDT + dt][1] = lds_tr16_b64<u * 32 * RB + 16 * RB>(vaddr[dt]);
    });
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vt[0][0]), "+v"(vt[0][1]), "+v"(vt[1][0]), "+v"(vt[1][1]), "+v"(vt[2][0]), "+v"(vt[2][1]), "+v"(vt[3][0]), "+v"(vt[3][1]),
                   "+v"(vt[4][0]), "+v"(vt[4][1]), "+v"(vt[5][0]), "+v"(vt[5][1]), "+v"(vt[6][0]), "+v"(vt[6][1]), "+v"(vt[7][0]), "+v"(vt[7][1]));
#pragma unroll
    for (int u = 0; u < 2; ++u) {                         // 32 keys per MFMA: sub-tiles 2u, 2u+1
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const u32x4 vv = u32x4{vt[u * NDT + dt][0][0], vt[u * NDT + dt][0][1], vt[u * NDT + dt][1][0], vt[u * NDT + dt][1][1]};
        const bf16x8 vf = __builtin_bit_cast(bf16x8, vv);
#pragma unroll
        for (int j = 0; j < QT; ++j) o[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[j][u], o[j][dt], 0, 0, 0);
      }
    }
    asm volatile("" ::: "memory");
  };
  // tiles [0, t_full): inside klen and, if causal, entirely visible to this wave's first (lowest) query tile -- hence to all of them
  int t_full = klen / KT;
  if (p.causal) {
    const int vis = qt0 + wave * QT * 16 + shift + 1;    // keys visible to the wave's lowest query row
    const int tc = vis > 0 ? vis / KT : 0;
    t_full = tc < t_full ? tc : t_full;
  }
  t_full = __builtin_amdgcn_readfirstlane(t_full < ntile ? t_full : ntile);
  int t = 0;
  for (; t < t_full; ++t) tile(t, std::false_type{});
  for (; t < ntile; ++t) tile(t, std::true_type{});
  // ---- finalize: l over the 4 lane groups, write O[q][d = dt*16 + 4g + r] ----
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    float l = l_run[j];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (qrow[j] < qlen) {
      const float inv = l > 0.f ? 1.f / l : 0.f;
      bf16_t* O = reinterpret_cast<bf16_t*>(p.O) + (int64_t)(q0 + qrow[j]) * p.ldo + (int64_t)head * HD;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        bf16x4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (bf16_t)(o[j][dt][r] * inv);
        *reinterpret_cast<bf16x4*>(O + dt * 16 + 4 * g) = w;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// (1c) attn_dma32_kernel (round 4): (1b) on 32x32x16 MFMAs.  At head_dim 64 the kernel is bound by the vector ISSUE port, not by the matrix
//      pipe (one exponential per 64 MFMA flop-lanes): per wave and 64-key tile (1b) issues 32 MFMAs (8 issue cycles
//      each, as many as a 32x32x16 that does twice the work) and runs the soft-max bookkeeping of two query tiles.  Here a wave owns ONE
//      tile of 32 queries (8 waves = 256 queries per workgroup): S^T = K.Q^T as 32 keys x 32 queries per MFMA chain (four 16-deep steps over d), half the MFMA instructions, one
//      lane swap (xor 32) instead of two, one set of running statistics.  The C layout (lane = query l & 31, rows 8i + 4(l >> 5) + r) again IS
//      the B operand of O^T = V^T.P^T: k step s of a 64-key tile takes the lane's accumulator groups i = 2s', 2s' + 1, i.e. keys
//      16s + 4h + r and 16s + 8 + 4h + r, and the V^T fragment is read from LDS (two transposing 8-byte reads) for exactly those rows.
//      Data movement (3-stage LDS-DMA ring, swizzled 128-byte rows) is (1b)'s.  A different summation order than (1b): used where every
//      sequence of the launch takes it (AttnArgs::form, set by the Whisper encoder: always 1500 frames), never chosen by the batch.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float xor_max_32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}

template <int NW>
__device__ __forceinline__ void attn_dma32_body(const AttnArgs& p) {
  constexpr int HD = 64, RB = 128, NSTAGE = 3;
  constexpr int QB = NW * 32;                          // queries per workgroup
  constexpr int NSEG = 8 / NW;                         // 1-KiB segments of each image a wave fills per tile
  constexpr int kStage = 2 * KT * RB;                  // K image then V image
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int seq = blockIdx.z, head = blockIdx.y;
  const int q0 = p.cu_q[seq], qlen = p.cu_q[seq + 1] - q0;
  const int k0 = p.cu_k[seq], klen = p.cu_k[seq + 1] - k0;
  const int qt0 = blockIdx.x * QB;
  if (qt0 >= qlen) return;
  const int kvh = head / (p.n_heads / p.n_kv_heads);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, h = lane >> 5;             // query column of the MFMA; k-group (A / B operands) = row-group half (C layout)
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.Q);
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.K) + (int64_t)kvh * HD;
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.V) + (int64_t)kvh * HD;

  const int shift = klen - qlen;                         // causal: key j visible iff j <= qrow + shift
  int kend = klen;
  if (p.causal) {
    int last = qt0 + QB - 1 < qlen ? qt0 + QB - 1 : qlen - 1;
    kend = last + shift + 1;
    if (kend > klen) kend = klen;
    if (kend < 0) kend = 0;
  }
  const int ntile = (kend + KT - 1) / KT;

  // ---- K / V tiles by LDS-DMA: exactly attn_dma_kernel's ring (wave w fills segments w and w + 4 of both images)
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K + (int64_t)k0 * p.ldk), 0,
                                                                       klen > 0 ? (int)(((int64_t)klen - 1) * p.ldk * 2 + RB) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(V + (int64_t)k0 * p.ldv), 0,
                                                                       klen > 0 ? (int)(((int64_t)klen - 1) * p.ldv * 2 + RB) : 0, 0x00020000);
  int kvo[NSEG], vvo[NSEG];
#pragma unroll
  for (int hh = 0; hh < NSEG; ++hh) {
    const int key = (wave + NW * hh) * 8 + lrow;
    kvo[hh] = (key * (int)p.ldk + lchunk * 8) * 2;
    vvo[hh] = (key * (int)p.ldv + lchunk * 8) * 2;
  }
  const int kstep = KT * (int)p.ldk * 2, vstep = KT * (int)p.ldv * 2;
  int ksoff = 0, vsoff = 0;
  unsigned dslot = 0;
  auto issue = [&]() {
#pragma unroll
    for (int hh = 0; hh < NSEG; ++hh) {
      char* dst = smem + (dslot + (wave + NW * hh) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)dst, 16, kvo[hh], ksoff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)(dst + KT * RB), 16, vvo[hh], vsoff, 0, 0);
    }
    ksoff += kstep; vsoff += vstep;
    dslot = dslot == (NSTAGE - 1) * kStage ? 0u : dslot + kStage;
  };

  if (ntile > 0) issue();
  // Q^T fragments (B operand: lane = query ql, k-group h): d step ds covers d = 16 ds + 8 h .. + 7
  Chunk16 qf[4];
  const int qrow = qt0 + wave * 32 + ql;                 // query index inside the sequence (may be >= qlen)
  {
    const int qc = qrow < qlen ? qrow : qlen - 1;
    const char* qp = reinterpret_cast<const char*>(Q + (int64_t)(q0 + qc) * p.ldq + (int64_t)head * HD);
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[ds] = *reinterpret_cast<const Chunk16*>(qp + ds * 32 + h * 16);
  }
  if (ntile > 1) {
    issue();
    if constexpr (NSEG == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(qf[0].v), "+v"(qf[1].v), "+v"(qf[2].v), "+v"(qf[3].v)::"memory");
    else asm volatile("s_waitcnt vmcnt(2)" : "+v"(qf[0].v), "+v"(qf[1].v), "+v"(qf[2].v), "+v"(qf[3].v)::"memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0].v), "+v"(qf[1].v), "+v"(qf[2].v), "+v"(qf[3].v)::"memory");
  }

  // LDS byte offsets inside a stage.  K (A operand): row = key ql (+ 32 for the second key tile), 16-B chunk (2 ds + h) ^ (row & 7).
  // V^T (transposing read: the 16 lanes of a group fetch a 4-row x 16-column block, lane 4 qq + pp gets rows 0..3 of ITS column):
  // row 4 h + qq (+ 16 s, + 8), chunk (4 dtile + 2 d16 + (pp >> 1)) ^ (row & 7), + 8 bytes for odd pp; row offsets are multiples of 8.
  unsigned kbase[4], vbase[2];
  {
    const int qi = lane & 15, d16 = (lane >> 4) & 1, qq = qi >> 2, pp = qi & 3, vrow = 4 * h + qq;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) kbase[ds] = ql * RB + (((ds * 2 + h) ^ (ql & 7)) << 4);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) vbase[dt] = vrow * RB + (((dt * 4 + d16 * 2 + (pp >> 1)) ^ (vrow & 7)) << 4) + (pp & 1) * 8;
  }

  const float sc = p.scale * 1.44269504088896340736f;    // softmax in base 2
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  unsigned cslot = 0;
  auto tile = [&](const int t, auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    if (t + 1 < ntile) { if constexpr (NSEG == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < ntile) issue();
    const unsigned so = cslot;
    cslot = cslot == (NSTAGE - 1) * kStage ? 0u : cslot + kStage;
    unsigned kaddr[4], vaddr[2];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) kaddr[ds] = so + kbase[ds];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) vaddr[dt] = so + KT * RB + vbase[dt];
    const int kt0 = t * KT;

    // ---- S^T[key][q]: two tiles of 32 keys, four 16-deep steps over d each ----
    f32x16 s[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[t2][e] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        const Chunk16 kf = *reinterpret_cast<const Chunk16*>(smem + (kaddr[ds] + t2 * 32 * RB));
        s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf.v), __builtin_bit_cast(bf16x8, qf[ds].v), s[t2], 0, 0, 0);
      }
    }
    // ---- online softmax (lane owns query ql; its keys: 32 t2 + 8 i + 4 h + r) ----
    const bool need_mask = MASKED && (kt0 + KT > klen || (p.causal && kt0 + KT - 1 > qt0 + wave * 32 + shift));     // wave-uniform
    if constexpr (MASKED) if (need_mask) {
      int key0 = kt0 + 4 * h;
      asm volatile("" : "+v"(key0));
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = key0 + t2 * 32 + (e >> 2) * 8 + (e & 3);
          const bool vis = key < klen && (!p.causal || key <= qrow + shift);
          if (!vis) s[t2][e] = -INFINITY;
        }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int e = 0; e < 16; ++e) mt = fmaxf(mt, s[t2][e]);
    mt = xor_max_32(mt);
    mt *= sc;
    const float m_new = fmaxf(m_run, mt);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    float lsum2[2] = {0.f, 0.f};
    const float nm = -m_use;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
    }
    // Key tile t2 at a time: exponentials of its 16 values, then its two k steps of O^T += V^T P^T -- the second tile's exponentials issue
    // in the shadow of the first tile's MFMAs (a 32x32x16 holds the vector issue port for 8 of its 32 cycles).
    static_for<0, 2>([&](auto T2) {
      constexpr int t2 = decltype(T2)::value;
      u32x2 vt[4][2];
      static_for<0, 4>([&](auto I) {
        constexpr int dt = decltype(I)::value / 2, sl = decltype(I)::value % 2, sg = 2 * t2 + sl;
        vt[dt * 2 + sl][0] = lds_tr16_b64<sg * 16 * RB>(vaddr[dt]);
        vt[dt * 2 + sl][1] = lds_tr16_b64<sg * 16 * RB + 8 * RB>(vaddr[dt]);
      });
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float ev = __builtin_amdgcn_exp2f(fmaf(s[t2][e], sc, nm));
        s[t2][e] = ev;
        lsum2[e & 1] += ev;
      }
      bf16x8 pf[2];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[sl][r] = (bf16_t)s[t2][8 * sl + r]; pf[sl][4 + r] = (bf16_t)s[t2][8 * sl + 4 + r]; }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vt[0][0]), "+v"(vt[0][1]), "+v"(vt[1][0]), "+v"(vt[1][1]), "+v"(vt[2][0]), "+v"(vt[2][1]), "+v"(vt[3][0]), "+v"(vt[3][1]));
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const u32x4 vv = u32x4{vt[dt * 2 + sl][0][0], vt[dt * 2 + sl][0][1], vt[dt * 2 + sl][1][0], vt[dt * 2 + sl][1][1]};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[sl], o[dt], 0, 0, 0);
        }
    });
    l_run = l_run * alpha + (lsum2[0] + lsum2[1]);
    m_run = m_new;
    asm volatile("" ::: "memory");
  };
  int t_full = klen / KT;
  if (p.causal) {
    const int vis = qt0 + wave * 32 + shift + 1;         // keys visible to the wave's lowest query row
    const int tc = vis > 0 ? vis / KT : 0;
    t_full = tc < t_full ? tc : t_full;
  }
  t_full = __builtin_amdgcn_readfirstlane(t_full < ntile ? t_full : ntile);
  int t = 0;
  for (; t < t_full; ++t) tile(t, std::false_type{});
  for (; t < ntile; ++t) tile(t, std::true_type{});
  // ---- finalize: l over the two lane halves, write O[q][d = 32 dt + 8 i + 4 h + r] ----
  float l = l_run;
  l += __shfl_xor(l, 32, 64);
  if (qrow < qlen) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    bf16_t* O = reinterpret_cast<bf16_t*>(p.O) + (int64_t)(q0 + qrow) * p.ldo + (int64_t)head * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16x4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = (bf16_t)(o[dt][4 * i + r] * inv);
        *reinterpret_cast<bf16x4*>(O + dt * 32 + 8 * i + 4 * h) = w;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Decode attention (q_len = 1 per row), flash-decoding inside one workgroup per (row, kv head):
//   * the G query heads that share the kv head are the 16 "query columns" of the 16x16 MFMA (G <= 8 valid)
//   * optional fused RoPE + KV-cache append (inv_freq != nullptr): q and the new k are rotated on load, the new k / v row
//     is written to the tail cache first and then read back like any other key (workgroup-scope visibility via the barrier)
//   * the 4 waves split the 64-key tiles; K fragments go HBM -> VGPR directly (A operand), V tiles go through a per-wave
//     LDS image for the transposed reads; per-wave (m, l, O^T) partials are merged through LDS at the end.
// The merge of two softmax partials (m, l, O): the ONE formula every decode-attention result is folded with, in registers (two waves) or from LDS (four /
// eight waves) -- so a row's bits do not depend on how many waves shared its keys, and the wave count may follow the row count.
constexpr int kDecSlots = 8;
__device__ __forceinline__ float decode_merge_weights(float m1, float m2, float& w1, float& w2) {
  const float m = fmaxf(m1, m2);
  const float mu = (m == -INFINITY) ? 0.f : m;
  w1 = exp2f(m1 - mu);
  w2 = exp2f(m2 - mu);
  return m;
}
__device__ __forceinline__ float decode_merge_val(float x1, float w1, float x2, float w2) { return __builtin_fmaf(x1, w1, x2 * w2); }

template <typename T, int HD, int NW>
__global__ __launch_bounds__(NW * 64, NW == 2 ? 2 : 1) void decode_attn_kernel(const DecodeAttnArgs p) {      // (two waves: a full launch -- at least two waves per SIMD)
  constexpr int kDecThreads = NW * 64;
  constexpr int RB = HD * (int)sizeof(T);
  constexpr int NCH = RB / 16;
  constexpr int SWZ = (RB == 128) ? 7 : 15;
  constexpr int NBLK = RB / 64;
  constexpr int NDT = HD / 16;
  constexpr int VEC = TypeInfo<T>::kVec;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sq = smem;                                   // [16][RB]  rotated queries (rows >= G are zero)
  char* sknew = smem + 16 * RB;                      // [RB] rotated new key, [RB] new value (fused append only)
  char* svnew = sknew + RB;
  char* sv = svnew + RB;                             // [NW waves][64][RB]  V tiles; reused for the merge
  const int row = blockIdx.x, kvh = blockIdx.y;
  const int G = p.n_heads / p.n_kv_heads;
  // srow: the row's STATE slot (position, cache rows).  Identity, except under the decode scheduler, whose compact activation rows
  // (Q / O) map onto the row slots of its pool through rowmap (greedy only: the slot is also the utterance and the tail row).
  const int srow = p.rowmap ? p.rowmap[row] : row;
  const int utt = srow / p.beams;
  const int plen = p.prefix_len[utt];
  const int ntail = p.step ? *p.step : p.cpos[srow] + 1;      // no shared step counter: every row is at its own step (scheduler)
  const int ctx = plen + ntail;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;
  char* svw = sv + wave * (64 * RB);                   // this wave's V image
  const T* qrow = reinterpret_cast<const T*>(p.Q) + (int64_t)row * p.ldq;
  T* Kt = reinterpret_cast<T*>(const_cast<void*>(p.Kt));
  T* Vt = reinterpret_cast<T*>(const_cast<void*>(p.Vt));
  // fused append: the new token's key index; its K/V come from LDS (sknew/svnew), never from the just-written cache row
  const int newkey = p.inv_freq ? plen + p.cpos[srow] : -1;

  constexpr int RPI = 64 / NCH;                        // V rows per load instruction (load_tile)
  const int vch = lane % NCH, vro = lane / NCH;
  auto kv_row = [&](const void* pre, const void* tail, int j) -> const char* {
    if (j < plen) return reinterpret_cast<const char*>(pre) + (((int64_t)utt * p.n_kv_heads + kvh) * p.max_prefix + j) * RB;
    const int t = j - plen;
    const int src = p.anc ? p.anc[(int64_t)row * p.max_tail + t] : srow;
    return reinterpret_cast<const char*>(tail) + (((int64_t)src * p.n_kv_heads + kvh) * p.max_tail + t) * RB;
  };
  // K fragments straight from the cache (A operand row = key (lane & 15), 16-byte chunk blk*4 + g); V row of key kt0 + lane
  auto load_tile = [&](int kt0, Chunk16 (&kf)[4][NBLK]) {
    // (wave-uniform) a tile inside the context that does not hold the token being appended: no per-lane predicates, no zero fill
    const bool whole = kt0 + 64 <= ctx && !(newkey >= kt0 && newkey < kt0 + 64);
    if (whole) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const char* kr = kv_row(p.Kp, p.Kt, kt0 + kt * 16 + qi);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) kf[kt][b] = *reinterpret_cast<const Chunk16*>(kr + (b * 4 + g) * 16);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int key = kt0 + kt * 16 + qi;
        const char* kr = (key < ctx && key != newkey) ? kv_row(p.Kp, p.Kt, key) : nullptr;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          if (kr) kf[kt][b] = *reinterpret_cast<const Chunk16*>(kr + (b * 4 + g) * 16);
          else kf[kt][b].v = u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
    // V: straight into this wave's LDS image [key row][chunk ^ (row & SWZ)] by LDS-DMA, no registers.  Lane l knows the row of key kt0 + l; instruction c
    // moves rows c RPI + l / NCH -- whole 128-byte lines -- and lane l's 16 bytes land at image offset 16 (64 c + l) = row RB + 16 (l % NCH): the swizzle is
    // applied to the SOURCE chunk.  Rows past the context (and the row of the token being appended, patched from LDS in tile()) are zero-filled.
    const int key = kt0 + lane;
    const char* vp = (whole || (key < ctx && key != newkey)) ? kv_row(p.Vp, p.Vt, key) : nullptr;
    const unsigned vlo = (unsigned)(uintptr_t)vp, vhi = (unsigned)((uintptr_t)vp >> 32);
    if (whole) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int r = c * RPI + vro;
        const uintptr_t bp = (uintptr_t)(unsigned)__shfl((int)vlo, r, 64) | ((uintptr_t)(unsigned)__shfl((int)vhi, r, 64) << 32);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bp + ((vch ^ (r & SWZ)) << 4)), (__attribute__((address_space(3))) void*)(svw + c * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int r = c * RPI + vro;
        const uintptr_t bp = (uintptr_t)(unsigned)__shfl((int)vlo, r, 64) | ((uintptr_t)(unsigned)__shfl((int)vhi, r, 64) << 32);
        char* dst = svw + c * 1024;
        if (bp) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bp + ((vch ^ (r & SWZ)) << 4)), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        else reinterpret_cast<Chunk16*>(dst)[lane].v = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };

  // ---- the first K/V tile of every wave is requested before anything else: its latency hides the RoPE phase ----
  Chunk16 kf[4][NBLK];
  int kt0 = wave * 64;
  if (kt0 < ctx) load_tile(kt0, kf);

  // ---- phase 0: queries (+ RoPE) into LDS; new k (+ RoPE) / v appended to the tail cache and kept in LDS ----
  for (int i = tid + G * HD; i < 16 * HD; i += kDecThreads) reinterpret_cast<T*>(sq)[i] = from_f32<T>(0.f);   // rows >= G
  if (p.inv_freq) {
    constexpr int half = HD / 2;
    const float pos = (float)p.pos[srow];
    for (int i = tid; i < (G + 1) * half; i += kDecThreads) {
      const int h = i / half, d = i - h * half;
      float sn, cs;
      if (p.rope_cs) { const float2 t = p.rope_cs[(int64_t)srow * half + d]; cs = t.x; sn = t.y; }
      else sincosf(pos * p.inv_freq[d], &sn, &cs);
      const T* src = h < G ? qrow + (int64_t)(kvh * G + h) * HD : qrow + (int64_t)(p.n_heads + kvh) * HD;
      const float x0 = to_f32(src[d]), x1 = to_f32(src[d + half]);
      const T r0 = from_f32<T>(x0 * cs - x1 * sn), r1 = from_f32<T>(x1 * cs + x0 * sn);
      if (h < G) {
        T* dst = reinterpret_cast<T*>(sq + h * RB);     // sq rows are read as B operand with row = lane & 15: no swizzle
        dst[d] = r0; dst[d + half] = r1;
      } else {
        T* kd = Kt + (((int64_t)srow * p.n_kv_heads + kvh) * p.max_tail + p.cpos[srow]) * HD;
        kd[d] = r0; kd[d + half] = r1;
        reinterpret_cast<T*>(sknew)[d] = r0; reinterpret_cast<T*>(sknew)[d + half] = r1;
      }
    }
    const T* vsrc = qrow + (int64_t)(p.n_heads + p.n_kv_heads + kvh) * HD;
    T* vd = Vt + (((int64_t)srow * p.n_kv_heads + kvh) * p.max_tail + p.cpos[srow]) * HD;
    for (int i = tid; i < HD; i += kDecThreads) { const T v = vsrc[i]; vd[i] = v; reinterpret_cast<T*>(svnew)[i] = v; }
  } else {
    for (int i = tid; i < G * HD; i += kDecThreads) reinterpret_cast<T*>(sq)[i] = qrow[(int64_t)kvh * G * HD + i];
  }
  __syncthreads();

  Chunk16 qf[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) qf[b] = *reinterpret_cast<const Chunk16*>(sq + qi * RB + (b * 4 + g) * 16);

  const float sc = p.scale * 1.44269504088896340736f;
  f32x4 o[NDT];
  float m_run, l_run;

  // One 64-key tile into the running (m_run, l_run, o) of the current SLOT.
  // (requesting tile i + 1 before computing tile i -- a second register set for the K fragments and the V row -- was measured for the 2-wave
  // form: nothing at 8 rows, 20 -> 26 us at 97 rows through the lost occupancy; the other workgroups of the CU hide the latency instead)
  auto tile = [&](int kt0, bool loaded) {
    if (!loaded) load_tile(kt0, kf);
    if (newkey >= kt0 && newkey < kt0 + 64) {             // wave-uniform: this tile holds the token being appended
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (kt0 + kt * 16 + qi == newkey) {
#pragma unroll
          for (int b = 0; b < NBLK; ++b) kf[kt][b] = *reinterpret_cast<const Chunk16*>(sknew + (b * 4 + g) * 16);
        }
      const int r = newkey - kt0;                         // its V row: from LDS into the image (the lanes that zero-filled it in load_tile)
      if (lane / NCH == r % RPI) {
        const int ch = lane % NCH;
        *reinterpret_cast<Chunk16*>(svw + r * RB + (ch << 4)) = *reinterpret_cast<const Chunk16*>(svnew + ((ch ^ (r & SWZ)) << 4));
      }
    }
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NBLK; ++b) MmaT<T>::run(kf[kt][b], qf[b], s[kt]);
    }
    float mt = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt0 + kt * 16 + 4 * g + r;
        float v = key < ctx ? s[kt][r] * sc : -INFINITY;
        s[kt][r] = v;
        mt = fmaxf(mt, v);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_use);
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { float e = exp2f(s[kt][r] - m_use); s[kt][r] = e; lsum += e; }
    l_run = l_run * alpha + lsum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[i] *= alpha;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's V tile (LDS-DMA, zero fill, patch) is in LDS before its own reads
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pf[r] = (bf16_t)s[2 * u][r]; pf[4 + r] = (bf16_t)s[2 * u + 1][r]; }
        const int qq = qi >> 2, pp = qi & 3;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const int c = dt * 2 + (pp >> 1);
          const int row0 = (2 * u) * 16 + 4 * g + qq, row1 = row0 + 16;
          auto a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(svw + row0 * RB + ((c ^ (row0 & SWZ)) << 4) + (pp & 1) * 8));
          auto a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(svw + row1 * RB + ((c ^ (row1 & SWZ)) << 4) + (pp & 1) * 8));
          bf16x4 v0 = __builtin_bit_cast(bf16x4, a0), v1 = __builtin_bit_cast(bf16x4, a1);
          bf16x8 vf;
#pragma unroll
          for (int r = 0; r < 4; ++r) { vf[r] = v0[r]; vf[4 + r] = v1[r]; }
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const int c = dt * 4 + (qi >> 2);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int vrow = kt * 16 + 4 * g + r;
            float vv = *reinterpret_cast<const float*>(svw + vrow * RB + ((c ^ (vrow & SWZ)) << 4) + (qi & 3) * 4);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, s[kt][r], o[dt], 0, 0, 0);
          }
        }
    }
  };
  // A slot = the tiles slot, slot + kDecSlots, ... of the context, accumulated in that order (online softmax); its partial (m, l, O^T) with l summed over
  // the lane groups.  What a row's result is made of does not depend on the wave that computes a slot: see decode_merge.
  auto run_slot = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run = -INFINITY; l_run = 0.f;
    for (int kt0 = slot * 64; kt0 < ctx; kt0 += kDecSlots * 64) tile(kt0, kt0 == wave * 64);      // (the wave's first tile was requested before the RoPE phase)
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
  };

  float* sm = reinterpret_cast<float*>(sv);            // after the barrier: [P][16] m, [P][16] l, [P][HD][16] O^T (P partials, reusing the V-tile region)
  if constexpr (NW == 2) {
    // Two waves: wave w owns the chain of slots w, w + 2, w + 4, w + 6, folded IN THAT ORDER in registers -- the fold the merge below applies to
    // the same slots when eight waves leave them in LDS.  (An empty slot is folded too: x (+) empty = x exactly, but for the sign of a zero.)
    constexpr int kSlotUnroll = HD == 64 ? kDecSlots / 2 : 1;
    f32x4 oa[NDT];
    float ma = 0.f, la = 0.f;
#pragma unroll kSlotUnroll
    for (int j = 0; j < kDecSlots / 2; ++j) {      // (head_dim 128 is at the register limit: one copy of the tile code there; 64: unrolled, 20.8 -> 20.1 us at 97 rows)
      run_slot(wave + 2 * j);
      if (j == 0) {
#pragma unroll
        for (int i = 0; i < NDT; ++i) oa[i] = o[i];
        ma = m_run; la = l_run;
        continue;
      }
      float w1, w2;
      const float mm = decode_merge_weights(ma, m_run, w1, w2);
      la = decode_merge_val(la, w1, l_run, w2);
#pragma unroll
      for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) oa[i][r] = decode_merge_val(oa[i][r], w1, o[i][r], w2);
      ma = mm;
    }
    __syncthreads();
    float* sl = sm + 2 * 16;
    float* so = sl + 2 * 16;
    if (g == 0) { sm[wave * 16 + qi] = ma; sl[wave * 16 + qi] = la; }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) so[(wave * HD + dt * 16 + 4 * g + r) * 16 + qi] = oa[dt][r];
  } else {
    // Four / eight waves: every slot's partial goes to LDS (a wave keeps its kDecSlots / NW partials in registers until all waves are done with their V images)
    constexpr int PW = kDecSlots / NW;
    f32x4 op[PW][NDT];
    float mp[PW], lp[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      run_slot(wave + j * NW);
#pragma unroll
      for (int i = 0; i < NDT; ++i) op[j][i] = o[i];
      mp[j] = m_run; lp[j] = l_run;
    }
    __syncthreads();
    float* sl = sm + kDecSlots * 16;
    float* so = sl + kDecSlots * 16;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int slot = wave + j * NW;
      if (g == 0) { sm[slot * 16 + qi] = mp[j]; sl[slot * 16 + qi] = lp[j]; }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) so[(slot * HD + dt * 16 + 4 * g + r) * 16 + qi] = op[j][dt][r];
    }
  }
  __syncthreads();
  // ---- the canonical merge: (((s0 (+) s2) (+) s4) (+) s6) (+) (((s1 (+) s3) (+) s5) (+) s7), element by element ----
  constexpr int P = NW == 2 ? 2 : kDecSlots;
  const float* sl = sm + P * 16;
  const float* so = sl + P * 16;
  T* O = reinterpret_cast<T*>(p.O) + (int64_t)row * p.ldo + (int64_t)kvh * G * HD;
  for (int i = tid; i < G * HD; i += kDecThreads) {
    const int h = i / HD, d = i - h * HD;
    float mc[2], lc[2], oc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {      // chain c: slots c, c + 2, ... (two waves: already folded by wave c)
      mc[c] = sm[c * 16 + h]; lc[c] = sl[c * 16 + h]; oc[c] = so[(c * HD + d) * 16 + h];
      if constexpr (NW != 2) {
#pragma unroll
        for (int j = 1; j < kDecSlots / 2; ++j) {
          const int sidx = c + 2 * j;
          float w1, w2;
          const float mm = decode_merge_weights(mc[c], sm[sidx * 16 + h], w1, w2);
          lc[c] = decode_merge_val(lc[c], w1, sl[sidx * 16 + h], w2);
          oc[c] = decode_merge_val(oc[c], w1, so[(sidx * HD + d) * 16 + h], w2);
          mc[c] = mm;
        }
      }
    }
    float w1, w2;
    (void)decode_merge_weights(mc[0], mc[1], w1, w2);
    const float den = decode_merge_val(lc[0], w1, lc[1], w2);
    const float num = decode_merge_val(oc[0], w1, oc[1], w2);
    O[i] = from_f32<T>(den > 0.f ? num / den : 0.f);
  }
}

// Beam search (round 4): FOUR beams of one utterance per workgroup.  With a GQA group of 4 the per-row kernel above fills 4 of the MFMA's 16
// query columns, and the 15 beams of an utterance each fetch the same prefix K / V (313 keys for a 16 s clip) -- 960 workgroups per layer
// and step.  Here the 16 columns are (beam bi, head h) = 4 x 4: the prefix tiles are read once for four beams and every column is real
// work; each beam's tail (its own ancestry-addressed rows, <= 64 positions) is one more key tile whose scores only count in that beam's four
// columns (the others are masked to -inf).  Per column the arithmetic is the per-row kernel's (same tile softmax, same wave merge); what
// changes is which keys share a tile, so the fp32 merge order differs from the per-row kernel's: the choice follows the decode MODE and the
// model (beams > 1, bf16, head_dim 64, group 4), never a row count.  bf16 / head_dim 64 / 4 waves only.
__global__ __launch_bounds__(256) void decode_attn_beam4_kernel(const DecodeAttnArgs p) {
  typedef bf16_t T;
  constexpr int HD = 64, NW = 4, G = 4, R = 4;
  constexpr int kDecThreads = NW * 64;
  constexpr int RB = HD * 2, NCH = RB / 16, SWZ = 7, NBLK = RB / 64, NDT = HD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sq = smem;                                   // [16][RB]  rotated queries, row = bi * G + h (rows of missing beams are zero)
  char* sknew = smem + 16 * RB;                      // [R][RB] rotated new keys, [R][RB] new values
  char* svnew = sknew + R * RB;
  char* sv = svnew + R * RB;                         // [NW waves][64][RB]  V tiles; reused for the merge
  const int ngrp = (p.beams + R - 1) / R;
  const int utt = blockIdx.x / ngrp, b0 = (blockIdx.x - utt * ngrp) * R, kvh = blockIdx.y;
  const int nb = p.beams - b0 < R ? p.beams - b0 : R;          // beams of this workgroup
  const int row0 = utt * p.beams + b0;
  const int plen = p.prefix_len[utt];
  const int ntail = *p.step;                                   // beam search: every row is at the same step
  const int nP = (plen + 63) >> 6;                             // prefix key tiles; tile nP + bi = the tail of beam bi
  const int ntiles = nP + nb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;
  T* Kt = reinterpret_cast<T*>(const_cast<void*>(p.Kt));
  T* Vt = reinterpret_cast<T*>(const_cast<void*>(p.Vt));

  constexpr int RPI = 64 / NCH;
  const int vch = lane % NCH, vro = lane / NCH;
  // K fragments straight from the caches (A operand row = key (lane & 15), 16-byte chunk blk * 4 + g); V row of key `lane`
  auto key_row = [&](const void* pre, const void* tail, int tile, int kk, bool& ok) -> const char* {
    if (tile < nP) {
      const int j = tile * 64 + kk;
      ok = j < plen;
      return reinterpret_cast<const char*>(pre) + (((int64_t)utt * p.n_kv_heads + kvh) * p.max_prefix + j) * RB;
    }
    const int bi = tile - nP, row = row0 + bi;
    const int newpos = p.inv_freq ? p.cpos[row] : -1;          // the token being appended comes from LDS, never from the just-written cache row
    ok = kk < ntail && kk != newpos;
    const int src = (ok && p.anc) ? p.anc[(int64_t)row * p.max_tail + kk] : row;
    return reinterpret_cast<const char*>(tail) + (((int64_t)src * p.n_kv_heads + kvh) * p.max_tail + kk) * RB;
  };
  auto load_tile = [&](int tile, Chunk16 (&kf)[4][NBLK], Chunk16 (&vr)[NCH]) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      bool ok;
      const char* kr = key_row(p.Kp, p.Kt, tile, kt * 16 + qi, ok);
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        if (ok) kf[kt][b] = *reinterpret_cast<const Chunk16*>(kr + (b * 4 + g) * 16);
        else kf[kt][b].v = u32x4{0u, 0u, 0u, 0u};
      }
    }
    // V: whole 128-byte lines per row, RPI rows per instruction (decode_attn_kernel: load_tile); lane l knows the row of key l
    bool okv;
    const char* vp = key_row(p.Vp, p.Vt, tile, lane, okv);
    if (!okv) vp = nullptr;
    const unsigned vlo = (unsigned)(uintptr_t)vp, vhi = (unsigned)((uintptr_t)vp >> 32);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int r = c * RPI + vro;
      const uintptr_t bp = (uintptr_t)(unsigned)__shfl((int)vlo, r, 64) | ((uintptr_t)(unsigned)__shfl((int)vhi, r, 64) << 32);
      if (bp) vr[c] = *reinterpret_cast<const Chunk16*>(bp + vch * 16);
      else vr[c].v = u32x4{0u, 0u, 0u, 0u};
    }
  };

  Chunk16 kf[4][NBLK], vr[NCH];
  int tile = wave;
  if (tile < ntiles) load_tile(tile, kf, vr);                  // the first tile's latency hides the RoPE phase

  // ---- phase 0: queries (+ RoPE) into LDS; new k (+ RoPE) / v of every beam appended to its tail cache row and kept in LDS ----
  for (int i = tid + nb * G * HD; i < 16 * HD; i += kDecThreads) reinterpret_cast<T*>(sq)[i] = from_f32<T>(0.f);
  if (p.inv_freq) {
    constexpr int half = HD / 2;
    for (int i = tid; i < nb * (G + 1) * half; i += kDecThreads) {
      const int bi = i / ((G + 1) * half), rem = i - bi * (G + 1) * half, h = rem / half, d = rem - h * half;
      const int row = row0 + bi;
      const T* qrow = reinterpret_cast<const T*>(p.Q) + (int64_t)row * p.ldq;
      float sn, cs;
      if (p.rope_cs) { const float2 t = p.rope_cs[(int64_t)row * half + d]; cs = t.x; sn = t.y; }
      else sincosf((float)p.pos[row] * p.inv_freq[d], &sn, &cs);
      const T* src = h < G ? qrow + (int64_t)(kvh * G + h) * HD : qrow + (int64_t)(p.n_heads + kvh) * HD;
      const float x0 = to_f32(src[d]), x1 = to_f32(src[d + half]);
      const T r0 = from_f32<T>(x0 * cs - x1 * sn), r1 = from_f32<T>(x1 * cs + x0 * sn);
      if (h < G) {
        T* dst = reinterpret_cast<T*>(sq + (bi * G + h) * RB);
        dst[d] = r0; dst[d + half] = r1;
      } else {
        T* kd = Kt + (((int64_t)row * p.n_kv_heads + kvh) * p.max_tail + p.cpos[row]) * HD;
        kd[d] = r0; kd[d + half] = r1;
        reinterpret_cast<T*>(sknew + bi * RB)[d] = r0; reinterpret_cast<T*>(sknew + bi * RB)[d + half] = r1;
      }
    }
    for (int i = tid; i < nb * HD; i += kDecThreads) {
      const int bi = i / HD, d = i - bi * HD, row = row0 + bi;
      const T v = (reinterpret_cast<const T*>(p.Q) + (int64_t)row * p.ldq + (int64_t)(p.n_heads + p.n_kv_heads + kvh) * HD)[d];
      (Vt + (((int64_t)row * p.n_kv_heads + kvh) * p.max_tail + p.cpos[row]) * HD)[d] = v;
      reinterpret_cast<T*>(svnew + bi * RB)[d] = v;
    }
  } else {
    for (int i = tid; i < nb * G * HD; i += kDecThreads) {
      const int bi = i / (G * HD), rem = i - bi * G * HD;
      reinterpret_cast<T*>(sq)[i] = (reinterpret_cast<const T*>(p.Q) + (int64_t)(row0 + bi) * p.ldq)[(int64_t)kvh * G * HD + rem];
    }
  }
  __syncthreads();

  Chunk16 qf[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) qf[b] = *reinterpret_cast<const Chunk16*>(sq + qi * RB + (b * 4 + g) * 16);

  const float sc = p.scale * 1.44269504088896340736f;
  f32x4 o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  char* svw = sv + wave * (64 * RB);
  const int mybeam = qi >> 2;                                   // the beam this lane's query column belongs to (G = 4)

  for (; tile < ntiles; tile += NW) {
    if (tile != wave) load_tile(tile, kf, vr);
    const bool is_tail = tile >= nP;
    const int bi = tile - nP;                                   // (tail tiles)
    const int nvalid = is_tail ? ntail : (plen - tile * 64 < 64 ? plen - tile * 64 : 64);
    if (is_tail && p.inv_freq) {                                // this tile holds the token beam bi appends in this step
      const int newpos = p.cpos[row0 + bi];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (kt * 16 + qi == newpos) {
#pragma unroll
          for (int b = 0; b < NBLK; ++b) kf[kt][b] = *reinterpret_cast<const Chunk16*>(sknew + bi * RB + (b * 4 + g) * 16);
        }
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (c * RPI + vro == newpos) vr[c] = *reinterpret_cast<const Chunk16*>(svnew + bi * RB + vch * 16);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int r = c * RPI + vro;
      *reinterpret_cast<Chunk16*>(svw + r * RB + ((vch ^ (r & SWZ)) << 4)) = vr[c];
    }
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NBLK; ++b) MmaT<T>::run(kf[kt][b], qf[b], s[kt]);
    }
    const bool mine = !is_tail || mybeam == bi;                 // a tail tile only counts in its own beam's columns
    float mt = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = kt * 16 + 4 * g + r;
        float v = (mine && kk < nvalid) ? s[kt][r] * sc : -INFINITY;
        s[kt][r] = v;
        mt = fmaxf(mt, v);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_use);
    float lsum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { float e = exp2f(s[kt][r] - m_use); s[kt][r] = e; lsum += e; }
    l_run = l_run * alpha + lsum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[i] *= alpha;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's V tile writes have landed before its own reads
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bf16x8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pf[r] = (bf16_t)s[2 * u][r]; pf[4 + r] = (bf16_t)s[2 * u + 1][r]; }
      const int qq = qi >> 2, pp = qi & 3;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int c = dt * 2 + (pp >> 1);
        const int rw0 = (2 * u) * 16 + 4 * g + qq, rw1 = rw0 + 16;
        auto a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(svw + rw0 * RB + ((c ^ (rw0 & SWZ)) << 4) + (pp & 1) * 8));
        auto a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(svw + rw1 * RB + ((c ^ (rw1 & SWZ)) << 4) + (pp & 1) * 8));
        bf16x4 v0 = __builtin_bit_cast(bf16x4, a0), v1 = __builtin_bit_cast(bf16x4, a1);
        bf16x8 vf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { vf[r] = v0[r]; vf[4 + r] = v1[r]; }
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // the tile's V reads are done before the next tile overwrites this wave's image
  }
  // ---- merge the NW waves: partials (m, l, O^T[d][column]) through LDS (reusing the V-tile region) ----
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  __syncthreads();
  float* sm = reinterpret_cast<float*>(sv);            // [NW][16] m, [NW][16] l, [NW][HD][16] O^T
  float* sl = sm + NW * 16;
  float* so = sl + NW * 16;
  if (g == 0) { sm[wave * 16 + qi] = m_run; sl[wave * 16 + qi] = l_run; }
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) so[(wave * HD + dt * 16 + 4 * g + r) * 16 + qi] = o[dt][r];
  __syncthreads();
  for (int i = tid; i < nb * G * HD; i += kDecThreads) {
    const int col = i / HD, d = i - col * HD, bi = col / G, h = col - bi * G;
    float mstar = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) mstar = fmaxf(mstar, sm[w * 16 + col]);
    if (mstar == -INFINITY) mstar = 0.f;
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = exp2f(sm[w * 16 + col] - mstar);
      num += f * so[(w * HD + d) * 16 + col];
      den += f * sl[w * 16 + col];
    }
    T* O = reinterpret_cast<T*>(p.O) + (int64_t)(row0 + bi) * p.ldo + (int64_t)(kvh * G + h) * HD;
    O[d] = from_f32<T>(den > 0.f ? num / den : 0.f);
  }
}

int launch_dec_beam4(const DecodeAttnArgs& a, hipStream_t st) {
  constexpr int RB = 128, NW = 4, R = 4;
  constexpr size_t tiles = (size_t)NW * 64 * RB, merge = (size_t)(2 * NW * 16 + NW * 64 * 16) * 4;
  constexpr size_t smem = 16 * RB + 2 * R * RB + (tiles > merge ? tiles : merge);
  const int n_utts = a.rows / a.beams, ngrp = (a.beams + R - 1) / R;
  hipLaunchKernelGGL(decode_attn_beam4_kernel, dim3(n_utts * ngrp, a.n_kv_heads), dim3(NW * 64), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int HD>
int launch_attn_t(const AttnArgs& a, hipStream_t st) {
  dim3 grid(ceil_div(a.max_q, 64), a.n_heads, a.n_seqs);
  hipLaunchKernelGGL((attn_varlen_kernel<T, HD>), grid, dim3(kThreads), 0, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int QT>
int launch_attn_dma(const AttnArgs& a, hipStream_t st) {
  constexpr int smem = 3 * 2 * KT * 128;
  auto kern = attn_dma_kernel<QT>;
  static bool attr = false;
  if (!attr) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr = true;
  }
  dim3 grid(ceil_div(a.max_q, 64 * QT), a.n_heads, a.n_seqs);
  hipLaunchKernelGGL(kern, grid, dim3(kThreads), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}


// 8 waves x 32 queries: 256-query workgroups halve how often a (sequence, head)'s K / V tiles are fetched (Whisper: 12 -> 6 workgroups per
// head); two workgroups = 16 waves per CU at <= 128 VGPRs.  (Non-template entry point: hipcc emitted no host stub for a __global__ template
// with this body.)
__global__ __launch_bounds__(512, 2) void attn_dma32_kernel(const AttnArgs p) { attn_dma32_body<8>(p); }

int launch_attn_dma32(const AttnArgs& a, hipStream_t st) {
  constexpr int smem = 3 * 2 * KT * 128;
  static bool attr = false;
  if (!attr) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dma32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr = true;
  }
  dim3 grid(ceil_div(a.max_q, 256), a.n_heads, a.n_seqs);
  hipLaunchKernelGGL(attn_dma32_kernel, grid, dim3(512), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int HD, int NWF = 0>
int launch_dec_t(const DecodeAttnArgs& a, hipStream_t st) {
  constexpr int RB = HD * (int)sizeof(T);
  constexpr int NW = NWF ? NWF : (8 * 64 * RB <= 128 * 1024) ? 8 : 4;      // waves per workgroup: 8 x 64 keys in one pass when the V tiles fit
  constexpr int P = NW == 2 ? 2 : kDecSlots;      // partials that meet in LDS (decode_attn_kernel: the canonical merge)
  constexpr size_t tiles = (size_t)NW * 64 * RB, merge = (size_t)(2 * P * 16 + P * HD * 16) * 4;
  constexpr size_t smem = 18 * RB + (tiles > merge ? tiles : merge);
  static_assert(smem <= 160 * 1024, "LDS");
  auto kern = decode_attn_kernel<T, HD, NW>;
  static bool attr = false;
  if (!attr && smem > 48 * 1024) {
    OAVSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = true;
  }
  dim3 grid(a.rows, a.n_kv_heads);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, st, a);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_attention(const AttnArgs& a, hipStream_t st) {
  if (a.n_seqs <= 0 || a.max_q <= 0) return 0;
  OAVSR_REQUIRE(a.head_dim == 64 || a.head_dim == 128, "attention: head_dim %d unsupported", a.head_dim);
  OAVSR_REQUIRE(a.n_heads % a.n_kv_heads == 0, "attention: heads %d not a multiple of kv heads %d", a.n_heads, a.n_kv_heads);
  const int vec = a.dtype == DT_F32 ? 4 : 8;
  OAVSR_REQUIRE(a.ldq % vec == 0 && a.ldk % vec == 0 && a.ldv % vec == 0 && a.ldo % vec == 0, "attention: rows must be 16-byte aligned");
  OAVSR_REQUIRE((((uintptr_t)a.Q | (uintptr_t)a.K | (uintptr_t)a.V | (uintptr_t)a.O) & 15) == 0, "attention: pointers must be 16-byte aligned");
  if (a.dtype == DT_F32) return a.head_dim == 64 ? launch_attn_t<float, 64>(a, st) : launch_attn_t<float, 128>(a, st);
  if (a.head_dim == 64) {
    if (a.form == 32) return launch_attn_dma32(a, st);      // the 32x32x16 form: a property of the CALL SITE (Whisper encoder), never of the batch
    // two query tiles per wave (128-row workgroups) once the sequences are long enough that the ragged last workgroup is cheap
    if (a.max_q >= 1024) return launch_attn_dma<2>(a, st);
    return launch_attn_dma<1>(a, st);
  }
  return launch_attn_t<bf16_t, 128>(a, st);
}

int launch_decode_attention(const DecodeAttnArgs& a, hipStream_t st) {
  if (a.rows <= 0) return 0;
  OAVSR_REQUIRE(a.head_dim == 64 || a.head_dim == 128, "decode attention: head_dim %d unsupported", a.head_dim);
  OAVSR_REQUIRE(a.n_heads % a.n_kv_heads == 0 && a.n_heads / a.n_kv_heads <= 8, "decode attention: GQA group must be <= 8");
  OAVSR_REQUIRE(!a.rowmap || (a.beams == 1 && !a.anc), "decode attention: a row map needs greedy rows (beams == 1, no ancestry table)");
  OAVSR_REQUIRE(a.step || a.cpos, "decode attention: per-row tail lengths come from cpos[] when there is no step counter");
  if (a.dtype == DT_F32) return a.head_dim == 64 ? launch_dec_t<float, 64>(a, st) : launch_dec_t<float, 128>(a, st);
  // beam search (B * K x kv heads workgroups, several per CU): 4-wave workgroups with half the LDS pack more of them onto a CU
  // than 8-wave ones.  The choice follows the decode MODE (beams > 1), never the row count: the key -> wave partition and with
  // it the fp32 merge order of a row is then the same whether its utterance is decoded alone or inside a batch.
  if (a.head_dim == 64 && a.beams > 1) {
    // GQA group of 4 (Llama-3.2-1B): four beams of an utterance per workgroup share the prefix K / V tiles (decode_attn_beam4_kernel).
    // Its tail tile holds <= 64 positions, so it serves workspaces whose max_new is <= 64 (the reference decodes 32 tokens, eval_OmniAVSR.py:
    // max_dec_tokens); max_tail is the WORKSPACE's max_new and decode_beam insists on max_new == that, so the kernel -- and its fp32 merge
    // order -- is fixed by (model, max_new), the same for an utterance alone and inside any batch ("what a result may depend on")
    if (a.n_heads == 4 * a.n_kv_heads && a.step && !a.rowmap && a.rows % a.beams == 0 && a.max_tail <= 64) return launch_dec_beam4(a, st);
    return launch_dec_t<bf16_t, 64, 4>(a, st);
  }
  // greedy rows: TWO waves per workgroup for full launches (a wave walks every second slot of 64-key tiles; up to four workgroups per CU): 97 rows 28 -> 20 us,
  // head_dim 128 50 -> 45 us against eight waves, and the CU-time falls with the LDS footprint (round 5: headline 327 -> 330 utt/s, Llama-3.1-8B 164 -> 166).
  // The merge order of a row is canonical (decode_attn_kernel: slots, decode_merge_*: round 6), so the wave count may follow the launch: up to two
  // workgroups per CU take EIGHT waves -- a row's tiles in one pass: 10.1 -> 6.8 us at one row, 11.5 -> 7.4 at 8, 15.7 -> 12.4 at 48 (16 s clips); 97 rows: 20.1 vs 22.3
  static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  if (a.rows * a.n_kv_heads <= 2 * ncu) return a.head_dim == 64 ? launch_dec_t<bf16_t, 64, 8>(a, st) : launch_dec_t<bf16_t, 128, 8>(a, st);
  return a.head_dim == 64 ? launch_dec_t<bf16_t, 64, 2>(a, st) : launch_dec_t<bf16_t, 128, 2>(a, st);
}

}  // namespace oavsr
