// HF-equivalent beam search on device: restates transformers 4.43.1 `GenerationMixin._beam_search` +
// `BeamSearchScorer.process / finalize` + `BeamHypotheses.add / is_done` for do_sample=False, early_stopping=False,
// length_penalty=1.0, one EOS id, `inputs_embeds`-only prompts (decoder_prompt_len = 0) — the reference's call at
// Omni_AVSR/modeling_OmniAVSR.py:313-322 (num_beams=15 by default, eval_OmniAVSR.py:215-226).
//
//   beam_topk_kernel   one workgroup per row: fp32 log-softmax statistics + exact top-2K of the row by a 3-round radix
//                      select on the order-preserving integer image of the logits (4 streaming passes over the row in L2)
//   beam_step_kernel   one wave per utterance: merge the K rows' candidates -> global top-2K (ties: lower beam*V+token
//                      first), then the sequential HF bookkeeping (EOS candidates of rank < K become hypotheses, the next K
//                      non-EOS candidates become the beams, `is_done` heuristic), then reorders token histories and the
//                      KV ancestry table (beam b's tail position t lives in row anc[b][t]; no KV bytes are moved)
//   beam_finalize_kernel  open beams are added, best hypothesis per utterance is written (EOS appended when there is room)
// Hypothesis scores are kept in fp64 exactly like the Python floats of the reference implementation.
#include <stdlib.h>
#include "kernels.h"

namespace oavsr {
namespace {

constexpr int kMaxK = 32;          // beams
constexpr int kMax2K = 64;
constexpr int kTopkThreads = 1024;

__device__ __forceinline__ unsigned key_of(float f) {
  unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float val_of(unsigned k) {
  unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

// finds digit d (from the top) such that  count(keys with digit > d) < need <= count(digit >= d); returns d, updates need
__device__ int pick_bin(const int* hist, int nbins, int& need, int* s_scratch) {
  // serial scan by thread 0 (nbins <= 2048): negligible next to the streaming passes
  if (threadIdx.x == 0) {
    int acc = 0, d = nbins - 1;
    for (; d >= 0; --d) {
      if (acc + hist[d] >= need) break;
      acc += hist[d];
    }
    if (d < 0) d = 0;
    s_scratch[0] = d; s_scratch[1] = need - acc;
  }
  __syncthreads();
  int d = s_scratch[0];
  need = s_scratch[1];
  __syncthreads();
  return d;
}

__global__ __launch_bounds__(kTopkThreads) void beam_topk_exact_kernel(const float* __restrict__ logits, int64_t ld, int V, int twoK,
                                                                        const float* __restrict__ beam_scores, float* __restrict__ cand_val,
                                                                        int* __restrict__ cand_tok, const int* __restrict__ need_exact) {
  if (need_exact && !need_exact[blockIdx.x]) return;     // fallback of beam_topk_fast_kernel: only rows it could not certify
  __shared__ int hist[2048];
  __shared__ float red[kTopkThreads / 64 * 2];
  __shared__ int scratch[4];
  __shared__ unsigned sel_key[kMax2K];
  __shared__ int sel_idx[kMax2K];
  __shared__ int sel_n;
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (int64_t)row * ld;
  // ---- pass 1: max, then sum exp (two sweeps; the row stays in L2) + histogram of the top 11 key bits ----
  float m = -INFINITY;
  for (int i = tid; i < 2048; i += kTopkThreads) hist[i] = 0;
  __syncthreads();
  for (int i = tid; i < V; i += kTopkThreads) {
    float v = x[i];
    m = fmaxf(m, v);
    atomicAdd(&hist[key_of(v) >> 21], 1);
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = red[0];
  for (int w = 1; w < kTopkThreads / 64; ++w) m = fmaxf(m, red[w]);
  __syncthreads();
  float s = 0.f;
  for (int i = tid; i < V; i += kTopkThreads) s += expf(x[i] - m);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  s = 0.f;
  for (int w = 0; w < kTopkThreads / 64; ++w) s += red[w];
  const float logz = logf(s);
  // ---- radix select of the twoK-th largest key: 11 + 11 + 10 bits ----
  int need = twoK < V ? twoK : V;
  const int d1 = pick_bin(hist, 2048, need, scratch);
  for (int i = tid; i < 2048; i += kTopkThreads) hist[i] = 0;
  __syncthreads();
  for (int i = tid; i < V; i += kTopkThreads) {
    unsigned k = key_of(x[i]);
    if ((int)(k >> 21) == d1) atomicAdd(&hist[(k >> 10) & 2047], 1);
  }
  __syncthreads();
  const int d2 = pick_bin(hist, 2048, need, scratch);
  for (int i = tid; i < 1024; i += kTopkThreads) hist[i] = 0;
  __syncthreads();
  const unsigned pre = ((unsigned)d1 << 21) | ((unsigned)d2 << 10);
  for (int i = tid; i < V; i += kTopkThreads) {
    unsigned k = key_of(x[i]);
    if ((k & 0xfffffc00u) == pre) atomicAdd(&hist[k & 1023], 1);
  }
  __syncthreads();
  const int d3 = pick_bin(hist, 1024, need, scratch);
  const unsigned thr = pre | (unsigned)d3;      // exact key of the twoK-th largest element; `need` of the == thr ones are kept
  // ---- collect: everything above the threshold, then the lowest-index elements equal to it ----
  if (tid == 0) sel_n = 0;
  __syncthreads();
  for (int i = tid; i < V; i += kTopkThreads) {
    unsigned k = key_of(x[i]);
    if (k > thr) { int p = atomicAdd(&sel_n, 1); if (p < kMax2K) { sel_key[p] = k; sel_idx[p] = i; } }
  }
  __syncthreads();
  int last = -1;
  for (int r = 0; r < need; ++r) {              // usually a single round
    int best = 0x7fffffff;
    for (int i = tid; i < V; i += kTopkThreads)
      if (i > last && key_of(x[i]) == thr && i < best) best = i;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int ob = __shfl_xor(best, o, 64); best = ob < best ? ob : best; }
    if ((tid & 63) == 0) hist[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
      int b = hist[0];
      for (int w = 1; w < kTopkThreads / 64; ++w) b = hist[w] < b ? hist[w] : b;
      int p = sel_n; sel_key[p] = thr; sel_idx[p] = b; sel_n = p + 1; scratch[2] = b;
    }
    __syncthreads();
    last = scratch[2];
    __syncthreads();
  }
  // ---- rank sort (value desc, index asc) and write  logp + beam_score ----
  const int n = sel_n;
  const float bs = beam_scores ? beam_scores[row] : 0.f;
  if (tid < n) {
    unsigned k = sel_key[tid]; int idx = sel_idx[tid];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (sel_key[j] > k) || (sel_key[j] == k && sel_idx[j] < idx);
    cand_val[(int64_t)row * twoK + rank] = ((val_of(k) - m) - logz) + bs;
    cand_tok[(int64_t)row * twoK + rank] = idx;
  }
}

// ---- fast path of the row top-2K (same outputs, bit for bit): two sweeps, no histogram atomics, no per-element insertion ----
// Sweep 1 is a plain max per thread (round 5 kept a sorted top-6 per thread in registers: with 64 lanes some lane inserts on nearly every
// element, so every wave ran the ~40-instruction insertion path all the time -- half of the kernel's 109 us at 120 x 128 261).  The 2K-th largest
// of the 1024 thread maxima is a LOWER BOUND tau of the row's 2K-th largest element (those 2K maxima are 2K distinct elements >= tau), and a tight
// one (~2K .. 1.5 x 2K elements of the row reach it).  Sweep 2 is the exact kernel's sum of exp (same element -> thread mapping, same reduction
// order => identical logZ bits) and, on the way, appends every element >= tau to a candidate list in LDS; the candidates are rank-sorted with the
// tie rule (value desc, index asc).  EXACT for every row whose candidates fit the list (kCandMax); otherwise -- a row with >= kCandMax elements
// tied at its 2K-th value -- need_exact[row] is set and beam_topk_exact_kernel redoes the row.
constexpr int kCandMax = 1024;

struct Cand { float v; int i; };
__device__ __forceinline__ bool beats(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

__global__ __launch_bounds__(kTopkThreads) void beam_topk_fast_kernel(const float* __restrict__ logits, int64_t ld, int V, int twoK,
                                                                       const float* __restrict__ beam_scores, float* __restrict__ cand_val,
                                                                       int* __restrict__ cand_tok, int* __restrict__ need_exact) {
  constexpr int NW = kTopkThreads / 64;
  __shared__ float red[NW];
  __shared__ float fin_v[NW * 64];                  // per wave: its 64 thread maxima, sorted descending
  __shared__ float c_v[kCandMax];
  __shared__ int c_i[kCandMax];
  __shared__ int c_n;
  __shared__ float s_tau;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (int64_t)row * ld;
  if (tid == 0) c_n = 0;
  const int need = twoK < V ? twoK : V;
  // ---- sweep 1: the thread's max ----
  float tm = -INFINITY;
  constexpr int UL = 16;                          // loads in flight per thread (the row is streamed from HBM / L2)
  for (int i0 = tid; i0 < V; i0 += kTopkThreads * UL) {
    float buf[UL];
#pragma unroll
    for (int u = 0; u < UL; ++u) { const int i = i0 + u * kTopkThreads; buf[u] = i < V ? x[i] : -INFINITY; }
#pragma unroll
    for (int u = 0; u < UL; ++u) tm = fmaxf(tm, buf[u]);
  }
  float m = wave_max(tm);
  if (lane == 0) red[wave] = m;
  // the wave's 64 maxima sorted descending (bitonic network over the lanes)
  float sv = tm;
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float ov = __shfl_xor(sv, j, 64);
      const bool up = ((lane & k) == 0);            // this k-block sorts descending (up) or ascending
      const bool lower = ((lane & j) == 0);
      const float hi = fmaxf(sv, ov), lo = fminf(sv, ov);
      sv = (up == lower) ? hi : lo;
    }
  fin_v[wave * 64 + lane] = sv;
  __syncthreads();
  m = red[0];
  for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
  // ---- wave 0: the need-th largest of the 1024 maxima = merge of the NW sorted lists (lane w < NW walks list w) ----
  if (wave == 0) {
    int pos = 0;
    float tau = -INFINITY;
    for (int r = 0; r < need; ++r) {
      const float bv = (lane < NW && pos < 64) ? fin_v[lane * 64 + pos] : -INFINITY;
      float wv = bv; int wl = lane;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(wv, o, 64); const int ol = __shfl_xor(wl, o, 64);
        if (ov > wv || (ov == wv && ol < wl)) { wv = ov; wl = ol; }
      }
      tau = wv;
      if (lane == wl) ++pos;
    }
    if (lane == 0) s_tau = tau;                     // (-inf when the row has fewer than `need` non-empty threads: everything is a candidate)
  }
  __syncthreads();
  const float tau = s_tau;
  // ---- sweep 2: sum of exp, identical order to the exact kernel; candidates >= tau on the way ----
  float s = 0.f;
  for (int i0 = tid; i0 < V; i0 += kTopkThreads * UL) {    // same element order per thread as the exact kernel
    float buf[UL];
#pragma unroll
    for (int u = 0; u < UL; ++u) { const int i = i0 + u * kTopkThreads; buf[u] = i < V ? x[i] : 0.f; }
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int i = i0 + u * kTopkThreads;
      if (i < V) {
        s += expf(buf[u] - m);
        if (buf[u] >= tau) { const int p = atomicAdd(&c_n, 1); if (p < kCandMax) { c_v[p] = buf[u]; c_i[p] = i; } }
      }
    }
  }
  s = wave_sum(s);
  __syncthreads();                                   // (red[] is read above by every thread before it is rewritten)
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = 0.f;
  for (int w = 0; w < NW; ++w) s += red[w];
  const float logz = logf(s);
  const int n = c_n;
  if (tid == 0) need_exact[row] = n > kCandMax ? 1 : 0;
  if (n > kCandMax) return;
  // ---- rank sort of the candidates (value desc, index asc); the first `need` ranks are the row's top 2K ----
  const float bs = beam_scores ? beam_scores[row] : 0.f;
  if (tid < n) {
    const float v = c_v[tid]; const int idx = c_i[tid];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += beats(c_v[j], c_i[j], v, idx) ? 1 : 0;
    if (rank < need) {
      cand_val[(int64_t)row * twoK + rank] = ((v - m) - logz) + bs;
      cand_tok[(int64_t)row * twoK + rank] = idx;
    }
  }
}

// One workgroup per utterance.  The utterance's top 2K of its K x 2K row candidates by RANK: every candidate counts the candidates that beat it
// (value desc, then flat index beam * V + token asc -- torch.topk's order over the flattened [K V] scores; flat indices are distinct, so ranks
// are too), 2 048 candidates at most on 512 threads (round 5: 2K rounds of a 64-lane arg-max with a barrier each, 70 us per step at K = 15).
constexpr int kStepThreads = 512;
__global__ __launch_bounds__(kStepThreads) void beam_step_kernel(BeamArgs a, int Kr /* rows per utterance holding candidates: 1 at step 0, else K */,
                                                                  const int* seq_old, const int* anc_old, int* seq_new, int* anc_new) {
  __shared__ float cv[kMaxK * kMax2K];
  __shared__ int cflat_b[kMaxK * kMax2K];
  __shared__ int ctok[kMaxK * kMax2K];
  __shared__ long cflat[kMaxK * kMax2K];         // beam * V + token: torch.topk's flat index
  __shared__ float top_v[kMax2K];
  __shared__ int top_b[kMax2K], top_t[kMax2K];
  __shared__ int nb_parent[kMaxK], nb_tok[kMaxK];
  __shared__ float nb_score[kMaxK];
  const int u = blockIdx.x, lane = threadIdx.x;
  const int K = a.K, twoK = 2 * a.K, V = a.V;
  const int cur_len = *a.step;                 // tokens generated so far (== input_ids.shape[-1] in HF's process())
  const int row0 = u * K;
  const int n = Kr * twoK;
  for (int i = lane; i < n; i += kStepThreads) {
    int b = i / twoK, j = i - b * twoK;
    int64_t src = (int64_t)((Kr == 1 ? u : row0 + b)) * twoK + j;
    const int tok = a.cand_tok[src];
    cv[i] = a.cand_val[src]; ctok[i] = tok; cflat_b[i] = b; cflat[i] = (long)b * V + tok;
  }
  __syncthreads();
  for (int i = lane; i < n; i += kStepThreads) {
    const float v = cv[i];
    const long flat = cflat[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float w = cv[j];
      rank += (w > v || (w == v && cflat[j] < flat)) ? 1 : 0;
    }
    if (rank < twoK) { top_v[rank] = v; top_b[rank] = cflat_b[i]; top_t[rank] = ctok[i]; }
  }
  __syncthreads();
  // ---- BeamSearchScorer.process for this batch element (sequential, lane 0) ----
  if (lane == 0) {
    int nnew = 0;
    const bool was_done = a.done[u] != 0;
    if (!was_done) {
      int hn = a.hyp_n[u];
      double* hs = a.hyp_score + (int64_t)u * K;
      int* hl = a.hyp_len + (int64_t)u * K;
      int* ht = a.hyp_tok + (int64_t)u * K * a.max_new;
      for (int r = 0; r < twoK && nnew < K; ++r) {
        const int tok = top_t[r], pb = top_b[r];
        const double sc = (double)top_v[r];
        if (tok == a.eos) {
          if (r >= K) continue;                                 // is_beam_token_worse_than_top_num_beams
          const double score = sc / (double)(cur_len + 1);      // generated_len = cur_len + 1, length_penalty 1.0
          double worst = 1e300;
          for (int i = 0; i < hn; ++i) worst = hs[i] < worst ? hs[i] : worst;
          if (hn < K || score > worst) {
            int slot = hn;
            if (hn == K) {                                      // list would exceed K: drop the worst (first minimum)
              int wi = 0;
              for (int i = 1; i < hn; ++i) if (hs[i] < hs[wi]) wi = i;
              for (int i = wi; i + 1 < hn; ++i) {               // keep insertion order of the survivors
                hs[i] = hs[i + 1]; hl[i] = hl[i + 1];
                for (int t = 0; t < a.max_new; ++t) ht[i * a.max_new + t] = ht[(i + 1) * a.max_new + t];
              }
              slot = hn - 1;
            } else {
              hn += 1;
            }
            hs[slot] = score; hl[slot] = cur_len;
            const int* sp = seq_old + (int64_t)(row0 + (Kr == 1 ? 0 : pb)) * a.max_new;
            for (int t = 0; t < cur_len; ++t) ht[slot * a.max_new + t] = sp[t];
          }
        } else {
          nb_parent[nnew] = pb; nb_tok[nnew] = tok; nb_score[nnew] = top_v[r]; ++nnew;
        }
      }
      a.hyp_n[u] = hn;
      // BeamHypotheses.is_done (early_stopping=False): worst_score >= best_sum_logprobs / (cur_len + 1)
      if (hn >= K) {
        double worst = 1e300;
        for (int i = 0; i < hn; ++i) worst = hs[i] < worst ? hs[i] : worst;
        const double highest = (double)top_v[0] / (double)(cur_len + 1);
        if (worst >= highest) a.done[u] = 1;
      }
    }
    for (; nnew < K; ++nnew) { nb_parent[nnew] = 0; nb_tok[nnew] = a.pad; nb_score[nnew] = 0.f; }   // done batch: pad beams
  }
  __syncthreads();
  // ---- reorder histories / ancestry, publish the next step's inputs ----
  for (int i = lane; i < K * cur_len; i += kStepThreads) {
    const int b = i / cur_len, t = i - b * cur_len;
    const int prow = row0 + (Kr == 1 ? 0 : nb_parent[b]);
    const int row = row0 + b;
    seq_new[(int64_t)row * a.max_new + t] = seq_old[(int64_t)prow * a.max_new + t];
    anc_new[(int64_t)row * a.max_new + t] = anc_old[(int64_t)prow * a.max_new + t];
  }
  if (lane < K) {
    const int b = lane, row = row0 + b;
    if (cur_len < a.max_new) { seq_new[(int64_t)row * a.max_new + cur_len] = nb_tok[b]; anc_new[(int64_t)row * a.max_new + cur_len] = row; }
    a.beam_scores[row] = nb_score[b];
    a.next_tok[row] = nb_tok[b];
    a.pos[row] = a.prefix_len[u] + cur_len;      // absolute position of the token just chosen
    a.cpos[row] = cur_len;                       // its tail-cache index
  }
  if (a.rope_cs) {      // the K rows of an utterance share the position: same expression as the attention kernels' inline path => same bits
    const float pos = (float)(a.prefix_len[u] + cur_len);
    for (int i = lane; i < K * a.rope_half; i += kStepThreads) {
      const int b = i / a.rope_half, d = i - b * a.rope_half;
      float sn, cs;
      sincosf(pos * a.inv_freq[d], &sn, &cs);
      a.rope_cs[(int64_t)(row0 + b) * a.rope_half + d] = make_float2(cs, sn);
    }
  }
}

__global__ void beam_bump_step_kernel(int* step) { *step += 1; }

__global__ __launch_bounds__(64) void beam_finalize_kernel(BeamArgs a, const int* seqs) {
  const int u = blockIdx.x;
  if (threadIdx.x != 0) return;
  const int K = a.K;
  const int cur_len = *a.step;
  int hn = a.hyp_n[u];
  double* hs = a.hyp_score + (int64_t)u * K;
  int* hl = a.hyp_len + (int64_t)u * K;
  int* ht = a.hyp_tok + (int64_t)u * K * a.max_new;
  if (!a.done[u]) {      // finalize(): every open beam becomes a hypothesis (same add() rule)
    for (int b = 0; b < K; ++b) {
      const double score = (double)a.beam_scores[u * K + b] / (double)cur_len;
      double worst = 1e300;
      for (int i = 0; i < hn; ++i) worst = hs[i] < worst ? hs[i] : worst;
      if (hn < K || score > worst) {
        int slot = hn;
        if (hn == K) {
          int wi = 0;
          for (int i = 1; i < hn; ++i) if (hs[i] < hs[wi]) wi = i;
          for (int i = wi; i + 1 < hn; ++i) {
            hs[i] = hs[i + 1]; hl[i] = hl[i + 1];
            for (int t = 0; t < a.max_new; ++t) ht[i * a.max_new + t] = ht[(i + 1) * a.max_new + t];
          }
          slot = hn - 1;
        } else {
          hn += 1;
        }
        hs[slot] = score; hl[slot] = cur_len;
        for (int t = 0; t < cur_len; ++t) ht[slot * a.max_new + t] = seqs[(int64_t)(u * K + b) * a.max_new + t];
      }
    }
  }
  // best = sorted(beams, key=score)[-1]  (stable sort: among equal scores the last inserted wins)
  int best = 0;
  for (int i = 1; i < hn; ++i) if (hs[i] >= hs[best]) best = i;
  int len = hl[best];
  int* out = a.out_ids + (int64_t)u * a.max_new;
  for (int t = 0; t < a.max_new; ++t) out[t] = t < len ? ht[best * a.max_new + t] : a.pad;
  if (len < a.max_new) { out[len] = a.eos; len += 1; }
  a.out_len[u] = len;
}

}  // namespace

int launch_beam_topk(const float* logits, int64_t ld, int rows, int V, int K, const float* beam_scores, float* cand_val, int* cand_tok,
                     int* need_exact, hipStream_t st) {
  OAVSR_REQUIRE(K >= 1 && K <= kMaxK, "beam search: num_beams=%d unsupported (max %d)", K, kMaxK);
  if (rows <= 0) return 0;
  const bool fast = true;      // threshold + candidate list first; rows whose candidates overflow the list fall through to the exact radix select
  if (fast && need_exact)
    hipLaunchKernelGGL(beam_topk_fast_kernel, dim3(rows), dim3(kTopkThreads), 0, st, logits, ld, V, 2 * K, beam_scores, cand_val, cand_tok, need_exact);
  hipLaunchKernelGGL(beam_topk_exact_kernel, dim3(rows), dim3(kTopkThreads), 0, st, logits, ld, V, 2 * K, beam_scores, cand_val, cand_tok,
                     (fast && need_exact) ? need_exact : nullptr);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_beam_step(const BeamArgs& a, int first, const int* seq_old, const int* anc_old, int* seq_new, int* anc_new, hipStream_t st) {
  OAVSR_REQUIRE(a.K >= 1 && a.K <= kMaxK, "beam search: num_beams=%d unsupported (max %d)", a.K, kMaxK);
  hipLaunchKernelGGL(beam_step_kernel, dim3(a.B), dim3(kStepThreads), 0, st, a, first ? 1 : a.K, seq_old, anc_old, seq_new, anc_new);
  hipLaunchKernelGGL(beam_bump_step_kernel, dim3(1), dim3(1), 0, st, a.step);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_beam_finalize(const BeamArgs& a, const int* seqs, hipStream_t st) {
  hipLaunchKernelGGL(beam_finalize_kernel, dim3(a.B), dim3(64), 0, st, a, seqs);
  OAVSR_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace oavsr
