// log_softmax + CTC loss + gradient (reference model.py:246 log_softmax(-1); model.py:203,248
// CTCLoss(blank, reduction='sum', zero_infinity=True) -> torch native _ctc_loss).
//
// Log-space alpha/beta recursion over the extended label sequence l' (2S+1 states, blank-interleaved).  With both
// alpha_t(s) and beta_t(s) including y_t(l'_s),
//   d nll / d lp[t][c] = - sum_{s: l'_s = c} exp(alpha_t(s) + beta_t(s) - ll) / y_t(c)
// (every term exp(alpha+beta-ll) <= 1, so the total log-likelihood ll is a safe common shift), followed by the
// log_softmax backward  dlogit = g - softmax * sum_c g.  Infeasible samples (ll = -inf): loss 0, gradient 0.
//
// Four launches on the caller's stream, split by what bounds each stage:
//   k_ctc_logsoftmax  one thread per frame; bytes: logits once -> log-prob scratch [N][Tp][CP] (CP = 32 for <= 32 classes).
//   k_ctc_recursion_pairs (round 6, the default)  grid (N, 2): the alpha and the beta recursion of a sample are independent
//                     workgroups.  T' dependent steps, pure latency: the rows live in REGISTERS as (blank, label) state pairs on
//                     overlapping wave tiles, one wave shift per step, halo exchange through LDS every 16 steps (see the kernel).
//   k_ctc_recursion   (rounds 2-5; targets beyond the pair tiles, and `recursion` = 1) the same with the rows in LDS: 4 waves, the
//                     step is branch-free (operand addresses and transition masks are fixed per thread before the loop), the
//                     frames' log-probs are staged in LDS a chunk ahead, and the per-step barrier drains the LDS counter only --
//                     the alpha/beta row stores stay in flight.
//   k_ctc_gradient    frame-parallel, grid (T'/16, N) x one wave per frame: the alpha/beta rows (2 x 4 x T' x L bytes per
//                     sample, beyond one XCD's L2 for long clips) stream back once at full-chip parallelism.
//   k_sum_small       loss = sum of the per-sample nll.
#include <math.h>

#include "ds2_common.h"

namespace {

constexpr int REC_THREADS = 256;  // one recursion: <= 4 states per thread (fast path) for targets up to 511 labels
// CP (template parameter of the kernels below): padded class stride of the log-prob scratch -- 32, 64, 128 or 256, the smallest
// that holds the model's classes (the reference takes any labels file, model.py:154-155; the English set has 29)
constexpr int LP_CHUNK = 8;       // recursion steps whose log-prob rows are staged in LDS together
constexpr int NS_FAST = 4;        // extended-label states per thread the branch-free recursion keeps in registers
constexpr int GRAD_WAVES = 4;     // waves (= frames in flight) per workgroup of the gradient pass
constexpr int GRAD_FRAMES = 16;   // frames per workgroup of the gradient pass

// workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): outstanding global stores and
// prefetch loads stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// log(exp a + exp b + exp c), branch-free.  v_exp_f32 / v_log_f32 (1 ulp in the base-2 domain) instead of the full-precision
// libm sequences: the argument of the log is in [1, 3] (or 0), where the absolute error of the bare instruction times ln 2 is
// < 2e-7 -- far below the fp32 resolution of the alpha/beta values themselves (magnitude ~3 T').  Round 6: `__logf` still expanded
// to a 12-instruction sequence (denormal pre-scaling the argument never needs + a compensated product with ln 2); the bare
// v_log_f32 is what the comment always promised.  All three -inf: exp(-inf) = 0, log2(0) = -inf.
// ms + ln(x), the product with ln 2 fused into the sum in EVERY kernel (an explicit fma: no context-dependent contraction)
__device__ __forceinline__ float add_ln_1to3(float ms, float x) { return __builtin_fmaf(__builtin_amdgcn_logf(x), 0.69314718056f, ms); }
// The shift ms is the largest term, held above -FLT_MAX so that "all -inf" needs no select: -inf - ms = -inf, sum 0, result -inf.
__device__ __forceinline__ float lse3(float a, float b, float c) {
  float m;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));      // (the builtin pair canonicalises each operand first)
  const float ms = __builtin_fmaxf(m, -3.4028234e38f);
  return add_ln_1to3(ms, __expf(a - ms) + __expf(b - ms) + __expf(c - ms));
}
// the same with c = -inf (exp(-inf) = 0 and x + 0 = x: bit-identical to lse3(a, b, -inf), one exponential less)
__device__ __forceinline__ float lse2(float a, float b) {
  const float ms = __builtin_fmaxf(__builtin_fmaxf(a, b), -3.4028234e38f);
  return add_ln_1to3(ms, __expf(a - ms) + __expf(b - ms));
}

template <int CP>
__global__ void __launch_bounds__(256) k_ctc_logsoftmax(const float* __restrict__ logits, long ldl,
                                                        const int* __restrict__ in_len, int Tp, int N, int C,
                                                        float* __restrict__ ws_lp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;     // frame row t*N + n
  if (i >= (long)Tp * N) return;
  const int t = (int)(i / N), n = (int)(i % N);
  if (t >= in_len[n]) return;
  const float* x = logits + i * ldl;
  float* lp = ws_lp + ((long)n * Tp + t) * CP;
  if constexpr (CP == 32) {
    float v[CP];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      v[c] = x[c < C ? c : 0];
      if (c < C) m = fmaxf(m, v[c]);
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c)
      if (c < C) sum += expf(v[c] - m);
    const float lz = m + logf(sum);
#pragma unroll
    for (int c = 0; c < CP; ++c)
      if (c < C) lp[c] = v[c] - lz;
  } else {                                             // larger label sets: three passes over the row (L1 / L2 resident)
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, x[c]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(x[c] - m);
    const float lz = m + logf(sum);
    for (int c = 0; c < C; ++c) lp[c] = x[c] - lz;
  }
}

// blockIdx.x = sample, blockIdx.y = 0: alpha (t = 0 .. Ti-1), 1: beta (t = Ti-1 .. 0)
template <int CP>
__global__ void __launch_bounds__(REC_THREADS) k_ctc_recursion(const int* __restrict__ targets, const int* __restrict__ toff,
                                                               const int* __restrict__ in_len, const int* __restrict__ tg_len,
                                                               int Tp, int blank, int Lmax, float* __restrict__ nll_out,
                                                               const float* __restrict__ ws_lp, float* __restrict__ ws_alpha,
                                                               float* __restrict__ ws_beta, float* __restrict__ ws_ll) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* row0 = reinterpret_cast<float*>(smem);      // previous / current rows (double buffer)   [Lmax] each
  float* row1 = row0 + Lmax;
  int* ext = reinterpret_cast<int*>(row1 + Lmax);    // [Lmax]
  float* lpl = reinterpret_cast<float*>(ext + Lmax); // [2 chunk buffers][LP_CHUNK][CP]: staged log-prob rows
  const int tid = threadIdx.x;
  const int n = blockIdx.x;
  const bool is_beta = blockIdx.y != 0;
  int Ti = in_len[n];
  if (Ti > Tp) Ti = Tp;
  const int S = tg_len[n];
  const int L = 2 * S + 1;
  if (Ti <= 0) {
    if (!is_beta && tid == 0) {
      nll_out[n] = 0.f;
      ws_ll[n] = INFINITY;                           // sentinel: no gradient for this sample
    }
    return;
  }
  const int* tg = targets + toff[n];
  const float* lp = ws_lp + (long)n * Tp * CP;                               // [Tp][CP] of this sample
  float* dst = (is_beta ? ws_beta : ws_alpha) + (long)n * Tp * Lmax;         // [Tp][Lmax]
  for (int s = tid; s < L; s += REC_THREADS) ext[s] = (s & 1) ? tg[s >> 1] : blank;
  constexpr int PE = LP_CHUNK * CP / REC_THREADS;    // elements of a chunk [LP_CHUNK][CP] each thread stages (1 for CP = 32)
  static_assert(PE * REC_THREADS == LP_CHUNK * CP, "whole prefetch elements per thread");
#pragma unroll
  for (int j = 0; j < PE; ++j) {
    const int e = tid + j * REC_THREADS, prow = e / CP, pcol = e % CP;
    const int i = prow < Ti ? prow : Ti - 1;
    lpl[prow * CP + pcol] = lp[(long)(is_beta ? Ti - 1 - i : i) * CP + pcol];
  }
  __syncthreads();
  // per-thread state tables of the branch-free step.  States past L are clamped to L-1: those lanes recompute and rewrite
  // the last state's value (same value, same address) instead of branching.  Masks are additive: 0 = transition allowed,
  // -inf = not.
  const int ns = (L + REC_THREADS - 1) / REC_THREADS;
  const bool fast = ns <= NS_FAST;
  int st_s[NS_FAST], st_e[NS_FAST], st_i1[NS_FAST], st_i2[NS_FAST];
  float st_m0[NS_FAST], st_m1[NS_FAST], st_m2[NS_FAST];
#pragma unroll
  for (int q = 0; q < NS_FAST; ++q) {
    int sq = tid + q * REC_THREADS;
    if (sq > L - 1) sq = L - 1;
    const int e = ext[sq];
    bool ok1, ok2;
    int j1, j2;
    if (!is_beta) {
      j1 = sq >= 1 ? sq - 1 : 0;
      j2 = sq >= 2 ? sq - 2 : 0;
      ok1 = sq >= 1;
      ok2 = sq >= 2 && e != blank && e != ext[j2];
      st_m0[q] = sq <= 1 ? 0.f : -INFINITY;          // alpha_0: the first blank and the first label
    } else {
      j1 = sq + 1 < L ? sq + 1 : L - 1;
      j2 = sq + 2 < L ? sq + 2 : L - 1;
      ok1 = sq + 1 < L;
      ok2 = sq + 2 < L && ext[j2] != blank && ext[j2] != e;
      st_m0[q] = sq >= L - 2 ? 0.f : -INFINITY;      // beta_{T-1}: the last blank and the last label
    }
    st_s[q] = sq;
    st_e[q] = e;
    st_i1[q] = j1;
    st_i2[q] = j2;
    st_m1[q] = ok1 ? 0.f : -INFINITY;
    st_m2[q] = ok2 ? 0.f : -INFINITY;
  }
  float* prev = row0;
  float* cur = row1;
  for (int i0 = 0; i0 < Ti; i0 += LP_CHUNK) {
    const float* lpc = lpl + ((i0 / LP_CHUNK) & 1) * LP_CHUNK * CP;
    float nxt[PE];                                   // the rows of the steps that go into the other chunk buffer
#pragma unroll
    for (int j = 0; j < PE; ++j) {
      const int e = tid + j * REC_THREADS;
      int pi = i0 + LP_CHUNK + e / CP;
      if (pi > Ti - 1) pi = Ti - 1;
      nxt[j] = lp[(long)(is_beta ? Ti - 1 - pi : pi) * CP + e % CP];
    }
    const int kend = Ti - i0 < LP_CHUNK ? Ti - i0 : LP_CHUNK;
    for (int k = 0; k < kend; ++k) {
      const int i = i0 + k;
      const int t = is_beta ? Ti - 1 - i : i;
      const float* lpt = lpc + k * CP;
      if (fast) {
        float p0[NS_FAST], p1[NS_FAST], p2[NS_FAST], pe[NS_FAST];
#pragma unroll
        for (int q = 0; q < NS_FAST; ++q)
          if (q < ns) {
            p0[q] = prev[st_s[q]];
            p1[q] = prev[st_i1[q]];
            p2[q] = prev[st_i2[q]];
            pe[q] = lpt[st_e[q]];
          }
#pragma unroll
        for (int q = 0; q < NS_FAST; ++q)
          if (q < ns) {
            const float rec = lse3(p0[q], p1[q] + st_m1[q], p2[q] + st_m2[q]);
            const float v = (i == 0 ? st_m0[q] : rec) + pe[q];
            cur[st_s[q]] = v;
            dst[(long)t * Lmax + st_s[q]] = v;
          }
      } else {
        // very long targets (> 511 labels): strided states, same arithmetic
        for (int s = tid; s < L; s += REC_THREADS) {
          const int e = ext[s];
          float v;
          if (i == 0) {
            const bool entry = is_beta ? s >= L - 2 : s <= 1;
            v = entry ? lpt[e] : -INFINITY;
          } else if (!is_beta) {
            const float a0 = prev[s];
            const float a1 = s >= 1 ? prev[s - 1] : -INFINITY;
            const float a2 = (s >= 2 && e != blank && e != ext[s - 2]) ? prev[s - 2] : -INFINITY;
            v = lse3(a0, a1, a2) + lpt[e];
          } else {
            const float b0 = prev[s];
            const float b1 = s + 1 < L ? prev[s + 1] : -INFINITY;
            const float b2 = (s + 2 < L && ext[s + 2] != blank && ext[s + 2] != e) ? prev[s + 2] : -INFINITY;
            v = lse3(b0, b1, b2) + lpt[e];
          }
          cur[s] = v;
          dst[(long)t * Lmax + s] = v;
        }
      }
      lds_barrier();
      float* tmp = prev;
      prev = cur;
      cur = tmp;
    }
#pragma unroll
    for (int j = 0; j < PE; ++j) lpl[(((i0 / LP_CHUNK) + 1) & 1) * LP_CHUNK * CP + tid + j * REC_THREADS] = nxt[j];
    lds_barrier();
  }
  if (!is_beta && tid == 0) {
    // `prev` holds alpha_{Ti-1}
    const float l1 = prev[L - 1];
    const float l2 = L > 1 ? prev[L - 2] : -INFINITY;
    const float ll = lse3(l1, l2, -INFINITY);
    // zero_infinity (reference model.py:203): ONLY an infinite loss -- an infeasible alignment -- is zeroed, loss and
    // gradient.  A NaN (poisoned logits, e.g. from a timed-out recurrent sweep) stays NaN in the loss and in the gradient,
    // as torch's ctc_loss does: it must never look like a sample that contributes nothing.
    const bool feasible = L <= 2 * Ti + 1 && ll != -INFINITY;
    nll_out[n] = feasible ? -ll : 0.f;
    ws_ll[n] = feasible ? ll : INFINITY;
  }
}

// The same recursion with ONE WAVE per (sample, direction) -- the rounds 3-5 fast path for 32 classes and SHORT targets (`recursion` = 2 / 3 of ds2_ctc_loss_grad).
// The states of the extended label sequence live in registers, NS per lane, blocked (lane l owns states l*NS .. l*NS + NS - 1); the
// neighbours s-1 / s-2 (alpha) or s+1 / s+2 (beta) of a lane's first / last states come from the adjacent lane by a wave
// shift.  No LDS round trip of the rows and no workgroup barrier per step (the 4-wave kernel above spends most of its 0.48 us per
// step there); the log-prob rows are staged in LDS by the wave itself, LP_CHUNK steps ahead.  Same formulas, same values.
// lane i <- lane i - 1 / lane i + 1 of the whole wave in ONE VALU instruction (GFX9 DPP wave shifts; the lane without a source gets
// 0, which only ever meets a -inf transition mask)
__device__ __forceinline__ float wave_from_below(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_from_above(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

template <int NS>
__global__ void __launch_bounds__(64) k_ctc_recursion_wave(const int* __restrict__ targets, const int* __restrict__ toff,
                                                           const int* __restrict__ in_len, const int* __restrict__ tg_len,
                                                           int Tp, int blank, int Lmax, float* __restrict__ nll_out,
                                                           const float* __restrict__ ws_lp, float* __restrict__ ws_alpha,
                                                           float* __restrict__ ws_beta, float* __restrict__ ws_ll) {
  constexpr int CP = 32;
  constexpr int PE = LP_CHUNK * CP / 64;             // staged elements per lane and chunk
  __shared__ float lpl[2][LP_CHUNK * CP];
  __shared__ float fin[64 * NS];
  const int lane = threadIdx.x;
  const int n = blockIdx.x;
  const bool is_beta = blockIdx.y != 0;
  int Ti = in_len[n];
  if (Ti > Tp) Ti = Tp;
  const int S = tg_len[n];
  const int L = 2 * S + 1;
  if (Ti <= 0) {
    if (!is_beta && lane == 0) {
      nll_out[n] = 0.f;
      ws_ll[n] = INFINITY;
    }
    return;
  }
  const int* tg = targets + toff[n];
  const float* lp = ws_lp + (long)n * Tp * CP;
  float* dst = (is_beta ? ws_beta : ws_alpha) + (long)n * Tp * Lmax;
  auto ext = [&](int s) { return (s & 1) ? tg[s >> 1] : blank; };
  int e_[NS];
  bool valid[NS];
  float m0[NS], m1[NS], m2[NS], prev[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int s = lane * NS + j;
    valid[j] = s < L;
    const int sc = valid[j] ? s : L - 1;
    const int e = ext(sc);
    e_[j] = e;
    bool ok1, ok2, entry;
    if (!is_beta) {
      ok1 = sc >= 1;
      ok2 = sc >= 2 && e != blank && e != ext(sc >= 2 ? sc - 2 : 0);
      entry = sc <= 1;
    } else {
      ok1 = sc + 1 < L;
      ok2 = sc + 2 < L && ext(sc + 2 < L ? sc + 2 : L - 1) != blank && ext(sc + 2 < L ? sc + 2 : L - 1) != e;
      entry = sc >= L - 2;
    }
    m0[j] = entry ? 0.f : -INFINITY;
    m1[j] = ok1 ? 0.f : -INFINITY;
    m2[j] = ok2 ? 0.f : -INFINITY;
    prev[j] = -INFINITY;
  }
  // stage the first chunk of log-prob rows
#pragma unroll
  for (int q = 0; q < PE; ++q) {
    const int el = lane + q * 64, prow = el / CP, pcol = el % CP;
    const int i = prow < Ti ? prow : Ti - 1;
    lpl[0][el] = lp[(long)(is_beta ? Ti - 1 - i : i) * CP + pcol];
  }
  __builtin_amdgcn_wave_barrier();
  for (int i0 = 0; i0 < Ti; i0 += LP_CHUNK) {
    const float* lpc = lpl[(i0 / LP_CHUNK) & 1];
    float nxt[PE];
#pragma unroll
    for (int q = 0; q < PE; ++q) {
      const int el = lane + q * 64;
      int pi = i0 + LP_CHUNK + el / CP;
      if (pi > Ti - 1) pi = Ti - 1;
      nxt[q] = lp[(long)(is_beta ? Ti - 1 - pi : pi) * CP + el % CP];
    }
    const int kend = Ti - i0 < LP_CHUNK ? Ti - i0 : LP_CHUNK;
    float pe_n[NS];                                    // emission terms one step ahead (their LDS latency under the step before)
#pragma unroll
    for (int j = 0; j < NS; ++j) pe_n[j] = lpc[e_[j]];
    for (int k = 0; k < kend; ++k) {
      const int i = i0 + k;
      const int t = is_beta ? Ti - 1 - i : i;
      float pe[NS], cur[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        pe[j] = pe_n[j];
        pe_n[j] = lpc[(k + 1 < LP_CHUNK ? k + 1 : k) * CP + e_[j]];
      }
      // the two states beyond this lane's block: alpha looks down (s-1, s-2 of the lane's first states), beta up
      float n1, n2;
      if (!is_beta) {
        n1 = wave_from_below(prev[NS - 1]);
        n2 = NS >= 2 ? wave_from_below(prev[NS >= 2 ? NS - 2 : 0]) : wave_from_below(n1);
      } else {
        n1 = wave_from_above(prev[0]);
        n2 = NS >= 2 ? wave_from_above(prev[NS >= 2 ? 1 : 0]) : wave_from_above(n1);
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        float a1, a2;
        if (!is_beta) {
          a1 = j >= 1 ? prev[j >= 1 ? j - 1 : 0] : n1;
          a2 = j >= 2 ? prev[j >= 2 ? j - 2 : 0] : (j == 1 ? n1 : n2);
        } else {
          a1 = j + 1 < NS ? prev[j + 1 < NS ? j + 1 : 0] : n1;
          a2 = j + 2 < NS ? prev[j + 2 < NS ? j + 2 : 0] : (j + 2 == NS ? n1 : n2);
        }
        const float rec = lse3(prev[j], a1 + m1[j], a2 + m2[j]);
        float v = (i == 0 ? m0[j] : rec) + pe[j];
        if (!valid[j]) v = -INFINITY;                  // states past L never feed a real one anything but -inf
        cur[j] = v;
        if (valid[j]) dst[(long)t * Lmax + lane * NS + j] = v;
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) prev[j] = cur[j];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < PE; ++q) lpl[((i0 / LP_CHUNK) + 1) & 1][lane + q * 64] = nxt[q];
    __builtin_amdgcn_wave_barrier();
  }
  if (!is_beta) {
#pragma unroll
    for (int j = 0; j < NS; ++j) fin[lane * NS + j] = prev[j];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      const float l1 = fin[L - 1];
      const float l2 = L > 1 ? fin[L - 2] : -INFINITY;
      const float ll = lse3(l1, l2, -INFINITY);
      const bool feasible = L <= 2 * Ti + 1 && ll != -INFINITY;      // zero_infinity: only an infinite loss is zeroed (see above)
      nll_out[n] = feasible ? -ll : 0.f;
      ws_ll[n] = feasible ? ll : INFINITY;
    }
  }
}

// Round 6: the recursion as overlapping tiles of state PAIRS, the default for targets of up to 783 labels and any class count.
//
// Pair P of a sample = the blank state 2P and the label state 2P + 1 of the extended sequence, in TRAVERSAL order: alpha walks the
// targets forwards and time forwards, beta walks both backwards -- in that order beta's rule (s, s+1, s+2) is alpha's (s, s-1, s-2),
// so one body serves both.  A lane keeps its pair in two registers; all it needs from outside is the label state of pair P - 1:
//     blank'  = lse(blank, label[P-1])                                  + lp[t][blank]
//     label'  = lse(label, blank, label[P-1] + (same label ? -inf : 0)) + lp[t][label]
// which is one wave shift.  No LDS round trip of the rows and no workgroup barrier per step (the four-wave kernel above spends most
// of its 0.5 us per step there); and, unlike the one-wave kernel, no limit of one SIMD: wave w holds pairs w (64 - K) ... + 63, i.e.
// its lowest K lanes repeat the top K pairs of the wave below (the halo).  A step without exchange invalidates one more halo lane
// from the bottom (lane j after j + 1 steps), so after K steps exactly the halo is stale: every K steps the waves swap the K
// boundary pairs through LDS (one barrier per K steps).  Only owned, existing states are stored.  The emission terms do not depend
// on the recursion: each lane gathers its two log-probs of a whole chunk of K steps from global memory one chunk ahead (the
// sample's log-prob rows are L2-resident), so there is no class-row staging and no class-count limit.  Same lse3 / lse2 terms in the
// same order as the kernels above: bit-identical rows.
template <int K, bool is_beta>
__device__ __forceinline__ void ctc_recursion_pairs(const int* __restrict__ targets, const int* __restrict__ toff,
                                                    const int* __restrict__ in_len, const int* __restrict__ tg_len, int Tp, int blank,
                                                    int LS, int CP, float* __restrict__ nll_out, const float* __restrict__ ws_lp,
                                                    float* __restrict__ ws_alpha, float* __restrict__ ws_beta,
                                                    float* __restrict__ ws_ll, float (*xch)[16][K][2], float* fin) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
  const int n = blockIdx.x;
  int Ti = in_len[n];
  if (Ti > Tp) Ti = Tp;
  const int S = tg_len[n];
  const int L = 2 * S + 1;
  if (Ti <= 0) {
    if (!is_beta && threadIdx.x == 0) {
      nll_out[n] = 0.f;
      ws_ll[n] = INFINITY;
    }
    return;
  }
  const int* tg = targets + toff[n];
  const int P = w * (64 - K) + lane;                 // pair index in traversal order
  const bool own = w == 0 || lane >= K;
  const int lab = P < S ? tg[is_beta ? S - 1 - P : P] : blank;
  const int labp = (P >= 1 && P - 1 < S) ? tg[is_beta ? S - P : P - 1] : -1;
  const float m_lo = P >= 1 ? 0.f : -INFINITY;                           // pair -1 does not exist
  const float m_skip = (P >= 1 && P < S && lab != labp) ? 0.f : -INFINITY;
  // A pair is ONE 8-byte store: rows have an even stride LS = 2 S_max + 2 and the beta rows start one float late (the host passes
  // ws_beta + 1), so alpha's {2P, 2P + 1} and beta's {L - 2 - 2P, L - 1 - 2P} are both 8-byte aligned; the label half of pair S (it
  // does not exist) lands in the row's pad slot.
  const bool st = own && 2 * P < L;
  const int sp = is_beta ? L - 2 - 2 * P : 2 * P;
  const float* lp = ws_lp + (long)n * Tp * CP;
  float* dst = (is_beta ? ws_beta : ws_alpha) + (long)n * Tp * LS;
  const long tstep = is_beta ? -1 : 1;
  const int t0 = is_beta ? Ti - 1 : 0;
  // step 0: the first blank and the first label
  float a0, a1;
  {
    const float* row = lp + (long)t0 * CP;
    a0 = (P == 0 ? 0.f : -INFINITY) + row[blank];
    a1 = (P == 0 ? 0.f : -INFINITY) + row[lab];
  }
  float2* drow = reinterpret_cast<float2*>(dst + (long)t0 * LS + sp);
  const long dstep = tstep * (LS / 2);
  if (st) *drow = is_beta ? float2{a1, a0} : float2{a0, a1};
  drow += dstep;
  auto step = [&](float pe0, float pe1) {
    const float below = wave_from_below(a1);
    const float v0 = lse2(a0, below + m_lo) + pe0;
    const float v1 = lse3(a1, a0, below + m_skip) + pe1;
    a0 = v0;
    a1 = v1;
    if (st) *drow = is_beta ? float2{v1, v0} : float2{v0, v1};
    drow += dstep;
  };
  // steps 1 .. Ti - 1 in chunks of K; the emission terms of a chunk are gathered one chunk ahead
  float pe0[K], pe1[K], nx0[K], nx1[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = 1 + k < Ti ? 1 + k : Ti - 1;
    const float* row = lp + (t0 + tstep * i) * CP;
    pe0[k] = row[blank];
    pe1[k] = row[lab];
  }
  for (int c0 = 1; c0 < Ti; c0 += K) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int i = c0 + K + k;
      if (i > Ti - 1) i = Ti - 1;
      const float* row = lp + (t0 + tstep * i) * CP;
      nx0[k] = row[blank];
      nx1[k] = row[lab];
    }
    if (c0 > 1 && W > 1) {                           // halo exchange: the top K pairs of wave w are the bottom K of wave w + 1
      const int par = ((c0 - 1) / K) & 1;
      if (lane >= 64 - K) {
        xch[par][w][lane - (64 - K)][0] = a0;
        xch[par][w][lane - (64 - K)][1] = a1;
      }
      lds_barrier();
      if (w > 0 && lane < K) {
        a0 = xch[par][w - 1][lane][0];
        a1 = xch[par][w - 1][lane][1];
      }
    }
    if (c0 + K <= Ti) {
#pragma unroll
      for (int k = 0; k < K; ++k) step(pe0[k], pe1[k]);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (c0 + k < Ti) step(pe0[k], pe1[k]);       // uniform
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pe0[k] = nx0[k];
      pe1[k] = nx1[k];
    }
  }
  if (!is_beta) {
    if (own && 2 * P == L - 1) fin[0] = a0;
    if (own && 2 * P + 1 == L - 2) fin[1] = a1;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float l1 = fin[0];
      const float l2 = L > 1 ? fin[1] : -INFINITY;
      const float ll = lse3(l1, l2, -INFINITY);
      const bool feasible = L <= 2 * Ti + 1 && ll != -INFINITY;      // zero_infinity: only an infinite loss is zeroed (see above)
      nll_out[n] = feasible ? -ll : 0.f;
      ws_ll[n] = feasible ? ll : INFINITY;
    }
  }
}
#ifndef DS2_PAIRS_K
#define DS2_PAIRS_K 16    // steps between halo exchanges = halo lanes (cfg3's call: 217 / 187 / 172 us with 4 / 8 / 16, r06s)
#endif
constexpr int PAIRS_K = DS2_PAIRS_K;
// blockIdx.x = sample, blockIdx.y = 0: alpha, 1: beta (the direction is a template parameter of the body: no per-step selects)
template <int K>
__global__ void __launch_bounds__(1024) k_ctc_recursion_pairs(const int* __restrict__ targets, const int* __restrict__ toff,
                                                              const int* __restrict__ in_len, const int* __restrict__ tg_len,
                                                              int Tp, int blank, int LS, int CP, float* __restrict__ nll_out,
                                                              const float* __restrict__ ws_lp, float* __restrict__ ws_alpha,
                                                              float* __restrict__ ws_beta, float* __restrict__ ws_ll) {
  __shared__ float xch[2][16][K][2];                 // [chunk parity][wave][halo lane]{blank, label}
  __shared__ float fin[2];
  if (blockIdx.y == 0)
    ctc_recursion_pairs<K, false>(targets, toff, in_len, tg_len, Tp, blank, LS, CP, nll_out, ws_lp, ws_alpha, ws_beta, ws_ll, xch, fin);
  else
    ctc_recursion_pairs<K, true>(targets, toff, in_len, tg_len, Tp, blank, LS, CP, nll_out, ws_lp, ws_alpha, ws_beta, ws_ll, xch, fin);
}
constexpr int PAIRS_MAX_WAVES = 16;
// waves a recursion over targets of up to `max_target_len` labels needs (max_target_len + 1 pairs)
static int pairs_waves(int max_target_len) {
  const int pp = max_target_len + 1;
  return pp <= 64 ? 1 : 1 + (pp - 64 + (64 - PAIRS_K) - 1) / (64 - PAIRS_K);
}

// grid (ceil(Tp / GRAD_FRAMES), N); one wave per frame.  Writes EVERY row of dlogits of its frames (zeros past the
// sample's length and for infeasible samples).
template <int CP>
__global__ void __launch_bounds__(GRAD_WAVES * 64) k_ctc_gradient(const int* __restrict__ targets, const int* __restrict__ toff,
                                                                  const int* __restrict__ in_len,
                                                                  const int* __restrict__ tg_len, int Tp, int N, int C,
                                                                  int blank, int Lmax, float grad_scale,
                                                                  float* __restrict__ dlogits, long ldg,
                                                                  const float* __restrict__ ws_lp,
                                                                  const float* __restrict__ ws_alpha,
                                                                  const float* __restrict__ ws_beta,
                                                                  const float* __restrict__ ws_ll) {
  __shared__ float acc[GRAD_WAVES][CP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y;
  int Ti = in_len[n];
  if (Ti > Tp) Ti = Tp;
  const int L = 2 * tg_len[n] + 1;
  const int* tg = targets + toff[n];
  const float ll = ws_ll[n];
  const bool live = ll != INFINITY;
  float* wacc = acc[wave];
  const int t_end = min((int)(blockIdx.x + 1) * GRAD_FRAMES, Tp);
  for (int t = blockIdx.x * GRAD_FRAMES + wave; t < t_end; t += GRAD_WAVES) {
    float* dl = dlogits + ((long)t * N + n) * ldg;
    if (!live || t >= Ti) {                          // wave-uniform
      for (int c = lane; c < ldg; c += 64) dl[c] = 0.f;
      continue;
    }
    const float* lpt = ws_lp + ((long)n * Tp + t) * CP;
    const float* al = ws_alpha + ((long)n * Tp + t) * Lmax;
    const float* be = ws_beta + ((long)n * Tp + t) * Lmax;
    for (int c = lane; c < CP; c += 64) wacc[c] = 0.f;
    __builtin_amdgcn_wave_barrier();
    float blank_part = 0.f;
    for (int s = lane; s < L; s += 64) {
      const float w = expf(al[s] + be[s] - ll);      // <= 1
      if (s & 1)
        atomicAdd(&wacc[tg[s >> 1]], w);             // within one wave: the order is fixed
      else
        blank_part += w;
    }
    blank_part = wave_sum(blank_part);
    __builtin_amdgcn_wave_barrier();
    constexpr int CL = (CP + 63) / 64;                 // classes per lane
    float g[CL], y[CL], gpart = 0.f;
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      const int c = lane + 64 * j;
      g[j] = y[j] = 0.f;
      if (c < C) {
        float a = wacc[c];
        if (c == blank) a += blank_part;
        y[j] = expf(lpt[c]);
        g[j] = a > 0.f ? -a / y[j] : 0.f;
      }
      gpart += g[j];
    }
    const float gs = wave_sum(gpart);
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      const int c = lane + 64 * j;
      if (c < C)
        dl[c] = (g[j] - y[j] * gs) * grad_scale;
      else if (c < ldg)
        dl[c] = 0.f;
    }
    for (int c = lane + 64 * CL; c < ldg; c += 64) dl[c] = 0.f;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- label sets of more than 256 classes (the reference takes any labels file, model.py:139,154-155,203) -----------------------
// Class rows no longer fit the LDS staging of the kernels above (a chunk of 8 rows of 1024 classes is 32 KB per buffer), and the
// recursion needs only the log-probabilities of the target's own extended labels: every state reads lp[t][ext[s]] straight from the
// scratch (L2-resident: written by the log-softmax pass just before), one step ahead of its use.  ldp = class stride (a multiple of
// 64).  Same formulas, same zero_infinity / NaN behaviour as k_ctc_recursion / k_ctc_gradient.
__global__ void __launch_bounds__(256) k_ctc_logsoftmax_big(const float* __restrict__ logits, long ldl, const int* __restrict__ in_len,
                                                            int Tp, int N, int C, int ldp, float* __restrict__ ws_lp) {
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);     // frame row t*N + n: one wave per frame
  const int lane = threadIdx.x & 63;
  if (i >= (long)Tp * N) return;
  const int t = (int)(i / N), n = (int)(i % N);
  if (t >= in_len[n]) return;
  const float* x = logits + i * ldl;
  float* lp = ws_lp + ((long)n * Tp + t) * ldp;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) sum += expf(x[c] - m);
  sum = wave_sum(sum);
  const float lz = m + logf(sum);
  for (int c = lane; c < C; c += 64) lp[c] = x[c] - lz;
}

__global__ void __launch_bounds__(REC_THREADS) k_ctc_recursion_big(const int* __restrict__ targets, const int* __restrict__ toff,
                                                                   const int* __restrict__ in_len, const int* __restrict__ tg_len,
                                                                   int Tp, int blank, int Lmax, int ldp, float* __restrict__ nll_out,
                                                                   const float* __restrict__ ws_lp, float* __restrict__ ws_alpha,
                                                                   float* __restrict__ ws_beta, float* __restrict__ ws_ll) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* prev = reinterpret_cast<float*>(smem);
  float* cur = prev + Lmax;
  int* ext = reinterpret_cast<int*>(cur + Lmax);
  const int tid = threadIdx.x, n = blockIdx.x;
  const bool is_beta = blockIdx.y != 0;
  int Ti = in_len[n];
  if (Ti > Tp) Ti = Tp;
  const int S = tg_len[n], L = 2 * S + 1;
  if (Ti <= 0) {
    if (!is_beta && tid == 0) {
      nll_out[n] = 0.f;
      ws_ll[n] = INFINITY;
    }
    return;
  }
  const int* tg = targets + toff[n];
  const float* lp = ws_lp + (long)n * Tp * ldp;
  float* dst = (is_beta ? ws_beta : ws_alpha) + (long)n * Tp * Lmax;
  for (int s = tid; s < L; s += REC_THREADS) ext[s] = (s & 1) ? tg[s >> 1] : blank;
  __syncthreads();
  for (int i = 0; i < Ti; ++i) {
    const int t = is_beta ? Ti - 1 - i : i;
    const float* lpt = lp + (long)t * ldp;
    for (int s = tid; s < L; s += REC_THREADS) {
      const int e = ext[s];
      float v;
      if (i == 0) {
        const bool entry = is_beta ? s >= L - 2 : s <= 1;
        v = entry ? lpt[e] : -INFINITY;
      } else if (!is_beta) {
        const float a0 = prev[s];
        const float a1 = s >= 1 ? prev[s - 1] : -INFINITY;
        const float a2 = (s >= 2 && e != blank && e != ext[s - 2]) ? prev[s - 2] : -INFINITY;
        v = lse3(a0, a1, a2) + lpt[e];
      } else {
        const float b0 = prev[s];
        const float b1 = s + 1 < L ? prev[s + 1] : -INFINITY;
        const float b2 = (s + 2 < L && ext[s + 2] != blank && ext[s + 2] != e) ? prev[s + 2] : -INFINITY;
        v = lse3(b0, b1, b2) + lpt[e];
      }
      cur[s] = v;
      dst[(long)t * Lmax + s] = v;
    }
    lds_barrier();
    float* tmp = prev;
    prev = cur;
    cur = tmp;
  }
  if (!is_beta && tid == 0) {
    const float l1 = prev[L - 1];
    const float l2 = L > 1 ? prev[L - 2] : -INFINITY;
    const float ll = lse3(l1, l2, -INFINITY);
    const bool feasible = L <= 2 * Ti + 1 && ll != -INFINITY;      // zero_infinity: see k_ctc_recursion
    nll_out[n] = feasible ? -ll : 0.f;
    ws_ll[n] = feasible ? ll : INFINITY;
  }
}

// grid (ceil(Tp / GRAD_FRAMES), N); one wave per frame; dynamic LDS: GRAD_WAVES x ldp floats
__global__ void __launch_bounds__(GRAD_WAVES * 64) k_ctc_gradient_big(const int* __restrict__ targets, const int* __restrict__ toff,
                                                                      const int* __restrict__ in_len, const int* __restrict__ tg_len,
                                                                      int Tp, int N, int C, int blank, int Lmax, int ldp, float grad_scale,
                                                                      float* __restrict__ dlogits, long ldg, const float* __restrict__ ws_lp,
                                                                      const float* __restrict__ ws_alpha, const float* __restrict__ ws_beta,
                                                                      const float* __restrict__ ws_ll) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wacc = reinterpret_cast<float*>(smem) + (long)wave * ldp;
  const int n = blockIdx.y;
  int Ti = in_len[n];
  if (Ti > Tp) Ti = Tp;
  const int L = 2 * tg_len[n] + 1;
  const int* tg = targets + toff[n];
  const float ll = ws_ll[n];
  const bool live = ll != INFINITY;
  const int t_end = min((int)(blockIdx.x + 1) * GRAD_FRAMES, Tp);
  for (int t = blockIdx.x * GRAD_FRAMES + wave; t < t_end; t += GRAD_WAVES) {
    float* dl = dlogits + ((long)t * N + n) * ldg;
    if (!live || t >= Ti) {                          // wave-uniform
      for (int c = lane; c < ldg; c += 64) dl[c] = 0.f;
      continue;
    }
    const float* lpt = ws_lp + ((long)n * Tp + t) * ldp;
    const float* al = ws_alpha + ((long)n * Tp + t) * Lmax;
    const float* be = ws_beta + ((long)n * Tp + t) * Lmax;
    for (int c = lane; c < ldp; c += 64) wacc[c] = 0.f;
    __builtin_amdgcn_wave_barrier();
    float blank_part = 0.f;
    for (int s = lane; s < L; s += 64) {
      const float w = expf(al[s] + be[s] - ll);      // <= 1
      if (s & 1)
        atomicAdd(&wacc[tg[s >> 1]], w);             // LDS atomics of one wave; a class that occurs in several states of one
      else                                           // 64-state pass is summed in the hardware's lane order (fixed)
        blank_part += w;
    }
    blank_part = wave_sum(blank_part);
    __builtin_amdgcn_wave_barrier();
    float gpart = 0.f;
    for (int c = lane; c < C; c += 64) {
      float a = wacc[c];
      if (c == blank) a += blank_part;
      const float y = expf(lpt[c]);
      const float g = a > 0.f ? -a / y : 0.f;
      wacc[c] = g;                                   // the lane's own class: reused as its g
      gpart += g;
    }
    const float gs = wave_sum(gpart);
    for (int c = lane; c < ldg; c += 64) dl[c] = c < C ? (wacc[c] - expf(lpt[c]) * gs) * grad_scale : 0.f;
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void k_sum_small(const float* __restrict__ v, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += v[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) *out = s;
}

}  // namespace

extern "C" {

// `recursion` argument of ds2_ctc_loss_grad (kernel selection for A/B runs and tests; identical results in every setting):
// 0 (default): the pair-tile kernel (k_ctc_recursion_pairs) for targets of up to 783 labels, the four-wave kernel beyond;
// 1: always the four-wave kernel (k_ctc_recursion / _big);  2: the one-wave kernel up to 255 labels (<= 32 classes), else four-wave;
// 3: the choice of rounds 3-5 -- one wave up to 63 labels (253 vs 330 us on 32 clips of 751 frames), four waves beyond (with 6 states
//    per lane -- cfg3's 180 labels -- the single SIMD's VALU throughput loses, 555 vs 414 us; profiles/r03g_ctc_ab.txt).
constexpr int CTC_MAX_CLASSES = 8192;     // the gradient pass keeps one row of per-class sums per wave in LDS (4 x 32 KB at 8192)
static int ctc_class_stride(int C) { return C <= 32 ? 32 : C <= 64 ? 64 : C <= 128 ? 128 : C <= 256 ? 256 : (C + 63) / 64 * 64; }

long ds2_ctc_ws_floats(int Tp, int N, int C, int max_target_len) {
  const long LS = 2L * max_target_len + 2;       // row stride of the alpha / beta rows: 2 S + 1 states + one pad (see ds2_ctc_loss_grad)
  return (long)N * Tp * ctc_class_stride(C) + 2L * N * Tp * LS + 2 + N;
}

int ds2_ctc_loss_grad(const float* logits, long ldl, const int* targets, const int* target_offsets, const int* input_lengths,
                      const int* target_lengths, int Tp, int N, int C, int blank, int max_target_len, float grad_scale,
                      float* nll, float* loss_sum, float* dlogits, long ldg, float* ws, int recursion, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  const int g_ctc_wave = recursion == 1 ? 0 : recursion == 2 ? 2 : recursion == 3 ? 1 : 0;
  const int pw = pairs_waves(max_target_len);
  const bool pairs = recursion == 0 && pw <= PAIRS_MAX_WAVES;
  DS2_REQUIRE(Tp > 0 && N > 0 && C > 0 && C <= CTC_MAX_CLASSES && blank >= 0 && blank < C && max_target_len >= 0, DS2_ERR_ARG);
  DS2_REQUIRE(ldg >= C && ldl >= C, DS2_ERR_ARG);
  const int CP = ctc_class_stride(C);
  // Row stride of the alpha / beta scratch rows: the 2 S_max + 1 states and one pad, i.e. EVEN, and the beta rows start one float
  // late -- the pair-tile recursion stores a (blank, label) pair with one aligned 8-byte store in both directions.  Every kernel below
  // takes the stride where it says Lmax.
  const int Lmax = 2 * max_target_len + 2;
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = dev >= 0 && dev < DS2_MAX_DEVICES ? dev : 0;
  float* ws_lp = ws;
  float* ws_alpha = ws + (long)N * Tp * CP;
  float* ws_beta = ws_alpha + (long)N * Tp * Lmax + 1;
  float* ws_ll = ws_beta + (long)N * Tp * Lmax + 1;
  if (C > 256) {      // large label sets: no class rows in LDS (see k_ctc_recursion_big)
    const size_t shm_r = (size_t)Lmax * 12, shm_g = (size_t)GRAD_WAVES * CP * 4;
    DS2_REQUIRE(shm_r <= 160 * 1024 && shm_g <= 160 * 1024, DS2_ERR_ARG);
    static size_t attr_r[DS2_MAX_DEVICES], attr_g[DS2_MAX_DEVICES];
    if (attr_r[dev] < shm_r) {
      (void)hipFuncSetAttribute((const void*)k_ctc_recursion_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_r);
      attr_r[dev] = shm_r;
    }
    if (attr_g[dev] < shm_g) {
      (void)hipFuncSetAttribute((const void*)k_ctc_gradient_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_g);
      attr_g[dev] = shm_g;
    }
    hipLaunchKernelGGL(k_ctc_logsoftmax_big, dim3(ds2_cdiv((long)Tp * N, 4)), dim3(256), 0, st, logits, ldl, input_lengths, Tp, N, C, CP, ws_lp);
    DS2_CHECK_LAUNCH();
    if (pairs)
      hipLaunchKernelGGL(k_ctc_recursion_pairs<PAIRS_K>, dim3(N, 2), dim3(64 * pw), 0, st, targets, target_offsets, input_lengths, target_lengths,
                         Tp, blank, Lmax, CP, nll, ws_lp, ws_alpha, ws_beta, ws_ll);
    else
      hipLaunchKernelGGL(k_ctc_recursion_big, dim3(N, 2), dim3(REC_THREADS), shm_r, st, targets, target_offsets, input_lengths, target_lengths,
                         Tp, blank, Lmax, CP, nll, ws_lp, ws_alpha, ws_beta, ws_ll);
    DS2_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ctc_gradient_big, dim3(ds2_cdiv(Tp, GRAD_FRAMES), N), dim3(GRAD_WAVES * 64), shm_g, st, targets, target_offsets,
                       input_lengths, target_lengths, Tp, N, C, blank, Lmax, CP, grad_scale, dlogits, ldg, ws_lp, ws_alpha, ws_beta, ws_ll);
    DS2_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(64), 0, st, nll, N, loss_sum);
    DS2_CHECK_LAUNCH();
    return 0;
  }
  const size_t shm = (size_t)Lmax * 12 + 2 * LP_CHUNK * CP * 4;
  DS2_REQUIRE(shm <= 160 * 1024, DS2_ERR_ARG);
#define DS2_CTC_LAUNCH(CPT, SLOT)                                                                                                   \
  {                                                                                                                                 \
    static size_t attr[DS2_MAX_DEVICES]; /* per device: the attribute belongs to the device's copy of the kernel */                 \
    if (attr[dev] < shm) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)k_ctc_recursion<CPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);           \
      attr[dev] = shm;                                                                                                              \
    }                                                                                                                               \
    hipLaunchKernelGGL(k_ctc_logsoftmax<CPT>, dim3(ds2_cdiv((long)Tp * N, 256)), dim3(256), 0, st, logits, ldl, input_lengths, Tp, \
                       N, C, ws_lp);                                                                                                \
    DS2_CHECK_LAUNCH();                                                                                                             \
    if (pairs) {                                                                                                                    \
      hipLaunchKernelGGL(k_ctc_recursion_pairs<PAIRS_K>, dim3(N, 2), dim3(64 * pw), 0, st, targets, target_offsets, input_lengths,  \
                         target_lengths, Tp, blank, Lmax, CPT, nll, ws_lp, ws_alpha, ws_beta, ws_ll);                               \
    } else if (CPT == 32 && ((g_ctc_wave == 1 && Lmax <= 128) || (g_ctc_wave == 2 && Lmax <= 512))) {                               \
      /* one wave per (sample, direction), 2 / 4 / 6 / 8 states per lane */                                                        \
      if (Lmax <= 128)                                                                                                              \
        hipLaunchKernelGGL(k_ctc_recursion_wave<2>, dim3(N, 2), dim3(64), 0, st, targets, target_offsets, input_lengths,            \
                           target_lengths, Tp, blank, Lmax, nll, ws_lp, ws_alpha, ws_beta, ws_ll);                                  \
      else if (Lmax <= 256)                                                                                                         \
        hipLaunchKernelGGL(k_ctc_recursion_wave<4>, dim3(N, 2), dim3(64), 0, st, targets, target_offsets, input_lengths,            \
                           target_lengths, Tp, blank, Lmax, nll, ws_lp, ws_alpha, ws_beta, ws_ll);                                  \
      else if (Lmax <= 384)                                                                                                         \
        hipLaunchKernelGGL(k_ctc_recursion_wave<6>, dim3(N, 2), dim3(64), 0, st, targets, target_offsets, input_lengths,            \
                           target_lengths, Tp, blank, Lmax, nll, ws_lp, ws_alpha, ws_beta, ws_ll);                                  \
      else                                                                                                                          \
        hipLaunchKernelGGL(k_ctc_recursion_wave<8>, dim3(N, 2), dim3(64), 0, st, targets, target_offsets, input_lengths,            \
                           target_lengths, Tp, blank, Lmax, nll, ws_lp, ws_alpha, ws_beta, ws_ll);                                  \
    } else {                                                                                                                        \
      hipLaunchKernelGGL(k_ctc_recursion<CPT>, dim3(N, 2), dim3(REC_THREADS), shm, st, targets, target_offsets, input_lengths,      \
                         target_lengths, Tp, blank, Lmax, nll, ws_lp, ws_alpha, ws_beta, ws_ll);                                    \
    }                                                                                                                               \
    DS2_CHECK_LAUNCH();                                                                                                             \
    hipLaunchKernelGGL(k_ctc_gradient<CPT>, dim3(ds2_cdiv(Tp, GRAD_FRAMES), N), dim3(GRAD_WAVES * 64), 0, st, targets,              \
                       target_offsets, input_lengths, target_lengths, Tp, N, C, blank, Lmax, grad_scale, dlogits, ldg, ws_lp,       \
                       ws_alpha, ws_beta, ws_ll);                                                                                   \
    DS2_CHECK_LAUNCH();                                                                                                             \
  }
  if (CP == 32) DS2_CTC_LAUNCH(32, 0)
  else if (CP == 64) DS2_CTC_LAUNCH(64, 1)
  else if (CP == 128) DS2_CTC_LAUNCH(128, 2)
  else DS2_CTC_LAUNCH(256, 3)
#undef DS2_CTC_LAUNCH
  hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(64), 0, st, nll, N, loss_sum);
  DS2_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
