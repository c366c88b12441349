// log_softmax + CTC loss + gradient in one launch (reference model.py:246 log_softmax(-1); model.py:203,248
// CTCLoss(blank, reduction='sum', zero_infinity=True) -> torch native _ctc_loss).
//
// One workgroup per sample (samples are independent).  Log-space alpha/beta recursion over the extended label sequence
// l' (2S+1 states, blank-interleaved); alpha rows are kept in a global scratch, the current/previous rows in LDS, one
// thread per state (strided for long targets).  Gradient: with both alpha_t(s) and beta_t(s) including y_t(l'_s),
//   d nll / d lp[t][c] = - sum_{s: l'_s = c} exp(alpha_t(s) + beta_t(s) - ll) / y_t(c)
// (every term exp(alpha+beta-ll) <= 1, so the total log-likelihood ll is a safe common shift), followed by the
// log_softmax backward  dlogit = g - softmax * sum_c g.  Infeasible samples (ll = -inf): loss 0, gradient 0.
// Latency-bound: alpha and beta run concurrently on the two halves of the workgroup (T' dependent steps, ONE barrier per
// step), the gradient pass is frame-parallel (one wave per frame); bytes: logits once, alpha/beta scratch write+read.
#include <math.h>

#include "ds2_common.h"

namespace {

constexpr int CTC_THREADS = 512;   // two halves of 256: alpha and beta, <= 2 states per thread for targets up to 255 labels
constexpr int CP = 32;  // padded class stride of the log-prob scratch

// log(exp a + exp b + exp c).  v_exp_f32 / v_log_f32 (1 ulp in the base-2 domain) instead of the full-precision libm
// sequences: the argument of the log is in [1, 3], where the absolute error of the fast form is < 2e-7 -- far below the
// fp32 resolution of the alpha/beta values themselves (magnitude ~3 T').
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == -INFINITY) return -INFINITY;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

__global__ void __launch_bounds__(CTC_THREADS) k_ctc(const float* __restrict__ logits, long ldl, const int* __restrict__ targets,
                                                      const int* __restrict__ toff, const int* __restrict__ in_len,
                                                      const int* __restrict__ tg_len, int Tp, int N, int C, int blank, int Lmax,
                                                      float grad_scale, float* __restrict__ nll_out, float* __restrict__ dlogits,
                                                      long ldg, float* __restrict__ ws_lp, float* __restrict__ ws_alpha,
                                                      float* __restrict__ ws_beta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rowA0 = reinterpret_cast<float*>(smem);     // alpha rows (double buffer)   [Lmax] each
  float* rowA1 = rowA0 + Lmax;
  float* rowB0 = rowA1 + Lmax;                       // beta rows
  float* rowB1 = rowB0 + Lmax;
  int* ext = reinterpret_cast<int*>(rowB1 + Lmax);   // [Lmax]
  float* acc = reinterpret_cast<float*>(ext + Lmax); // [CTC_THREADS / 64 waves][CP]
  float* lpbuf = acc + (CTC_THREADS / 64) * CP;      // [2 recursions][2 parities][CP]: log-prob rows of the current step
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x;
  int Ti = in_len[n];
  if (Ti > Tp) Ti = Tp;
  const int S = tg_len[n];
  const int L = 2 * S + 1;
  const int* tg = targets + toff[n];
  float* lp = ws_lp + (long)n * Tp * CP;                 // [Tp][CP] of this sample
  float* alpha = ws_alpha + (long)n * Tp * Lmax;         // [Tp][Lmax]
  float* beta = ws_beta + (long)n * Tp * Lmax;

  for (int s = tid; s < L; s += CTC_THREADS) ext[s] = (s & 1) ? tg[s >> 1] : blank;
  // ---- log_softmax of every valid frame
  for (int t = tid; t < Ti; t += CTC_THREADS) {
    const float* x = logits + ((long)t * N + n) * ldl;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, x[c]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(x[c] - m);
    const float lz = m + logf(sum);
    for (int c = 0; c < C; ++c) lp[(long)t * CP + c] = x[c] - lz;
  }
  // rows past the sample's length get a zero gradient
  for (long i = tid; i < (long)(Tp - Ti) * ldg; i += CTC_THREADS) {
    const long t = Ti + i / ldg, c = i % ldg;
    dlogits[(t * N + n) * ldg + c] = 0.f;
  }
  __syncthreads();

  bool feasible = Ti > 0 && L <= 2 * Ti + 1;   // necessary; the recursion decides exactly
  if (Ti <= 0) {
    if (tid == 0) nll_out[n] = 0.f;
    return;
  }
  // ---- alpha (threads 0..127, t = 0 .. Ti-1) and beta (threads 128..255, t = Ti-1 .. 0) run CONCURRENTLY: the two
  //      recursions are independent, so the serial depth is Ti instead of 2 Ti; one barrier per step serves both.
  {
    constexpr int HALF = CTC_THREADS / 2;
    const bool is_beta = tid >= HALF;
    const int ht = is_beta ? tid - HALF : tid;
    float* prev = is_beta ? rowB0 : rowA0;
    float* cur = is_beta ? rowB1 : rowA1;
    float* dst = is_beta ? beta : alpha;
    float* lpl = lpbuf + (is_beta ? 2 * CP : 0);       // this recursion's two row buffers
    // the frame's 29 log-probs are staged in LDS one step AHEAD (a dependent L2 round trip per step otherwise)
    if (ht < CP) lpl[ht] = lp[(long)(is_beta ? Ti - 1 : 0) * CP + ht];
    __syncthreads();
    for (int i = 0; i < Ti; ++i) {
      const int t = is_beta ? Ti - 1 - i : i;
      const float* lpt = lpl + (i & 1) * CP;
      float nxt = 0.f;
      const bool pre = ht < CP && i + 1 < Ti;
      if (pre) nxt = lp[(long)(is_beta ? t - 1 : t + 1) * CP + ht];
      for (int s = ht; s < L; s += HALF) {
        const int e = ext[s];
        float v;
        if (i == 0) {
          if (!is_beta)
            v = s == 0 ? lpt[blank] : (s == 1 ? lpt[e] : -INFINITY);
          else
            v = (s == L - 1 || s == L - 2) ? lpt[e] : -INFINITY;
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
      if (pre) lpl[((i + 1) & 1) * CP + ht] = nxt;
      __syncthreads();
      float* tmp = prev;
      prev = cur;
      cur = tmp;
    }
  }
  // after the loop every thread's `prev` of the alpha half holds alpha_{Ti-1}: it is rowA0 or rowA1 by parity
  const float* alast = (Ti & 1) ? rowA1 : rowA0;
  const float l1 = alast[L - 1];
  const float l2 = L > 1 ? alast[L - 2] : -INFINITY;
  const float ll = lse3(l1, l2, -INFINITY);
  feasible = feasible && (ll != -INFINITY) && !isnan(ll);
  if (tid == 0) nll_out[n] = feasible ? -ll : 0.f;
  if (!feasible) {
    for (long i = tid; i < (long)Ti * ldg; i += CTC_THREADS) {
      const long t = i / ldg, c = i % ldg;
      dlogits[(t * N + n) * ldg + c] = 0.f;
    }
    return;
  }
  // ---- gradient: frames are independent now -> one wave per frame (alpha/beta rows come back from L2)
  float* wacc = acc + wave * CP;
  for (int t = wave; t < Ti; t += CTC_THREADS / 64) {
    const float* lpt = lp + (long)t * CP;
    const float* al = alpha + (long)t * Lmax;
    const float* be = beta + (long)t * Lmax;
    if (lane < CP) wacc[lane] = 0.f;
    __builtin_amdgcn_wave_barrier();
    float blank_part = 0.f;
    for (int s = lane; s < L; s += 64) {
      const float w = expf(al[s] + be[s] - ll);   // <= 1
      if (s & 1)
        atomicAdd(&wacc[ext[s]], w);
      else
        blank_part += w;
    }
    blank_part = wave_sum(blank_part);
    __builtin_amdgcn_wave_barrier();
    float g = 0.f, y = 0.f;
    if (lane < C) {
      float a = wacc[lane];
      if (lane == blank) a += blank_part;
      y = expf(lpt[lane]);
      g = a > 0.f ? -a / y : 0.f;
    }
    const float gs = wave_sum(g);
    float* dl = dlogits + ((long)t * N + n) * ldg;
    if (lane < C)
      dl[lane] = (g - y * gs) * grad_scale;
    else if (lane < ldg)
      dl[lane] = 0.f;
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

long ds2_ctc_ws_floats(int Tp, int N, int C, int max_target_len) {
  (void)C;
  const long Lmax = 2L * max_target_len + 1;
  return (long)N * Tp * CP + 2L * N * Tp * Lmax;
}

int ds2_ctc_loss_grad(const float* logits, long ldl, const int* targets, const int* target_offsets, const int* input_lengths,
                      const int* target_lengths, int Tp, int N, int C, int blank, int max_target_len, float grad_scale,
                      float* nll, float* loss_sum, float* dlogits, long ldg, float* ws, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(Tp > 0 && N > 0 && C > 0 && C <= CP && blank >= 0 && blank < C && max_target_len >= 0, DS2_ERR_ARG);
  DS2_REQUIRE(ldg >= C && ldg <= 64 && ldl >= C, DS2_ERR_ARG);
  const int Lmax = 2 * max_target_len + 1;
  const size_t shm = (size_t)Lmax * 20 + (CTC_THREADS / 64) * CP * 4 + 4 * CP * 4;
  DS2_REQUIRE(shm <= 60 * 1024, DS2_ERR_ARG);
  float* ws_lp = ws;
  float* ws_alpha = ws + (long)N * Tp * CP;
  float* ws_beta = ws_alpha + (long)N * Tp * Lmax;
  hipLaunchKernelGGL(k_ctc, dim3(N), dim3(CTC_THREADS), shm, st, logits, ldl, targets, target_offsets, input_lengths,
                     target_lengths, Tp, N, C, blank, Lmax, grad_scale, nll, dlogits, ldg, ws_lp, ws_alpha, ws_beta);
  DS2_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(64), 0, st, nll, N, loss_sum);
  DS2_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
