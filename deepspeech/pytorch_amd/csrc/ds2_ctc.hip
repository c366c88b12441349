// log_softmax + CTC loss + gradient in one launch (reference model.py:246 log_softmax(-1); model.py:203,248
// CTCLoss(blank, reduction='sum', zero_infinity=True) -> torch native _ctc_loss).
//
// One workgroup per sample (samples are independent).  Log-space alpha/beta recursion over the extended label sequence
// l' (2S+1 states, blank-interleaved); alpha rows are kept in a global scratch, the current/previous rows in LDS, one
// thread per state (strided for long targets).  Gradient: with both alpha_t(s) and beta_t(s) including y_t(l'_s),
//   d nll / d lp[t][c] = - sum_{s: l'_s = c} exp(alpha_t(s) + beta_t(s) - ll) / y_t(c)
// (every term exp(alpha+beta-ll) <= 1, so the total log-likelihood ll is a safe common shift), followed by the
// log_softmax backward  dlogit = g - softmax * sum_c g.  Infeasible samples (ll = -inf): loss 0, gradient 0.
// Latency-bound (T' dependent steps, 3 barriers per step); bytes: logits once, alpha scratch write+read.
#include <math.h>

#include "ds2_common.h"

namespace {

constexpr int CTC_THREADS = 256;
constexpr int CP = 32;  // padded class stride of the log-prob scratch

__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

__global__ void __launch_bounds__(CTC_THREADS) k_ctc(const float* __restrict__ logits, long ldl, const int* __restrict__ targets,
                                                      const int* __restrict__ toff, const int* __restrict__ in_len,
                                                      const int* __restrict__ tg_len, int Tp, int N, int C, int blank, int Lmax,
                                                      float grad_scale, float* __restrict__ nll_out, float* __restrict__ dlogits,
                                                      long ldg, float* __restrict__ ws_lp, float* __restrict__ ws_alpha) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rowA = reinterpret_cast<float*>(smem);      // [Lmax]
  float* rowB = rowA + Lmax;                         // [Lmax]
  int* ext = reinterpret_cast<int*>(rowB + Lmax);    // [Lmax]
  float* acc = reinterpret_cast<float*>(ext + Lmax); // [CP]
  float* wred = acc + CP;                            // [8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x;
  int Ti = in_len[n];
  if (Ti > Tp) Ti = Tp;
  const int S = tg_len[n];
  const int L = 2 * S + 1;
  const int* tg = targets + toff[n];
  float* lp = ws_lp + (long)n * Tp * CP;                 // [Tp][CP] of this sample
  float* alpha = ws_alpha + (long)n * Tp * Lmax;         // [Tp][Lmax]

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
  float ll = -INFINITY;
  if (Ti > 0) {
    // ---- alpha
    float* prev = rowA;
    float* cur = rowB;
    for (int s = tid; s < L; s += CTC_THREADS) {
      float v = -INFINITY;
      if (s == 0) v = lp[blank];
      if (s == 1) v = lp[ext[1]];
      prev[s] = v;
      alpha[s] = v;
    }
    __syncthreads();
    for (int t = 1; t < Ti; ++t) {
      const float* lpt = lp + (long)t * CP;
      for (int s = tid; s < L; s += CTC_THREADS) {
        const int e = ext[s];
        const float a = prev[s];
        const float b = s >= 1 ? prev[s - 1] : -INFINITY;
        const float c = (s >= 2 && e != blank && e != ext[s - 2]) ? prev[s - 2] : -INFINITY;
        const float v = lse3(a, b, c) + lpt[e];
        cur[s] = v;
        alpha[(long)t * Lmax + s] = v;
      }
      __syncthreads();
      float* tmp = prev;
      prev = cur;
      cur = tmp;
    }
    const float l1 = prev[L - 1];
    const float l2 = L > 1 ? prev[L - 2] : -INFINITY;
    ll = lse3(l1, l2, -INFINITY);
    __syncthreads();
    feasible = feasible && (ll != -INFINITY) && !isnan(ll);
    if (tid == 0) nll_out[n] = feasible ? -ll : 0.f;

    if (!feasible) {
      for (long i = tid; i < (long)Ti * ldg; i += CTC_THREADS) {
        const long t = i / ldg, c = i % ldg;
        dlogits[(t * N + n) * ldg + c] = 0.f;
      }
    } else {
      // ---- beta + gradient, t = Ti-1 .. 0 ; prev = beta_{t+1}, cur = beta_t
      float* bnext = rowA;
      float* bcur = rowB;
      for (int t = Ti - 1; t >= 0; --t) {
        const float* lpt = lp + (long)t * CP;
        const float* al = alpha + (long)t * Lmax;
        if (tid < CP) acc[tid] = 0.f;
        float blank_part = 0.f;
        // beta_t
        for (int s = tid; s < L; s += CTC_THREADS) {
          const int e = ext[s];
          float v;
          if (t == Ti - 1) {
            v = (s == L - 1 || s == L - 2) ? lpt[e] : -INFINITY;
          } else {
            const float a = bnext[s];
            const float b = s + 1 < L ? bnext[s + 1] : -INFINITY;
            const float c = (s + 2 < L && ext[s + 2] != blank && ext[s + 2] != e) ? bnext[s + 2] : -INFINITY;
            v = lse3(a, b, c) + lpt[e];
          }
          bcur[s] = v;
        }
        __syncthreads();   // acc zeroed, beta_t complete
        for (int s = tid; s < L; s += CTC_THREADS) {
          const float w = expf(al[s] + bcur[s] - ll);   // <= 1
          if (s & 1)
            atomicAdd(&acc[ext[s]], w);
          else
            blank_part += w;
        }
        blank_part = wave_sum(blank_part);
        if (lane == 0) wred[wave] = blank_part;
        __syncthreads();   // label atomics + per-wave blank sums visible
        if (tid < 64) {
          float g = 0.f, y = 0.f;
          if (tid < C) {
            float a = acc[tid];
            if (tid == blank) a += wred[0] + wred[1] + wred[2] + wred[3];
            y = expf(lpt[tid]);
            g = a > 0.f ? -a / y : 0.f;
          }
          const float gs = wave_sum(g);
          float* dl = dlogits + ((long)t * N + n) * ldg;
          if (tid < C)
            dl[tid] = (g - y * gs) * grad_scale;
          else if (tid < ldg)
            dl[tid] = 0.f;
        }
        __syncthreads();   // acc / wred consumed before the next step zeroes them
        float* tmp = bnext;
        bnext = bcur;
        bcur = tmp;
      }
    }
  } else {
    if (tid == 0) nll_out[n] = 0.f;
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
  return (long)N * Tp * CP + (long)N * Tp * Lmax;
}

int ds2_ctc_loss_grad(const float* logits, long ldl, const int* targets, const int* target_offsets, const int* input_lengths,
                      const int* target_lengths, int Tp, int N, int C, int blank, int max_target_len, float grad_scale,
                      float* nll, float* loss_sum, float* dlogits, long ldg, float* ws, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(Tp > 0 && N > 0 && C > 0 && C <= CP && blank >= 0 && blank < C && max_target_len >= 0, DS2_ERR_ARG);
  DS2_REQUIRE(ldg >= C && ldg <= 64 && ldl >= C, DS2_ERR_ARG);
  const int Lmax = 2 * max_target_len + 1;
  const size_t shm = (size_t)Lmax * 12 + CP * 4 + 8 * 4;
  DS2_REQUIRE(shm <= 60 * 1024, DS2_ERR_ARG);
  float* ws_lp = ws;
  float* ws_alpha = ws + (long)N * Tp * CP;
  hipLaunchKernelGGL(k_ctc, dim3(N), dim3(CTC_THREADS), shm, st, logits, ldl, targets, target_offsets, input_lengths,
                     target_lengths, Tp, N, C, blank, Lmax, grad_scale, nll, dlogits, ldg, ws_lp, ws_alpha);
  DS2_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(64), 0, st, nll, N, loss_sum);
  DS2_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
