// BatchNorm family (reference model.py:159,162 BatchNorm2d; model.py:28-33,86,196 SequenceWise BatchNorm1d).
//
// Everything is a "column norm" over a row-major matrix X[R][ld] with C channels in the fastest dimension:
//   * conv activations live in NFTC layout  -> R = N*F*T', C = 32, row = (n*F + f)*T' + t
//   * sequence activations are (T'*N) x H   -> R = T'*N,   C = H
// Statistics include every row (also the zero rows the time mask / padding produced), exactly like the reference.
// HBM-bound streaming kernels: 16-byte loads per lane, fp32 accumulation, deterministic two-stage reduction
// (per-block partials -> fp64 finalize).  Algorithmic bytes: read X once for stats, read X + write Y for apply.
#include "ds2_common.h"

namespace {

constexpr int NORM_RY = 8;     // rows per block iteration (threadIdx.y)
constexpr int NORM_CX = 32;    // 16-byte column chunks per block (threadIdx.x)

// maps a row of a conv activation (NFTC) to (n, t) for the time mask
struct RowMap {
  int F, Tp;  // F == 0 -> no mapping (sequence matrix)
  __device__ __forceinline__ void nt(long row, int& n, int& t) const {
    t = (int)(row % Tp);
    n = (int)(row / ((long)F * Tp));
  }
};

// The element arithmetic of the forward and backward passes, spelled with explicit fused operations: the forward apply, the backward
// reduce and the backward apply must take the SAME Hardtanh decision for an element on a clamp boundary, whatever the compiler would
// contract in each context.
__device__ __forceinline__ float bn_affine(float x, float sc, float sh) { return __builtin_fmaf(x, sc, sh); }
__device__ __forceinline__ bool hardtanh_open(float y) { return y > 0.f && y < 20.f; }
__device__ __forceinline__ float bn_dx(float g, float xh, float sc, float k1, float k2) { return sc * __builtin_fmaf(-xh, k2, g - k1); }

// ---------------------------------------------------------------------------------------------------------
// column sums / sums of squares -> partial[blockIdx.y][C]
// ---------------------------------------------------------------------------------------------------------
// Thread map shared by the two reduction kernels.  A block is NORM_CX x NORM_RY threads; with fewer than NORM_CX 16-byte
// column chunks (the conv activations have C = 32 channels = 4 bf16 chunks) the spare x-lanes take extra ROWS instead
// of idling: x = rsub * cpb + chunk, and one block iteration covers NORM_RY * rpx rows.
struct RedMap {
  int cpb, rpx, chunk, rsub;   // chunks per block in x, row sub-groups in x, this thread's chunk / row sub-group
  bool ok;
  __device__ __forceinline__ RedMap(int chunks) {
    cpb = chunks < NORM_CX ? chunks : NORM_CX;
    rpx = NORM_CX / cpb;
    const int tx = threadIdx.x;
    rsub = tx / cpb;
    chunk = blockIdx.x * cpb + tx % cpb;
    ok = rsub < rpx && chunk < chunks;
  }
  __device__ __forceinline__ long first_row() const { return ((long)blockIdx.y * NORM_RY + threadIdx.y) * rpx + rsub; }
  __device__ __forceinline__ long row_step() const { return (long)gridDim.y * NORM_RY * rpx; }
};
// block reduction of V values per thread over threadIdx.y and the row sub-groups; result in the threads (y = 0, rsub = 0)
template <int V>
__device__ __forceinline__ void red_block(float (*red)[NORM_CX * V + 1], const RedMap& m, const float (&v)[V], float (&out)[V]) {
  __syncthreads();
#pragma unroll
  for (int i = 0; i < V; ++i) red[threadIdx.y][threadIdx.x * V + i] = v[i];
  __syncthreads();
  if (threadIdx.y == 0 && m.rsub == 0) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float a = 0.f;
      for (int k = 0; k < m.rpx; ++k)
#pragma unroll
        for (int y = 0; y < NORM_RY; ++y) a += red[y][(threadIdx.x + k * m.cpb) * V + i];
      out[i] = a;
    }
  }
}

template <typename T, bool SQ>
__global__ void __launch_bounds__(NORM_RY* NORM_CX) k_colstats(const T* __restrict__ X, long R, int C, long ld,
                                                                float* __restrict__ psum, float* __restrict__ psq) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[NORM_RY][NORM_CX * V + 1];
  const RedMap m(C / V);
  float s[V], q[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s[i] = q[i] = 0.f;
  if (m.ok) {
    for (long r = m.first_row(); r < R; r += m.row_step()) {
      float v[V];
      Vec16<T>::load(X + r * ld + (long)m.chunk * V, v);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        s[i] += v[i];
        if (SQ) q[i] += v[i] * v[i];
      }
    }
  }
  float o[V];
  red_block<V>(red, m, s, o);
  if (threadIdx.y == 0 && m.rsub == 0 && m.ok) {
#pragma unroll
    for (int i = 0; i < V; ++i) psum[(long)blockIdx.y * C + (long)m.chunk * V + i] = o[i];
  }
  if (SQ) {
    red_block<V>(red, m, q, o);
    if (threadIdx.y == 0 && m.rsub == 0 && m.ok) {
#pragma unroll
      for (int i = 0; i < V; ++i) psq[(long)blockIdx.y * C + (long)m.chunk * V + i] = o[i];
    }
  }
}

// Column sums of the per-block partials, partial[P][C], accumulated in fp64.  A finalize block is 32 columns x 8 slices of
// P: every thread keeps its loads independent (a single thread walking P = 256 partials is 256 dependent L2 round trips,
// ~60 us for a kernel that moves a few KB), the 8 slices meet in LDS.  Returns the sums to the threads with py == 0.
constexpr int FIN_COLS = 32, FIN_SLICES = 8;
template <int NA>
__device__ __forceinline__ void finalize_sums(const float* const (&arr)[NA], int P, int C, int c, int py, double (&tot)[NA]) {
  __shared__ double sh[NA][FIN_SLICES][FIN_COLS];
  double acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acc[a] = 0.0;
  if (c < C) {
#pragma unroll 8
    for (int p = py; p < P; p += FIN_SLICES) {
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[a] += (double)arr[a][(long)p * C + c];
    }
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) sh[a][py][threadIdx.x & (FIN_COLS - 1)] = acc[a];
  __syncthreads();
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    double t = 0.0;
    if (py == 0) {
#pragma unroll
      for (int q = 0; q < FIN_SLICES; ++q) t += sh[a][q][threadIdx.x & (FIN_COLS - 1)];
    }
    tot[a] = t;
  }
}

__global__ void __launch_bounds__(256) k_colsum_finalize(const float* __restrict__ psum, int P, int C, float* __restrict__ out,
                                                          float scale) {
  const int c = blockIdx.x * FIN_COLS + (threadIdx.x & (FIN_COLS - 1)), py = threadIdx.x / FIN_COLS;
  const float* const arr[1] = {psum};
  double tot[1];
  finalize_sums<1>(arr, P, C, c, py, tot);
  if (py == 0 && c < C) out[c] = (float)(tot[0] * (double)scale);
}

// training-mode statistics -> mean, rstd, scale = gamma*rstd, shift = beta - mean*scale; running stats update
// (momentum, UNBIASED variance for running_var -- torch semantics), num_batches_tracked += 1.
__global__ void k_bn_finalize(const float* __restrict__ psum, const float* __restrict__ psq, int P, int C, double count,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                              float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                              long long* __restrict__ num_batches_tracked, float* __restrict__ mean_out,
                              float* __restrict__ rstd_out, float* __restrict__ scale_out, float* __restrict__ shift_out) {
  const int c = blockIdx.x * FIN_COLS + (threadIdx.x & (FIN_COLS - 1)), py = threadIdx.x / FIN_COLS;
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
  const float* const arr[2] = {psum, psq};
  double tot[2];
  finalize_sums<2>(arr, P, C, c, py, tot);
  if (py != 0 || c >= C) return;
  const double s = tot[0], q = tot[1];
  double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  double rstd = 1.0 / sqrt(var + (double)eps);
  mean_out[c] = (float)mean;
  rstd_out[c] = (float)rstd;
  float sc = (float)((double)gamma[c] * rstd);
  scale_out[c] = sc;
  shift_out[c] = (float)((double)beta[c] - mean * (double)gamma[c] * rstd);
  if (running_mean) {
    double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
  }
}

// eval-mode: scale/shift from the running statistics
__global__ void k_bn_eval_coeffs(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps,
                                 float* __restrict__ scale_out, float* __restrict__ shift_out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double rstd = 1.0 / sqrt((double)running_var[c] + (double)eps);
  scale_out[c] = (float)((double)gamma[c] * rstd);
  shift_out[c] = (float)((double)beta[c] - (double)running_mean[c] * (double)gamma[c] * rstd);
}

// ---------------------------------------------------------------------------------------------------------
// apply:  y = x*scale + shift   [CONV: then Hardtanh(0,20) and the time mask (model.py:61-68,160,163)]
// SEQ_OUT: additionally the output is written in sequence layout X0[(t*N + n)][f*32 + c] (the collapse+transpose of
// model.py:219-221 fused into the store; the internal feature order f*32+c is undone on the weight side).
// ---------------------------------------------------------------------------------------------------------
template <typename T, bool CONV, bool SEQ_OUT>
__global__ void __launch_bounds__(256) k_bn_apply(const T* __restrict__ X, T* __restrict__ Y, long R, int C, long ldx,
                                                   long ldy, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, RowMap rm,
                                                   const int* __restrict__ lens, int N) {
  constexpr int V = Vec16<T>::N;
  const int chunks = C / V;
  const long total = R * chunks;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long r = e / chunks;
    int c0 = (int)(e % chunks) * V;
    float v[V];
    bool live = true;
    int n = 0, t = 0;
    if (CONV) {
      rm.nt(r, n, t);
      live = t < lens[n];
    }
    if (live) {
      Vec16<T>::load(X + r * ldx + c0, v);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float y = bn_affine(v[i], scale[c0 + i], shift[c0 + i]);
        if (CONV) y = fminf(fmaxf(y, 0.f), 20.f);
        v[i] = y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) v[i] = 0.f;
    }
    if (SEQ_OUT) {
      int f = (int)((r / rm.Tp) % rm.F);
      Vec16<T>::store(Y + ((long)t * N + n) * ldy + (long)f * C + c0, v);
    } else {
      Vec16<T>::store(Y + r * ldy + c0, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward.  g = upstream gradient, gated for the conv case by Hardtanh' (0 < x*scale+shift < 20, strict, like
// torch hardtanh_backward) and by the time mask.  SEQ_IN: upstream gradient is read from the sequence layout.
//   reduce : sum_r g, sum_r g*xhat            (xhat = (x-mean)*rstd)
//   apply  : dx = scale * (g - c1 - xhat*c2)  [* mask]   with c1 = sum g / M, c2 = sum g*xhat / M
// ---------------------------------------------------------------------------------------------------------
template <typename T, bool CONV, bool SEQ_IN>
__device__ __forceinline__ bool load_gated(const T* __restrict__ G, const T* __restrict__ X, long r, int c0, int C,
                                           long ldg, long ldx, const float* __restrict__ scale,
                                           const float* __restrict__ shift, const float* __restrict__ mean,
                                           const float* __restrict__ rstd, const RowMap& rm, const int* __restrict__ lens,
                                           int N, float (&g)[Vec16<T>::N], float (&xh)[Vec16<T>::N]) {
  constexpr int V = Vec16<T>::N;
  int n = 0, t = 0, f = 0;
  if (CONV) {   // row -> (n, f, t) in 32-bit arithmetic (N*F*T' < 2^31: checked by the entry)
    const unsigned ur = (unsigned)r, q = ur / (unsigned)rm.Tp;
    t = (int)(ur - q * (unsigned)rm.Tp);
    n = (int)(q / (unsigned)rm.F);
    f = (int)(q - (unsigned)n * (unsigned)rm.F);
  }
  // all three loads are issued unconditionally and together (a masked row's values are discarded): a load of lens[n] that
  // gates the other two is two dependent memory round trips per row
  const int len = CONV ? lens[n] : 0;
  float x[V];
  Vec16<T>::load(X + r * ldx + c0, x);
  if (SEQ_IN) {
    Vec16<T>::load(G + ((long)t * N + n) * ldg + (long)f * C + c0, g);
  } else {
    Vec16<T>::load(G + r * ldg + c0, g);
  }
  if (CONV && t >= len) return false;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    if (CONV) {
      if (!hardtanh_open(bn_affine(x[i], scale[c0 + i], shift[c0 + i]))) g[i] = 0.f;
    }
    xh[i] = (x[i] - mean[c0 + i]) * rstd[c0 + i];
  }
  return true;
}

template <typename T, bool CONV, bool SEQ_IN>
__global__ void __launch_bounds__(NORM_RY* NORM_CX)
    k_bn_bwd_reduce(const T* __restrict__ G, const T* __restrict__ X, long R, int C, long ldg, long ldx,
                    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                    const float* __restrict__ rstd, RowMap rm, const int* __restrict__ lens, int N,
                    float* __restrict__ psum_g, float* __restrict__ psum_gx) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[NORM_RY][NORM_CX * V + 1];
  const RedMap m(C / V);
  float s[V], q[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s[i] = q[i] = 0.f;
  if (m.ok) {
    for (long r = m.first_row(); r < R; r += m.row_step()) {
      float g[V], xh[V];
      if (load_gated<T, CONV, SEQ_IN>(G, X, r, m.chunk * V, C, ldg, ldx, scale, shift, mean, rstd, rm, lens, N, g, xh)) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          s[i] += g[i];
          q[i] += g[i] * xh[i];
        }
      }
    }
  }
  float o[V];
  red_block<V>(red, m, s, o);
  if (threadIdx.y == 0 && m.rsub == 0 && m.ok) {
#pragma unroll
    for (int i = 0; i < V; ++i) psum_g[(long)blockIdx.y * C + (long)m.chunk * V + i] = o[i];
  }
  red_block<V>(red, m, q, o);
  if (threadIdx.y == 0 && m.rsub == 0 && m.ok) {
#pragma unroll
    for (int i = 0; i < V; ++i) psum_gx[(long)blockIdx.y * C + (long)m.chunk * V + i] = o[i];
  }
}

__global__ void k_bn_bwd_finalize(const float* __restrict__ psum_g, const float* __restrict__ psum_gx, int P, int C,
                                  double count, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                  float* __restrict__ c1, float* __restrict__ c2) {
  const int c = blockIdx.x * FIN_COLS + (threadIdx.x & (FIN_COLS - 1)), py = threadIdx.x / FIN_COLS;
  const float* const arr[2] = {psum_g, psum_gx};
  double tot[2];
  finalize_sums<2>(arr, P, C, c, py, tot);
  if (py != 0 || c >= C) return;
  const double s = tot[0], q = tot[1];
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
  c1[c] = (float)(s / count);
  c2[c] = (float)(q / count);
}

template <typename T, bool CONV, bool SEQ_IN>
__global__ void __launch_bounds__(256)
    k_bn_bwd_apply(const T* __restrict__ G, const T* __restrict__ X, T* __restrict__ DX, long R, int C, long ldg, long ldx,
                   long lddx, const float* __restrict__ scale, const float* __restrict__ shift,
                   const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ c1,
                   const float* __restrict__ c2, RowMap rm, const int* __restrict__ lens, int N) {
  constexpr int V = Vec16<T>::N;
  const int chunks = C / V;
  const long total = R * chunks;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long r = e / chunks;
    int c0 = (int)(e % chunks) * V;
    float g[V], xh[V], o[V];
    if (load_gated<T, CONV, SEQ_IN>(G, X, r, c0, C, ldg, ldx, scale, shift, mean, rstd, rm, lens, N, g, xh)) {
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = bn_dx(g[i], xh[i], scale[c0 + i], c1[c0 + i], c2[c0 + i]);
    } else {
      // masked position: the mask that follows the conv (model.py:61-68) zeroes this gradient
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = 0.f;
    }
    Vec16<T>::store(DX + r * lddx + c0, o);
  }
}

// Round 6: the same pass on the reductions' thread map -- a thread owns ONE 16-byte column chunk for the whole launch, so the six
// per-channel vectors of its chunk (scale, shift, mean, rstd, c1, c2: 48 values in bf16 storage) are loaded once instead of once per
// row (the grid-stride form above re-read them for every 16 bytes of payload: 3.1-3.9 TB/s on the conv tensors and at H = 1280), and
// two rows are in flight per thread.  Same arithmetic per element: bit-identical output.
template <typename T, bool CONV, bool SEQ_IN>
__global__ void __launch_bounds__(NORM_RY* NORM_CX)
    k_bn_bwd_apply_cols(const T* __restrict__ G, const T* __restrict__ X, T* __restrict__ DX, long R, int C, long ldg, long ldx,
                        long lddx, const float* __restrict__ scale, const float* __restrict__ shift,
                        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ c1,
                        const float* __restrict__ c2, RowMap rm, const int* __restrict__ lens, int N) {
  constexpr int V = Vec16<T>::N;
  const RedMap m(C / V);
  if (!m.ok) return;
  const int c0 = m.chunk * V;
  float sc[V], sh[V], mu[V], rs[V], k1[V], k2[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    sc[i] = scale[c0 + i];
    sh[i] = CONV ? shift[c0 + i] : 0.f;
    mu[i] = mean[c0 + i];
    rs[i] = rstd[c0 + i];
    k1[i] = c1[c0 + i];
    k2[i] = c2[c0 + i];
  }
  const long step = m.row_step();
  auto addr = [&](long r, int& t, int& len, const T*& gp) {
    int n = 0, f = 0;
    t = 0;
    if (CONV) {   // row -> (n, f, t) in 32-bit arithmetic (N*F*T' < 2^31: checked by the entry)
      const unsigned ur = (unsigned)r, q = ur / (unsigned)rm.Tp;
      t = (int)(ur - q * (unsigned)rm.Tp);
      n = (int)(q / (unsigned)rm.F);
      f = (int)(q - (unsigned)n * (unsigned)rm.F);
    }
    len = CONV ? lens[n] : 0;
    gp = SEQ_IN ? G + ((long)t * N + n) * ldg + (long)f * C + c0 : G + r * ldg + c0;
  };
  auto finish = [&](long r, int t, int len, float (&x)[V], float (&g)[V]) {
    float o[V];
    if (CONV && t >= len) {      // masked position: the mask that follows the conv (model.py:61-68) zeroes this gradient
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float gi = g[i];
        if (CONV) {
          if (!hardtanh_open(bn_affine(x[i], sc[i], sh[i]))) gi = 0.f;
        }
        const float xh = (x[i] - mu[i]) * rs[i];
        o[i] = bn_dx(gi, xh, sc[i], k1[i], k2[i]);
      }
    }
    Vec16<T>::store(DX + r * lddx + c0, o);
  };
  long r = m.first_row();
  for (; r + step < R; r += 2 * step) {
    int t0, t1, l0, l1;
    const T *g0p, *g1p;
    addr(r, t0, l0, g0p);
    addr(r + step, t1, l1, g1p);
    float x0[V], x1[V], g0[V], g1[V];
    Vec16<T>::load(X + r * ldx + c0, x0);
    Vec16<T>::load(X + (r + step) * ldx + c0, x1);
    Vec16<T>::load(g0p, g0);
    Vec16<T>::load(g1p, g1);
    finish(r, t0, l0, x0, g0);
    finish(r + step, t1, l1, x1, g1);
  }
  if (r < R) {
    int t0, l0;
    const T* g0p;
    addr(r, t0, l0, g0p);
    float x0[V], g0[V];
    Vec16<T>::load(X + r * ldx + c0, x0);
    Vec16<T>::load(g0p, g0);
    finish(r, t0, l0, x0, g0);
  }
}

// Row-blocks (= partial sums per column) of a column reduction.  The reductions are latency-bound on their loads (one 16-byte
// load per thread and iteration in flight): ~1024 workgroups in total -- 4 per CU -- instead of 256 took the 125-250 MB conv
// tensors (C = 32: a single column block) from 1-2 to 3 TB/s; wide matrices already get their workgroups from the column blocks.
inline int norm_grid_y(long R, int col_blocks = 1) {
  long gy = (R + NORM_RY - 1) / NORM_RY;
  long cap = 1024 / (col_blocks < 1 ? 1 : col_blocks);
  if (cap < 256) cap = 256;
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  return (int)gy;
}

template <typename T>
int colstats_impl(const void* X, long R, int C, long ld, float* psum, float* psq, int* P_out, hipStream_t st) {
  constexpr int V = Vec16<T>::N;
  DS2_REQUIRE(C % V == 0 && ld % V == 0, DS2_ERR_ALIGN);
  dim3 blk(NORM_CX, NORM_RY);
  dim3 grd(ds2_cdiv(C / V, NORM_CX), norm_grid_y(R, ds2_cdiv(C / V, NORM_CX)));
  if (psq)
    hipLaunchKernelGGL((k_colstats<T, true>), grd, blk, 0, st, (const T*)X, R, C, ld, psum, psq);
  else
    hipLaunchKernelGGL((k_colstats<T, false>), grd, blk, 0, st, (const T*)X, R, C, ld, psum, psq);
  *P_out = grd.y;
  DS2_CHECK_LAUNCH();
  return 0;
}

inline int apply_grid(long total) {
  long g = (total + 255) / 256;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" {

int ds2_norm_partials(long R) { return norm_grid_y(R); }

int ds2_colsum(int dtype, const void* X, long R, int C, long ld, float* out, float scale, float* ws, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  int P = 0, rc;
  if (dtype == DS2_F32)
    rc = colstats_impl<float>(X, R, C, ld, ws, nullptr, &P, st);
  else
    rc = colstats_impl<bf16_t>(X, R, C, ld, ws, nullptr, &P, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_colsum_finalize, dim3(ds2_cdiv(C, FIN_COLS)), dim3(256), 0, st, ws, P, C, out, scale);
  DS2_CHECK_LAUNCH();
  return 0;
}

// mode: 0 = sequence matrix (BatchNorm1d), 1 = conv activation in NFTC (hardtanh + time mask),
//       2 = conv activation, output/upstream-gradient in sequence layout (the last conv BN feeding the RNN stack)
int ds2_bn_fwd(int dtype, int mode, int training, const void* X, void* Y, long R, int C, long ldx, long ldy, int F, int Tp,
               int N, const int* lens, const float* gamma, const float* beta, float* running_mean, float* running_var,
               long long* num_batches_tracked, float eps, float momentum, float* save_mean, float* save_rstd,
               float* save_scale, float* save_shift, float* ws, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  DS2_REQUIRE(mode >= 0 && mode <= 2, DS2_ERR_ARG);
  DS2_REQUIRE(mode == 0 || R < (1L << 31), DS2_ERR_ARG);      // conv rows are decomposed in 32-bit arithmetic
  int rc, P = 0;
  if (training) {
    float* psum = ws;
    float* psq = ws + (long)norm_grid_y(R) * C;          // sized for the largest partial count (ds2_norm_partials)
    rc = dtype == DS2_F32 ? colstats_impl<float>(X, R, C, ldx, psum, psq, &P, st)
                          : colstats_impl<bf16_t>(X, R, C, ldx, psum, psq, &P, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bn_finalize, dim3(ds2_cdiv(C, FIN_COLS)), dim3(256), 0, st, psum, psq, P, C, (double)R, gamma, beta,
                       eps, momentum, running_mean, running_var, num_batches_tracked, save_mean, save_rstd, save_scale,
                       save_shift);
  } else {
    hipLaunchKernelGGL(k_bn_eval_coeffs, dim3(ds2_cdiv(C, 128)), dim3(128), 0, st, C, gamma, beta, running_mean,
                       running_var, eps, save_scale, save_shift);
  }
  DS2_CHECK_LAUNCH();
  RowMap rm{F, Tp};
  const int V = dtype == DS2_F32 ? 4 : 8;
  DS2_REQUIRE(C % V == 0 && ldx % V == 0 && ldy % V == 0, DS2_ERR_ALIGN);
  dim3 g(apply_grid(R * (C / V))), b(256);
#define LAUNCH_APPLY(TT)                                                                                              \
  if (mode == 0)                                                                                                      \
    hipLaunchKernelGGL((k_bn_apply<TT, false, false>), g, b, 0, st, (const TT*)X, (TT*)Y, R, C, ldx, ldy, save_scale,  \
                       save_shift, rm, lens, N);                                                                      \
  else if (mode == 1)                                                                                                 \
    hipLaunchKernelGGL((k_bn_apply<TT, true, false>), g, b, 0, st, (const TT*)X, (TT*)Y, R, C, ldx, ldy, save_scale,   \
                       save_shift, rm, lens, N);                                                                      \
  else                                                                                                                \
    hipLaunchKernelGGL((k_bn_apply<TT, true, true>), g, b, 0, st, (const TT*)X, (TT*)Y, R, C, ldx, ldy, save_scale,    \
                       save_shift, rm, lens, N);
  if (dtype == DS2_F32) {
    LAUNCH_APPLY(float)
  } else {
    LAUNCH_APPLY(bf16_t)
  }
#undef LAUNCH_APPLY
  DS2_CHECK_LAUNCH();
  return 0;
}

int ds2_bn_bwd(int dtype, int mode, const void* G, const void* X, void* DX, long R, int C, long ldg, long ldx, long lddx,
               int F, int Tp, int N, const int* lens, const float* save_mean, const float* save_rstd,
               const float* save_scale, const float* save_shift, float* dgamma, float* dbeta, float* ws, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  DS2_REQUIRE(mode >= 0 && mode <= 2, DS2_ERR_ARG);
  DS2_REQUIRE(mode == 0 || R < (1L << 31), DS2_ERR_ARG);      // conv rows are decomposed in 32-bit arithmetic
  const int V = dtype == DS2_F32 ? 4 : 8;
  DS2_REQUIRE(C % V == 0 && ldx % V == 0 && ldg % V == 0 && lddx % V == 0, DS2_ERR_ALIGN);
  RowMap rm{F, Tp};
  const int P = norm_grid_y(R, ds2_cdiv(C / V, NORM_CX));
  float* ps = ws;
  float* pq = ws + (long)P * C;
  float* c1 = pq + (long)P * C;
  float* c2 = c1 + C;
  dim3 blk(NORM_CX, NORM_RY), grd(ds2_cdiv(C / V, NORM_CX), P);
#define LAUNCH_RED(TT, CV, SI)                                                                                          \
  hipLaunchKernelGGL((k_bn_bwd_reduce<TT, CV, SI>), grd, blk, 0, st, (const TT*)G, (const TT*)X, R, C, ldg, ldx,          \
                     save_scale, save_shift, save_mean, save_rstd, rm, lens, N, ps, pq)
  // apply pass: column chunks fixed per thread (k_bn_bwd_apply_cols); ~2048 workgroups in all
  const int acb = ds2_cdiv(C / V, NORM_CX), arpx = NORM_CX / (C / V < NORM_CX ? C / V : NORM_CX);
  long agy = ds2_cdiv(R, (long)NORM_RY * arpx * 2);
  if (agy > 2048 / acb) agy = 2048 / acb;
  if (agy < 1) agy = 1;
  const dim3 agrd(acb, (unsigned)agy);
#ifdef DS2_BN_APPLY_GRIDSTRIDE   /* A/B (tools/ab_variants.py): the grid-stride form of rounds 1-5 */
#define LAUNCH_APP(TT, CV, SI)                                                                                          \
  hipLaunchKernelGGL((k_bn_bwd_apply<TT, CV, SI>), dim3(apply_grid(R*(C / V))), dim3(256), 0, st, (const TT*)G,           \
                     (const TT*)X, (TT*)DX, R, C, ldg, ldx, lddx, save_scale, save_shift, save_mean, save_rstd, c1, c2, \
                     rm, lens, N)
#else
#define LAUNCH_APP(TT, CV, SI)                                                                                          \
  hipLaunchKernelGGL((k_bn_bwd_apply_cols<TT, CV, SI>), agrd, blk, 0, st, (const TT*)G, (const TT*)X, (TT*)DX, R, C, ldg, \
                     ldx, lddx, save_scale, save_shift, save_mean, save_rstd, c1, c2, rm, lens, N)
#endif
#define BOTH(WHICH)                                        \
  if (dtype == DS2_F32) {                                  \
    if (mode == 0) WHICH(float, false, false);             \
    else if (mode == 1) WHICH(float, true, false);         \
    else WHICH(float, true, true);                         \
  } else {                                                 \
    if (mode == 0) WHICH(bf16_t, false, false);            \
    else if (mode == 1) WHICH(bf16_t, true, false);        \
    else WHICH(bf16_t, true, true);                        \
  }
  BOTH(LAUNCH_RED)
  DS2_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(ds2_cdiv(C, FIN_COLS)), dim3(256), 0, st, ps, pq, P, C, (double)R, dgamma, dbeta,
                     c1, c2);
  DS2_CHECK_LAUNCH();
  BOTH(LAUNCH_APP)
  DS2_CHECK_LAUNCH();
#undef BOTH
#undef LAUNCH_RED
#undef LAUNCH_APP
  return 0;
}

}  // extern "C"
