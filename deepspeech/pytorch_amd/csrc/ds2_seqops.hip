// Small sequence ops: direction sum (model.py:101), 2-D transpose (operand re-layout for the wgrad GEMMs), Lookahead
// (model.py:105-135 + Hardtanh model.py:189-193), inference softmax (model.py:72-77).  All HBM-bound streaming kernels.
#include "ds2_common.h"

namespace {

template <typename T>
__global__ void __launch_bounds__(256) k_add2(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long nvec) {
  constexpr int V = Vec16<T>::N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    float x[V], y[V];
    Vec16<T>::load(a + i * V, x);
    Vec16<T>::load(b + i * V, y);
#pragma unroll
    for (int k = 0; k < V; ++k) x[k] += y[k];
    Vec16<T>::store(o + i * V, x);
  }
}

// out[i] = sum_{s < slices} src[s * n + i], slices added in index order (fixed summation order: the K-slices of a split product)
__global__ void __launch_bounds__(256) k_sum_slices(const float* __restrict__ src, float* __restrict__ o, long nvec, int slices) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    float4 acc = reinterpret_cast<const float4*>(src)[i];
    for (int s = 1; s < slices; ++s) {
      const float4 v = reinterpret_cast<const float4*>(src)[(long)s * nvec + i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(o)[i] = acc;
  }
}

// dst[c][r] = src[r][c]; tile 64x64 through LDS (row stride padded by one 4-byte word -> conflict-free column reads)
template <typename E>   // E = storage element (uint16_t or float)
__global__ void __launch_bounds__(256) k_transpose(const E* __restrict__ src, E* __restrict__ dst, long R, int C, long lds_,
                                                    long ldd) {
  constexpr int V = 16 / (int)sizeof(E);
  constexpr int PADE = 4 / (int)sizeof(E) > 0 ? 4 / (int)sizeof(E) : 1;
  __shared__ E tile[64][64 + PADE];
  const long r0 = (long)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * (64 / V); i += 256) {
    const int rr = i / (64 / V), cv = i % (64 / V);
    const long r = r0 + rr;
    const int c = c0 + cv * V;
    alignas(16) E tmp[V];
    if (r < R && c < C) {
      *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(src + r * lds_ + c);
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) tmp[k] = (E)0;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) tile[rr][cv * V + k] = tmp[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * (64 / V); i += 256) {
    const int cc = i / (64 / V), rv = i % (64 / V);
    const int c = c0 + cc;
    const long r = r0 + rv * V;
    if (c < C && r < ldd) {   // ldd % V == 0: rows r in [R, ldd) are written as zeros (tile is zero there)
      alignas(16) E tmp[V];
#pragma unroll
      for (int k = 0; k < V; ++k) tmp[k] = tile[rv * V + k][cc];
      *reinterpret_cast<uint4*>(dst + (long)c * ldd + r) = *reinterpret_cast<uint4*>(tmp);
    }
  }
}

// Lookahead (model.py:105-135): a depthwise convolution over time, y[t][n][h] = hardtanh( sum_k w[h][k] * x[t+k][n][h] ), x[t>=Tp] = 0.
// HBM-bound by nature (x is read once per output and tap only through the caches); the kernels are organised so that the LOAD
// INSTRUCTION count is small: a thread owns one 16-byte channel vector of one sample for a block of LA_TB consecutive frames and
// walks the taps outermost, so a tap's weights are fetched once per LA_TB outputs (the first version fetched them per output: 9
// load instructions per output-tap, 3.1 ms per forward on config 5b; now 1.5).  The summation order over the taps is unchanged.
constexpr int LA_TB = 16;

template <typename T>
__global__ void __launch_bounds__(256) k_lookahead_fwd(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                        T* __restrict__ pre, int Tp, int N, int H, int ctx) {
  constexpr int V = Vec16<T>::N;
  const int hv = H / V;
  const long total = (long)ds2_cdiv_dev(Tp, LA_TB) * N * hv;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int h0 = (int)(e % hv) * V;
    const long rest = e / hv;
    const int n = (int)(rest % N), t0 = (int)(rest / N) * LA_TB;
    float acc[LA_TB][V];
#pragma unroll
    for (int j = 0; j < LA_TB; ++j)
#pragma unroll
      for (int i = 0; i < V; ++i) acc[j][i] = 0.f;
    for (int k = 0; k < ctx; ++k) {
      float wk[V];
#pragma unroll
      for (int i = 0; i < V; ++i) wk[i] = w[(long)(h0 + i) * ctx + k];
#pragma unroll
      for (int j = 0; j < LA_TB; ++j) {
        const int t = t0 + j + k;
        if (t < Tp) {
          float xv[V];
          Vec16<T>::load(x + ((long)t * N + n) * H + h0, xv);
#pragma unroll
          for (int i = 0; i < V; ++i) acc[j][i] = fmaf(wk[i], xv[i], acc[j][i]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < LA_TB; ++j) {
      if (t0 + j >= Tp) break;
      const long o = ((long)(t0 + j) * N + n) * H + h0;
      if (pre) Vec16<T>::store(pre + o, acc[j]);   // pre-activation, kept for Hardtanh' in backward
#pragma unroll
      for (int i = 0; i < V; ++i) acc[j][i] = fminf(fmaxf(acc[j][i], 0.f), 20.f);
      Vec16<T>::store(y + o, acc[j]);
    }
  }
}

// backward wrt x: g = dy * [0 < pre < 20];  dx[t] = sum_k w[h][k] * g[t-k]
template <typename T>
__global__ void __launch_bounds__(256) k_lookahead_bwd_x(const T* __restrict__ dy, const T* __restrict__ pre,
                                                          const float* __restrict__ w, T* __restrict__ dx, int Tp, int N, int H,
                                                          int ctx) {
  constexpr int V = Vec16<T>::N;
  const int hv = H / V;
  const long total = (long)ds2_cdiv_dev(Tp, LA_TB) * N * hv;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int h0 = (int)(e % hv) * V;
    const long rest = e / hv;
    const int n = (int)(rest % N), t0 = (int)(rest / N) * LA_TB;
    float acc[LA_TB][V];
#pragma unroll
    for (int j = 0; j < LA_TB; ++j)
#pragma unroll
      for (int i = 0; i < V; ++i) acc[j][i] = 0.f;
    for (int k = 0; k < ctx; ++k) {
      float wk[V];
#pragma unroll
      for (int i = 0; i < V; ++i) wk[i] = w[(long)(h0 + i) * ctx + k];
#pragma unroll
      for (int j = 0; j < LA_TB; ++j) {
        const int t = t0 + j - k;
        if (t >= 0 && t0 + j < Tp) {
          float g[V], p[V];
          const long off = ((long)t * N + n) * H + h0;
          Vec16<T>::load(dy + off, g);
          Vec16<T>::load(pre + off, p);
#pragma unroll
          for (int i = 0; i < V; ++i)
            if (p[i] > 0.f && p[i] < 20.f) acc[j][i] = fmaf(wk[i], g[i], acc[j][i]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < LA_TB; ++j) {
      if (t0 + j >= Tp) break;
      Vec16<T>::store(dx + ((long)(t0 + j) * N + n) * H + h0, acc[j]);
    }
  }
}

// four consecutive channels as one 8-byte (bf16) / 16-byte (fp32) load
__device__ __forceinline__ void la_load4(const bf16_t* p, float (&o)[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ void la_load4(const float* p, float (&o)[4]) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}

// backward wrt w: dw[h][k] = sum_{t,n} g[t][n][h] * x[t+k][n][h]; partial over row blocks -> ws[P][H*ctx].  A thread owns FOUR
// channels (one vector load per operand and row) and 32 taps in registers per pass.
constexpr int LA_ROWS = 256;  // (t,n) rows per block
template <typename T>
__global__ void __launch_bounds__(256) k_lookahead_bwd_w(const T* __restrict__ dy, const T* __restrict__ pre,
                                                          const T* __restrict__ x, float* __restrict__ partial, int Tp, int N,
                                                          int H, int ctx) {
  const int h0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (h0 >= H) return;
  const long rows = (long)Tp * N;
  const long r0 = (long)blockIdx.y * LA_ROWS, r1 = min(rows, r0 + LA_ROWS);
  float* dst = partial + (long)blockIdx.y * H * ctx + (long)h0 * ctx;
  for (int k0 = 0; k0 < ctx; k0 += 32) {               // 32 taps in registers per pass (the default context is 20: one pass)
    float acc[32][4];
#pragma unroll
    for (int k = 0; k < 32; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;
    for (long r = r0; r < r1; ++r) {
      float p[4], g[4];
      la_load4(pre + r * H + h0, p);
      bool act[4], any = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        act[i] = p[i] > 0.f && p[i] < 20.f;
        any |= act[i];
      }
      if (!any) continue;
      la_load4(dy + r * H + h0, g);
      const int t = (int)(r / N);
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (k0 + k < ctx && t + k0 + k < Tp) {
          float xv[4];
          la_load4(x + (r + (long)(k0 + k) * N) * H + h0, xv);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (act[i]) acc[k][i] = fmaf(g[i], xv[i], acc[k][i]);
        }
    }
#pragma unroll
    for (int k = 0; k < 32; ++k)
      if (k0 + k < ctx) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[(long)i * ctx + k0 + k] = acc[k][i];
      }
  }
}

// ---- sliding-window versions for the reference's default context (CTX compiled in) ----------------------------------------
// The kernels above read every operand row once per TAP (20 x the tensor through L1/L2: 1.9 ms for the weight gradient, 0.9 for
// the data gradient, 0.5 forward at config 5b).  Here a thread walks a time segment of one (sample, four channels) column with the
// last CTX operand values in registers: one load per operand and frame.  The frame loop is unrolled CTX times so that every
// window slot is a compile-time register index; a segment re-reads CTX-1 halo frames (19 % at 100-frame segments).  The
// summation order over the taps is the one of the kernels above.
constexpr int LA_SEG_STEPS = 5;          // a segment = LA_SEG_STEPS * CTX frames
__device__ __forceinline__ void la_store4(bf16_t* p, const float (&v)[4]) {
  uint2 o;
  o.x = cvt_pk_bf16(v[0], v[1]);
  o.y = cvt_pk_bf16(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = o;
}
__device__ __forceinline__ void la_store4(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
struct LaCol { int h0, n, t0, t1; bool ok; };
template <int CTX>
__device__ __forceinline__ LaCol la_column(int Tp, int N, int H) {
  const int hv = H / 4;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  LaCol c;
  c.h0 = (int)(e % hv) * 4;
  const long rest = e / hv;
  c.n = (int)(rest % N);
  c.t0 = (int)(rest / N) * (LA_SEG_STEPS * CTX);
  c.t1 = min(Tp, c.t0 + LA_SEG_STEPS * CTX);
  c.ok = c.t0 < Tp;
  return c;
}

// Round 6: the walk was latency-bound -- a frame's operand was loaded, converted and multiplied in the same step, behind a branch on
// the segment end (one dependent L2 / HBM round trip per frame: 2.6 TB/s forward).  Now the RAW words of the operands of the next
// LA_Q frames sit in a small register queue (loaded LA_Q steps ahead from clamped addresses, no branch; converted, masked and gated
// when they are consumed), only the stores are guarded.  Same values, same summation order.
#ifndef DS2_LA_Q
#define DS2_LA_Q 4
#endif
constexpr int LA_Q = DS2_LA_Q;
template <typename T> struct LaRaw;
template <> struct LaRaw<bf16_t> {
  typedef uint2 type;
  static __device__ __forceinline__ uint2 load(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ void cvt(const uint2& v, float (&o)[4]) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
};
template <> struct LaRaw<float> {
  typedef float4 type;
  static __device__ __forceinline__ float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void cvt(const float4& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};
// row of frame t clamped into [0, Tp)
__device__ __forceinline__ long la_row(int t, int Tp, long ts, long col) { return (long)min(max(t, 0), Tp - 1) * ts + col; }

template <typename T, int CTX>
__global__ void __launch_bounds__(256) k_lookahead_fwd_slide(const T* __restrict__ x, const float* __restrict__ w,
                                                              T* __restrict__ y, T* __restrict__ pre, int Tp, int N, int H) {
  static_assert(CTX % LA_Q == 0, "queue slots are compile-time indices of the unrolled frame loop");
  typedef LaRaw<T> Raw;
  const LaCol c = la_column<CTX>(Tp, N, H);
  if (!c.ok) return;
  float wk[CTX][4], xw[CTX][4];
#pragma unroll
  for (int k = 0; k < CTX; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) wk[k][i] = w[(long)(c.h0 + i) * CTX + k];
  const long col = (long)c.n * H + c.h0, ts = (long)N * H;
#pragma unroll
  for (int s = 0; s < CTX - 1; ++s) {                       // x[t0 .. t0+CTX-2] -> slots 0 .. CTX-2
    la_load4(x + la_row(c.t0 + s, Tp, ts, col), xw[s]);
    if (c.t0 + s >= Tp)
#pragma unroll
      for (int i = 0; i < 4; ++i) xw[s][i] = 0.f;
  }
  typename Raw::type q[LA_Q];                               // x[t + CTX-1 + d], d < LA_Q, of the frame t about to be computed
#pragma unroll
  for (int d = 0; d < LA_Q; ++d) q[d] = Raw::load(x + la_row(c.t0 + CTX - 1 + d, Tp, ts, col));
  for (int tb = c.t0; tb < c.t1; tb += CTX) {
#pragma unroll
    for (int j = 0; j < CTX; ++j) {
      const int t = tb + j;
      const int tn = t + CTX - 1, slot = (j + CTX - 1) % CTX;
      Raw::cvt(q[j % LA_Q], xw[slot]);
      if (tn >= Tp)
#pragma unroll
        for (int i = 0; i < 4; ++i) xw[slot][i] = 0.f;
      q[j % LA_Q] = Raw::load(x + la_row(tn + LA_Q, Tp, ts, col));
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < CTX; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(wk[k][i], xw[(j + k) % CTX][i], acc[i]);
      if (t < c.t1) {
        const long o = (long)t * ts + col;
        if (pre) la_store4(pre + o, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fminf(fmaxf(acc[i], 0.f), 20.f);
        la_store4(y + o, acc);
      }
    }
  }
}

// g[t] = dy[t] * [0 < pre[t] < 20] of this thread's column; zero outside [0, Tp)
template <typename T>
__device__ __forceinline__ void la_load_g(const T* __restrict__ dy, const T* __restrict__ pre, int t, int Tp, long ts, long col,
                                          float (&g)[4]) {
  const long o = la_row(t, Tp, ts, col);
  float p[4];
  la_load4(dy + o, g);
  la_load4(pre + o, p);
  const bool in = t >= 0 && t < Tp;
#pragma unroll
  for (int i = 0; i < 4; ++i) g[i] = (in && p[i] > 0.f && p[i] < 20.f) ? g[i] : 0.f;
}
// the same from queued raw words
template <typename T>
__device__ __forceinline__ void la_gate(const typename LaRaw<T>::type& rg, const typename LaRaw<T>::type& rp, int t, int Tp,
                                        float (&g)[4]) {
  float p[4];
  LaRaw<T>::cvt(rg, g);
  LaRaw<T>::cvt(rp, p);
  const bool in = t >= 0 && t < Tp;
#pragma unroll
  for (int i = 0; i < 4; ++i) g[i] = (in && p[i] > 0.f && p[i] < 20.f) ? g[i] : 0.f;
}

template <typename T, int CTX>
__global__ void __launch_bounds__(256) k_lookahead_bwd_x_slide(const T* __restrict__ dy, const T* __restrict__ pre,
                                                                const float* __restrict__ w, T* __restrict__ dx, int Tp, int N,
                                                                int H) {
  typedef LaRaw<T> Raw;
  const LaCol c = la_column<CTX>(Tp, N, H);
  if (!c.ok) return;
  float wk[CTX][4], gw[CTX][4];
#pragma unroll
  for (int k = 0; k < CTX; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) wk[k][i] = w[(long)(c.h0 + i) * CTX + k];
  const long col = (long)c.n * H + c.h0, ts = (long)N * H;
#pragma unroll
  for (int s = 1; s < CTX; ++s) la_load_g(dy, pre, c.t0 - CTX + s, Tp, ts, col, gw[s]);   // g[t0-CTX+1 .. t0-1] -> slots 1 .. CTX-1
  typename Raw::type qg[LA_Q], qp[LA_Q];                    // dy / pre of frames t + d, d < LA_Q
#pragma unroll
  for (int d = 0; d < LA_Q; ++d) {
    qg[d] = Raw::load(dy + la_row(c.t0 + d, Tp, ts, col));
    qp[d] = Raw::load(pre + la_row(c.t0 + d, Tp, ts, col));
  }
  for (int tb = c.t0; tb < c.t1; tb += CTX) {
#pragma unroll
    for (int j = 0; j < CTX; ++j) {
      const int t = tb + j;
      la_gate<T>(qg[j % LA_Q], qp[j % LA_Q], t, Tp, gw[j]);
      qg[j % LA_Q] = Raw::load(dy + la_row(t + LA_Q, Tp, ts, col));
      qp[j % LA_Q] = Raw::load(pre + la_row(t + LA_Q, Tp, ts, col));
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < CTX; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(wk[k][i], gw[(j - k + CTX) % CTX][i], acc[i]);
      if (t < c.t1) la_store4(dx + (long)t * ts + col, acc);
    }
  }
}

// partial[(segment * N + n)][h * CTX + k] = sum over the segment's frames t' of g[t'-k] * x[t']
template <typename T, int CTX>
__global__ void __launch_bounds__(256) k_lookahead_bwd_w_slide(const T* __restrict__ dy, const T* __restrict__ pre,
                                                                const T* __restrict__ x, float* __restrict__ partial, int Tp,
                                                                int N, int H) {
  typedef LaRaw<T> Raw;
  const LaCol c = la_column<CTX>(Tp, N, H);
  if (!c.ok) return;
  float acc[CTX][4], gw[CTX][4];
#pragma unroll
  for (int k = 0; k < CTX; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;
  const long col = (long)c.n * H + c.h0, ts = (long)N * H;
#pragma unroll
  for (int s = 1; s < CTX; ++s) la_load_g(dy, pre, c.t0 - CTX + s, Tp, ts, col, gw[s]);
  typename Raw::type qg[LA_Q], qp[LA_Q], qx[LA_Q];
#pragma unroll
  for (int d = 0; d < LA_Q; ++d) {
    qg[d] = Raw::load(dy + la_row(c.t0 + d, Tp, ts, col));
    qp[d] = Raw::load(pre + la_row(c.t0 + d, Tp, ts, col));
    qx[d] = Raw::load(x + la_row(c.t0 + d, Tp, ts, col));
  }
  for (int tb = c.t0; tb < c.t1; tb += CTX) {
#pragma unroll
    for (int j = 0; j < CTX; ++j) {
      const int t = tb + j;
      float xv[4];
      la_gate<T>(qg[j % LA_Q], qp[j % LA_Q], t, Tp, gw[j]);
      Raw::cvt(qx[j % LA_Q], xv);
      qg[j % LA_Q] = Raw::load(dy + la_row(t + LA_Q, Tp, ts, col));
      qp[j % LA_Q] = Raw::load(pre + la_row(t + LA_Q, Tp, ts, col));
      qx[j % LA_Q] = Raw::load(x + la_row(t + LA_Q, Tp, ts, col));
      if (t >= c.t1)                                         // frames of the NEXT segment (or past the end) do not count here
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = 0.f;
#pragma unroll
      for (int k = 0; k < CTX; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[k][i] = fmaf(gw[(j - k + CTX) % CTX][i], xv[i], acc[k][i]);
    }
  }
  float* dst = partial + ((long)(c.t0 / (LA_SEG_STEPS * CTX)) * N + c.n) * H * CTX + (long)c.h0 * CTX;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < CTX; k += 4)
      *reinterpret_cast<float4*>(dst + (long)i * CTX + k) = make_float4(acc[k][i], acc[k + 1][i], acc[k + 2][i], acc[k + 3][i]);
}
constexpr int LA_CTX = 20;               // the compiled-in context (reference default, model.py:117); others take the kernels above

__global__ void __launch_bounds__(256) k_softmax_rows(const float* __restrict__ in, float* __restrict__ out, long rows, int C,
                                                       long ldi, long ldo) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
    const float* x = in + r * ldi;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(x[c] - m);
    const float inv = 1.f / s;
    for (int c = 0; c < C; ++c) out[r * ldo + c] = expf(x[c] - m) * inv;
  }
}

// Weight re-layout after an optimizer step, one pass: fp32 parameter matrix [R][C] -> bf16 copy dst[R][ldd] and (optional)
// bf16 transpose dstT[Cout][lddT]; both may be windows of direction-stacked operands.  perm_c > 0 additionally maps the
// reference's conv feature order to the internal one (out column j = f*perm_c + c  <-  source column c*perm_f + f) and
// zero-fills the columns [C, Cout).  64x64 tiles: coalesced 4-byte loads, row pairs packed into LDS words (33-word column
// stride: conflict-free), 16-byte transposed stores.  Bytes: 4 read + 2 (+2) written per element -- HBM-bound.
__global__ void __launch_bounds__(256) k_cast_transpose(const float* __restrict__ src, long lds_, int R, int C, int perm_c,
                                                        int perm_f, int Cout, uint16_t* __restrict__ dst, long ldd,
                                                        uint16_t* __restrict__ dstT, long lddT) {
  __shared__ uint32_t tileT[64 * 33];
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int j = c0 + tx;                       // output column
  int sc = j;                                  // source column
  if (perm_c > 0) sc = (j % perm_c) * perm_f + j / perm_c;
  const bool cvalid = j < C;                   // real data (pad columns are zero)
  if (!cvalid) sc = 0;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int r = r0 + ty * 16 + i;
    if (r >= R) r = R - 1;
    v[i] = src[(long)r * lds_ + sc];
  }
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    const float a = cvalid ? v[i] : 0.f, b = cvalid ? v[i + 1] : 0.f;
    const uint32_t pk = cvt_pk_bf16(a, b);
    const int r = r0 + ty * 16 + i;
    if (dst != nullptr && j < Cout) {
      if (r < R) dst[(long)r * ldd + j] = (uint16_t)(pk & 0xffffu);
      if (r + 1 < R) dst[(long)(r + 1) * ldd + j] = (uint16_t)(pk >> 16);
    }
    tileT[tx * 33 + ty * 8 + (i >> 1)] = pk;
  }
  if (dstT == nullptr) return;
  __syncthreads();
  const int col = tid >> 2, part = tid & 3;    // 16 rows (32 bytes) of one output row of the transpose
  const int jo = c0 + col, ro = r0 + part * 16;
  if (jo >= Cout || ro >= R) return;           // R % 16 == 0: a 16-row run is valid as a whole
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = tileT[col * 33 + part * 8 + k];
  uint4* o = reinterpret_cast<uint4*>(dstT + (long)jo * lddT + ro);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// Kernel layouts of the small weights, all in one launch (what the conv / head stages need every step after the optimizer moved
// the fp32 parameters): conv1 tap-major, conv2 tap-major (forward) and the two flipped row-parity sub-kernels (dgrad), the head
// weight zero-padded to 32 classes and its transpose.  Flat index over the concatenated outputs.
struct SmallLayouts {
  const float *w1, *w2, *wfc;     // conv.seq_module.0.weight (32,1,41,11), conv.seq_module.3.weight (32,32,21,11), fc weight (C,H)
  int C, H;
  float* w1k;                     // [451][32] f32: w1k[k][c] = w1[c][k]
  void* w2t;                      // [21][11][32 co][32 ci]  (T)
  void* w2d0;                     // [11][11][32 ci][32 co]  (T): rows kh = 0,2,..,20 flipped, columns flipped
  void* w2d1;                     // [10][11][32 ci][32 co]  (T): rows kh = 1,3,..,19 flipped
  void* wfcp;                     // [32][H] (T), rows >= C zero
  void* wfcT;                     // [H][32] (T)
};
template <typename T>
__global__ void __launch_bounds__(256) k_small_layouts(SmallLayouts a) {
  const long n1 = 451 * 32, n2 = 231L * 1024, n3 = 121L * 1024, n4 = 110L * 1024, n5 = 32L * a.H, n6 = 32L * a.H;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n1 + n2 + n3 + n4 + n5 + n6; i += (long)gridDim.x * 256) {
    long e = i;
    if (e < n1) {
      const int k = (int)(e / 32), c = (int)(e % 32);
      a.w1k[e] = a.w1[c * 451 + k];
      continue;
    }
    e -= n1;
    if (e < n2) {
      const int ci = (int)(e % 32), co = (int)((e / 32) % 32), tap = (int)(e / 1024);     // tap = kh*11 + kw
      stf((T*)a.w2t + e, a.w2[((long)co * 32 + ci) * 231 + tap]);
      continue;
    }
    e -= n2;
    if (e < n3 + n4) {
      const int q = e < n3 ? 0 : 1;
      const long f = q ? e - n3 : e;
      const int KHq = q ? 10 : 11;
      const int co = (int)(f % 32), ci = (int)((f / 32) % 32), b = (int)((f / 1024) % 11), r = (int)(f / (1024 * 11));
      const int kh = q + 2 * (KHq - 1 - r), kw = 10 - b;
      stf((T*)(q ? a.w2d1 : a.w2d0) + f, a.w2[((long)co * 32 + ci) * 231 + kh * 11 + kw]);
      continue;
    }
    e -= n3 + n4;
    if (e < n5) {
      const int c = (int)(e / a.H), h = (int)(e % a.H);
      stf((T*)a.wfcp + e, c < a.C ? a.wfc[(long)c * a.H + h] : 0.f);
      continue;
    }
    e -= n5;
    {
      const int h = (int)(e / 32), c = (int)(e % 32);
      stf((T*)a.wfcT + e, c < a.C ? a.wfc[(long)c * a.H + h] : 0.f);
    }
  }
}

// x[i] *= *s  (upstream gradient of the summed CTC loss, model.py:248, applied to the gradient computed with the loss)
__global__ void __launch_bounds__(256) k_scale_by(float* __restrict__ x, const float* __restrict__ s, long n) {
  const float v = *s;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= v;
}

inline int ew_grid(long n) {
  long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}


// Bias gradients of a recurrent layer from the BPTT sweep's per-sample sums (ds2_rnn_persist_bwd: dBacc [D][N][NB*H], NB = 4 for
// GRU -- dr, dz, dn, dq -- else G): bias_ih.grad [D][G*H] = sum over samples of planes 0..G-1, bias_hh.grad [D][G*H] = the same,
// except GRU: planes 0, 1, 3 (the n slot of the hidden side is dq = dn * r).  One thread per (direction, plane, unit); fixed
// summation order.
__global__ void __launch_bounds__(256) k_rnn_bias_grads(const float* __restrict__ bacc, int D, int N, int H, int G, int NB,
                                                        float* __restrict__ dbih, float* __restrict__ dbhh) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long per_d = (long)NB * H;
  if (i >= D * per_d) return;
  const int d = (int)(i / per_d), c = (int)(i % per_d), plane = c / H, u = c % H;
  const float* p = bacc + (long)d * N * per_d + c;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += p[(long)n * per_d];
  const long gh = (long)G * H;
  if (plane < G) dbih[d * gh + (long)plane * H + u] = s;
  if (NB == G)
    dbhh[d * gh + (long)plane * H + u] = s;
  else if (plane != 2)
    dbhh[d * gh + (long)(plane == 3 ? 2 : plane) * H + u] = s;
}

}  // namespace

// ---- fp32 operand -> three bf16 K-segments (fp32-class GEMMs on the bf16 matrix pipe) -------------------------------------------
// x = hi + lo (+ a residual below 2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi).  A product of two fp32 operands is then
// a_hi b_hi + a_hi b_lo + a_lo b_hi (+ terms below 2^-17 |a b|): with the A operand laid out as [hi | hi | lo] and the B operand as
// [hi | lo | hi] along K, ONE bf16 GEMM with K' = 3 Kp and fp32 accumulation yields the three sums.  The exact-fp32 MFMA runs at
// 1/16 of the bf16 rate (157 vs 2 500 TFLOP/s), three bf16 products at 1/3: config 2's GEMMs (fp32 parity mode, M = 808 rows) go
// from 37 TFLOP/s to ~5x that.  mode 0 = A pattern, 1 = B pattern; segments are Kp wide (K rounded up to the 64-element K-tile),
// zero beyond K.
__global__ void __launch_bounds__(256) k_split3(const float* __restrict__ src, long lds_, long rows, int K, int Kp, int mode,
                                                uint16_t* __restrict__ dst, long ldd) {
  const int chunks = Kp / 8;
  const long total = rows * chunks;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / chunks;
    const int c0 = (int)(e % chunks) * 8;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.f;
    if (c0 + 8 <= K) {
      const float4 a = *reinterpret_cast<const float4*>(src + r * lds_ + c0), b = *reinterpret_cast<const float4*>(src + r * lds_ + c0 + 4);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
      for (int i = 0; i < 8; ++i)
        if (c0 + i < K) x[i] = src[r * lds_ + c0 + i];
    }
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hi[i] = cvt_pk_bf16(x[2 * i], x[2 * i + 1]);
      const float h0 = __uint_as_float(hi[i] << 16), h1 = __uint_as_float(hi[i] & 0xffff0000u);
      lo[i] = cvt_pk_bf16(x[2 * i] - h0, x[2 * i + 1] - h1);
    }
    const uint4 H = make_uint4(hi[0], hi[1], hi[2], hi[3]), L = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    // modes 0 / 1: the three segments side by side in a row (K' = 3 Kp: operands of an NT product); modes 2 / 3 (round 6): stacked by
    // ROWS ([3 rows][ldd]: operands of a TN product, whose contraction index is the row -- the weight gradients over the activations
    // as stored, no transposes)
    const long seg = (mode & 2) ? rows * ldd : (long)Kp;
    const bool a_pat = (mode & 1) == 0;
    uint16_t* d = dst + r * ldd + c0;
    *reinterpret_cast<uint4*>(d) = H;
    *reinterpret_cast<uint4*>(d + seg) = a_pat ? H : L;
    *reinterpret_cast<uint4*>(d + 2 * seg) = a_pat ? L : H;
  }
}

// Zero the padding rows (t >= lens[n]) of a [T' x N] sequence matrix: one workgroup per group of 4 rows, 16 bytes per lane.
__global__ void __launch_bounds__(256) k_zero_pad_rows(unsigned char* X, long ld_bytes, int row_bytes, const int* __restrict__ lens, int N, long R) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int t = (int)(r / N), n = (int)(r % N);
  if (t < lens[n]) return;
  unsigned char* p = X + r * ld_bytes;
  for (int o = (threadIdx.x & 63) * 16; o < row_bytes; o += 64 * 16) *reinterpret_cast<uint4*>(p + o) = make_uint4(0, 0, 0, 0);
}

// Word copy between any two device-visible buffers -- the pinned host staging slots of the per-step int32 tables -> device memory,
// the persistent kernels' error word -> its pinned host mirror.  An in-stream kernel instead of hipMemcpyAsync: the runtime sends
// pinned copies to the SDMA engines, and with the host thread a few steps ahead of the device those transfers run UNDER the kernels of
// earlier steps -- a recurrent sweep hit by one took 2.1-3.0 ms instead of 1.0-1.3 (tools/step_jitter.py, profiles/r05d_*).
__global__ void __launch_bounds__(256) k_copy_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}

extern "C" {

int ds2_version(void) { return 100; }

int ds2_copy_words(const void* src, void* dst, long n_words, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(n_words >= 0 && (n_words == 0 || (src != nullptr && dst != nullptr)), DS2_ERR_ARG);
  DS2_REQUIRE((((uintptr_t)src) & 3) == 0 && (((uintptr_t)dst) & 3) == 0, DS2_ERR_ALIGN);
  if (n_words == 0) return 0;
  const long blocks = (n_words + 255) / 256;
  hipLaunchKernelGGL(k_copy_words, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, st, (const uint32_t*)src, (uint32_t*)dst, n_words);
  DS2_CHECK_LAUNCH();
  return 0;
}

const char* ds2_error_string(int code) {
  switch (code) {
    case DS2_OK: return "ok";
    case DS2_ERR_DTYPE: return "ds2hip: unknown dtype";
    case DS2_ERR_ARG: return "ds2hip: bad argument (dimension / null pointer / unsupported combination)";
    case DS2_ERR_ALIGN: return "ds2hip: pointer or leading dimension violates the 16-byte alignment contract";
    default: return hipGetErrorString((hipError_t)code);
  }
}

int ds2_add2(int dtype, const void* a, const void* b, void* out, long n, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  const int V = dtype == DS2_F32 ? 4 : 8;
  DS2_REQUIRE(n % V == 0, DS2_ERR_ALIGN);
  if (dtype == DS2_F32)
    hipLaunchKernelGGL(k_add2<float>, dim3(ew_grid(n / V)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, n / V);
  else
    hipLaunchKernelGGL(k_add2<bf16_t>, dim3(ew_grid(n / V)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n / V);
  DS2_CHECK_LAUNCH();
  return 0;
}

// out[n] f32 = sum over `slices` consecutive [n] blocks of src, added in index order (n % 4 == 0, 16-byte aligned).  The reduction of a
// product whose contraction was cut into K-slices (ds2_gemm_nt with the slices as its batch): deterministic, unlike atomic split-K.
int ds2_sum_slices(const float* src, float* out, long n, int slices, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(src != nullptr && out != nullptr && slices >= 1 && n > 0, DS2_ERR_ARG);
  DS2_REQUIRE(n % 4 == 0 && ((((uintptr_t)src) | ((uintptr_t)out)) & 15) == 0, DS2_ERR_ALIGN);
  hipLaunchKernelGGL(k_sum_slices, dim3(ew_grid(n / 4)), dim3(256), 0, st, src, out, n / 4, slices);
  DS2_CHECK_LAUNCH();
  return 0;
}

int ds2_zero_pad_rows(int dtype, void* X, long ld, int cols, const int* lens, int Tp, int N, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  const int es = dtype == DS2_F32 ? 4 : 2;
  DS2_REQUIRE(X != nullptr && lens != nullptr && Tp > 0 && N > 0 && cols > 0 && ld >= cols, DS2_ERR_ARG);
  DS2_REQUIRE((ld * es) % 16 == 0 && ((long)cols * es) % 16 == 0 && (((uintptr_t)X) & 15) == 0, DS2_ERR_ALIGN);
  const long R = (long)Tp * N;
  hipLaunchKernelGGL(k_zero_pad_rows, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, (unsigned char*)X, ld * es, cols * es, lens, N, R);
  DS2_CHECK_LAUNCH();
  return 0;
}

int ds2_split3_bf16(const float* src, long lds_, long rows, int K, int Kp, int mode, void* dst, long ldd, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(src != nullptr && dst != nullptr && rows > 0 && K > 0 && Kp >= K && (mode >= 0 && mode <= 3), DS2_ERR_ARG);
  DS2_REQUIRE((mode & 2) ? Kp % 8 == 0 : Kp % 64 == 0, DS2_ERR_ARG);
  DS2_REQUIRE(lds_ % 4 == 0 && ldd >= ((mode & 2) ? (long)Kp : 3L * Kp) && ldd % 8 == 0 && (((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 15) == 0, DS2_ERR_ALIGN);
  const long total = rows * (Kp / 8);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_split3, dim3(grid), dim3(256), 0, st, src, lds_, rows, K, Kp, mode, (uint16_t*)dst, ldd);
  DS2_CHECK_LAUNCH();
  return 0;
}

int ds2_transpose(int dtype, const void* src, void* dst, long R, int C, long lds_, long ldd, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  const int V = dtype == DS2_F32 ? 4 : 8;
  DS2_REQUIRE(C % V == 0 && lds_ % V == 0 && ldd % V == 0 && ldd >= R, DS2_ERR_ALIGN);
  dim3 grid(ds2_cdiv(C, 64), ds2_cdiv(ldd, 64));
  if (dtype == DS2_F32)
    hipLaunchKernelGGL(k_transpose<float>, grid, dim3(256), 0, st, (const float*)src, (float*)dst, R, C, lds_, ldd);
  else
    hipLaunchKernelGGL(k_transpose<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)src, (uint16_t*)dst, R, C, lds_, ldd);
  DS2_CHECK_LAUNCH();
  return 0;
}

int ds2_cast_transpose_bf16(const float* src, long lds_, int R, int C, int perm_c, int perm_f, int Cout, void* dst, long ldd,
                            void* dstT, long lddT, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(src != nullptr && R > 0 && C > 0 && Cout >= C && (dst != nullptr || dstT != nullptr), DS2_ERR_ARG);
  DS2_REQUIRE(perm_c == 0 || (perm_c > 0 && perm_f > 0 && perm_c * perm_f == C), DS2_ERR_ARG);
  DS2_REQUIRE(R % 16 == 0, DS2_ERR_ALIGN);
  if (dst != nullptr) DS2_REQUIRE(ldd >= Cout, DS2_ERR_ARG);
  if (dstT != nullptr) DS2_REQUIRE(lddT >= R && lddT % 8 == 0 && ((uintptr_t)dstT & 15) == 0, DS2_ERR_ALIGN);
  hipLaunchKernelGGL(k_cast_transpose, dim3(ds2_cdiv(Cout, 64), ds2_cdiv(R, 64)), dim3(256), 0, st, src, lds_, R, C, perm_c,
                     perm_f, Cout, (uint16_t*)dst, ldd, (uint16_t*)dstT, lddT);
  DS2_CHECK_LAUNCH();
  return 0;
}

int ds2_small_weight_layouts(int dtype, const float* w1, const float* w2, const float* wfc, int C, int H, float* w1k, void* w2t,
                             void* w2d0, void* w2d1, void* wfcp, void* wfcT, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  DS2_REQUIRE(w1 && w2 && wfc && w1k && w2t && w2d0 && w2d1 && wfcp && wfcT && C > 0 && C <= 32 && H > 0, DS2_ERR_ARG);
  SmallLayouts a{w1, w2, wfc, C, H, w1k, w2t, w2d0, w2d1, wfcp, wfcT};
  const long total = 451 * 32 + 231L * 1024 + 231L * 1024 + 64L * H;
  if (dtype == DS2_F32)
    hipLaunchKernelGGL(k_small_layouts<float>, dim3(ew_grid(total)), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(k_small_layouts<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, st, a);
  DS2_CHECK_LAUNCH();
  return 0;
}

int ds2_scale_by(float* x, const float* s, long n, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(x && s && n > 0, DS2_ERR_ARG);
  hipLaunchKernelGGL(k_scale_by, dim3(ew_grid(n)), dim3(256), 0, st, x, s, n);
  DS2_CHECK_LAUNCH();
  return 0;
}

static int la_segments(int Tp) { return ds2_cdiv(Tp, LA_SEG_STEPS * LA_CTX); }
// y = hardtanh(lookahead(x)); `pre` (same shape, may be null in eval) keeps the pre-activation for backward
int ds2_lookahead_fwd(int dtype, const void* x, const float* w, void* y, void* pre, int Tp, int N, int H, int ctx,
                      ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  const int V = dtype == DS2_F32 ? 4 : 8;
  DS2_REQUIRE(H % V == 0 && ctx > 0, DS2_ERR_ARG);
  const long total = (long)ds2_cdiv(Tp, LA_TB) * N * (H / V);
  if (ctx == LA_CTX) {
    const int grid = ds2_cdiv((long)la_segments(Tp) * N * (H / 4), 256);
    if (dtype == DS2_F32)
      hipLaunchKernelGGL((k_lookahead_fwd_slide<float, LA_CTX>), dim3(grid), dim3(256), 0, st, (const float*)x, w, (float*)y, (float*)pre, Tp, N, H);
    else
      hipLaunchKernelGGL((k_lookahead_fwd_slide<bf16_t, LA_CTX>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, w, (bf16_t*)y, (bf16_t*)pre, Tp, N, H);
    DS2_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == DS2_F32)
    hipLaunchKernelGGL(k_lookahead_fwd<float>, dim3(ew_grid(total)), dim3(256), 0, st, (const float*)x, w, (float*)y, (float*)pre, Tp, N, H, ctx);
  else
    hipLaunchKernelGGL(k_lookahead_fwd<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, st, (const bf16_t*)x, w, (bf16_t*)y, (bf16_t*)pre, Tp, N, H, ctx);
  DS2_CHECK_LAUNCH();
  return 0;
}

static int la_row_blocks(int Tp, int N) { return ds2_cdiv((long)Tp * N, LA_ROWS); }
long ds2_lookahead_ws_floats(int Tp, int N, int H, int ctx) {
  const long P = ctx == LA_CTX ? (long)la_segments(Tp) * N : la_row_blocks(Tp, N);
  return P * H * ctx + (long)ds2_norm_partials(P) * H * ctx;
}
int ds2_lookahead_bwd(int dtype, const void* x, const float* w, const void* pre, const void* dy, void* dx, float* dw, int Tp,
                      int N, int H, int ctx, float* ws, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  const int V = dtype == DS2_F32 ? 4 : 8;
  DS2_REQUIRE(H % V == 0 && ctx > 0 && (H * ctx) % 4 == 0, DS2_ERR_ARG);
  const long total = (long)ds2_cdiv(Tp, LA_TB) * N * (H / V);
  const int C = H * ctx;
  if (ctx == LA_CTX) {
    const int Ps = la_segments(Tp) * N, grid = ds2_cdiv((long)Ps * (H / 4), 256);
    if (dtype == DS2_F32) {
      hipLaunchKernelGGL((k_lookahead_bwd_x_slide<float, LA_CTX>), dim3(grid), dim3(256), 0, st, (const float*)dy, (const float*)pre, w, (float*)dx, Tp, N, H);
      hipLaunchKernelGGL((k_lookahead_bwd_w_slide<float, LA_CTX>), dim3(grid), dim3(256), 0, st, (const float*)dy, (const float*)pre, (const float*)x, ws, Tp, N, H);
    } else {
      hipLaunchKernelGGL((k_lookahead_bwd_x_slide<bf16_t, LA_CTX>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)pre, w, (bf16_t*)dx, Tp, N, H);
      hipLaunchKernelGGL((k_lookahead_bwd_w_slide<bf16_t, LA_CTX>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)pre, (const bf16_t*)x, ws, Tp, N, H);
    }
    DS2_CHECK_LAUNCH();
    return ds2_colsum(DS2_F32, ws, Ps, C, C, dw, 1.0f, ws + (long)Ps * C, st_);
  }
  const int P = la_row_blocks(Tp, N);
  dim3 gw(ds2_cdiv(H, 1024), P);                           // a thread of k_lookahead_bwd_w owns four channels
  if (dtype == DS2_F32) {
    hipLaunchKernelGGL(k_lookahead_bwd_x<float>, dim3(ew_grid(total)), dim3(256), 0, st, (const float*)dy, (const float*)pre, w, (float*)dx, Tp, N, H, ctx);
    hipLaunchKernelGGL(k_lookahead_bwd_w<float>, gw, dim3(256), 0, st, (const float*)dy, (const float*)pre, (const float*)x, ws, Tp, N, H, ctx);
  } else {
    hipLaunchKernelGGL(k_lookahead_bwd_x<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)pre, w, (bf16_t*)dx, Tp, N, H, ctx);
    hipLaunchKernelGGL(k_lookahead_bwd_w<bf16_t>, gw, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)pre, (const bf16_t*)x, ws, Tp, N, H, ctx);
  }
  DS2_CHECK_LAUNCH();
  return ds2_colsum(DS2_F32, ws, P, C, C, dw, 1.0f, ws + (long)P * C, st_);
}

int ds2_rnn_bias_grads(int cell, int D, int N, int H, const float* dBacc, float* dbih, float* dbhh, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dBacc && dbih && dbhh && D >= 1 && D <= 2 && N > 0 && H > 0 && cell >= 0 && cell <= 2, DS2_ERR_ARG);
  const int G = cell == 0 ? 3 : cell == 1 ? 4 : 1, NB = cell == 0 ? 4 : G;
  hipLaunchKernelGGL(k_rnn_bias_grads, dim3(ds2_cdiv((long)D * NB * H, 256)), dim3(256), 0, st, dBacc, D, N, H, G, NB, dbih, dbhh);
  DS2_CHECK_LAUNCH();
  return 0;
}

int ds2_softmax_rows(const float* logits, float* probs, long rows, int C, long ld_in, long ld_out, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(rows > 0 && C > 0, DS2_ERR_ARG);
  hipLaunchKernelGGL(k_softmax_rows, dim3(ew_grid(rows)), dim3(256), 0, st, logits, probs, rows, C, ld_in, ld_out);
  DS2_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
