// Optimizer step of the training loop as hand-written multi-tensor kernels: global-norm gradient
// clipping (Lightning's gradient_clip_val: 400, reference configs/an4.yaml:12 -> torch.nn.utils.clip_grad_norm_) fused with
// AdamW (reference model.py:283-289) or SGD with Nesterov momentum (model.py:275-281), and -- for the recurrent weight
// matrices, 98 % of the parameters -- the bf16 operand copies and transposes the next forward / backward need, written in
// the same pass (what nn.GRU.flatten_parameters + the autocast weight casts do in the reference, model.py:97-99).
//
// HBM-bound: per parameter 16 B read (p, g, m, v) + 12 B written (p, m, v) [+ 4 B of bf16 layouts]; the clip coefficient
// is produced on the device (no host synchronisation) by a deterministic two-stage sum of squares.
// Arithmetic follows torch's single-tensor implementations statement by statement (torch/optim/adam.py _single_tensor_adam,
// torch/optim/sgd.py _single_tensor_sgd) in fp32; the scalars (1 - lr*wd, bias corrections, step size) are computed by the
// caller in double exactly as torch does and passed as floats.
#include "ds2_common.h"

namespace {

constexpr int MAXT = 56;          // tensors per multi-tensor launch (the table travels as a kernel argument: < 4 KB)
constexpr int CHUNK = 16384;      // elements per workgroup pass

struct Hyper {
  int mode;                       // 0 AdamW, 1 SGD (Nesterov)
  float decay;                    // AdamW: 1 - lr*wd (param.mul_)          SGD: wd (grad += wd*param)
  float w1;                       // AdamW: 1 - beta1 (lerp weight)          SGD: momentum
  float beta2, w2;                // AdamW: beta2, 1 - beta2
  float bc2_sqrt, eps;            // AdamW: sqrt(1 - beta2^t), eps
  float neg_step;                 // AdamW: -(lr / (1 - beta1^t))            SGD: -lr
  int first;                      // SGD: 1 on the first step (momentum buffer = grad)
  const float* clip;              // device: clip[1] = coefficient the gradients are scaled by (null: 1)
};

__device__ __forceinline__ void update(const Hyper& h, float cs, float& p, float g, float& m, float& v) {
  g *= cs;                                        // clip_grad_norm_: grad.mul_(clip_coef_clamped)
  if (h.mode == 0) {
    p *= h.decay;                                 // param.mul_(1 - lr * weight_decay)
    m = m + h.w1 * (g - m);                       // exp_avg.lerp_(grad, 1 - beta1)
    v = v * h.beta2;                              // exp_avg_sq.mul_(beta2)
    v = v + h.w2 * g * g;                         //           .addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    p = p + h.neg_step * m / denom;               // param.addcdiv_(exp_avg, denom, value = -step_size)
  } else {
    g = g + h.decay * p;                          // grad.add(param, alpha = weight_decay)
    m = h.first ? g : m * h.w1 + g;               // buf = grad | buf.mul_(momentum).add_(grad)
    g = g + h.w1 * m;                             // nesterov: grad.add(buf, alpha = momentum)
    p = p + h.neg_step * g;                       // param.add_(grad, alpha = -lr)
  }
}

struct SumsqTable {
  const float* g[MAXT];
  long n[MAXT];
  int first_block[MAXT + 1];      // workgroup range of tensor i
  int count;
};

// stage 1: workgroup partial sums of squares (fixed order inside the workgroup)
__global__ void __launch_bounds__(256) k_sumsq(SumsqTable t, float* __restrict__ partials, int base) {
  __shared__ float red[4];
  int b = blockIdx.x, i = 0;
  while (i + 1 < t.count && b >= t.first_block[i + 1]) ++i;
  const long off = (long)(b - t.first_block[i]) * CHUNK;
  const float* g = t.g[i] + off;
  const long n = min((long)CHUNK, t.n[i] - off);
  float s = 0.f;
  if ((((uintptr_t)g) & 15) == 0) {               // 16-byte loads (chunks start at multiples of 16384 elements)
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const long n4 = n >> 2;
    for (long e = threadIdx.x; e < n4; e += 256) {
      const float4 x = g4[e];
      s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    }
    for (long e = (n4 << 2) + threadIdx.x; e < n; e += 256) s += g[e] * g[e];
  } else {
    for (long e = threadIdx.x; e < n; e += 256) s += g[e] * g[e];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[base + b] = (red[0] + red[1]) + (red[2] + red[3]);
}
// stage 2: out[0] = total L2 norm, out[1] = min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ void __launch_bounds__(256) k_clip_coef(const float* __restrict__ partials, int n, float max_norm, float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int e = threadIdx.x; e < n; e += 256) s += (double)partials[e];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
    const float c = max_norm / (norm + 1e-6f);
    out[0] = norm;
    out[1] = max_norm > 0.f ? (c < 1.f ? c : 1.f) : 1.f;
  }
}

struct OptTable {
  float* p[MAXT];
  const float* g[MAXT];
  float* m[MAXT];
  float* v[MAXT];                 // AdamW only
  long n[MAXT];
  int first_block[MAXT + 1];
  int count;
};

__global__ void __launch_bounds__(256) k_opt_multi(OptTable t, Hyper h) {
  int b = blockIdx.x, i = 0;
  while (i + 1 < t.count && b >= t.first_block[i + 1]) ++i;
  const long off = (long)(b - t.first_block[i]) * CHUNK;
  const long n = min((long)CHUNK, t.n[i] - off);
  float* p = t.p[i] + off;
  const float* g = t.g[i] + off;
  float* m = t.m[i] + off;
  float* v = h.mode == 0 ? t.v[i] + off : nullptr;
  const float cs = h.clip ? h.clip[1] : 1.f;
  for (long e = threadIdx.x; e < n; e += 256) {
    float pv = p[e], mv = m[e], vv = h.mode == 0 ? v[e] : 0.f;
    update(h, cs, pv, g[e], mv, vv);
    p[e] = pv;
    m[e] = mv;
    if (h.mode == 0) v[e] = vv;
  }
}

// One recurrent weight matrix p[R][C] (row stride = C, reference layout): update + bf16 copy dst[R][ldd] and/or bf16 transpose
// dstT[Cout][lddT].  perm_c > 0: output column j = f*perm_c + c takes source column c*perm_f + f (rnns.0 reads the conv
// features in the kernels' [f][c] order, see ds2_cast_transpose_bf16); output columns [C, Cout) are zero.  64x64 tiles through
// LDS; every source element is visited exactly once (the column map is a bijection on [0, C)).
__global__ void __launch_bounds__(256) k_opt_matrix(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int R, int C, int perm_c, int perm_f, int Cout,
                                                    uint16_t* __restrict__ dst, long ldd, uint16_t* __restrict__ dstT, long lddT,
                                                    Hyper h) {
  __shared__ uint32_t tileT[64 * 33];
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int j = c0 + tx;                       // output column
  int sc = j;                                  // source column
  if (perm_c > 0) sc = (j % perm_c) * perm_f + j / perm_c;
  const bool cvalid = j < C;
  const float cs = h.clip ? h.clip[1] : 1.f;
  float val[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty * 16 + i;
    val[i] = 0.f;
    if (cvalid && r < R) {
      const long o = (long)r * C + sc;
      float pv = p[o], mv = m[o], vv = h.mode == 0 ? v[o] : 0.f;
      update(h, cs, pv, g[o], mv, vv);
      p[o] = pv;
      m[o] = mv;
      if (h.mode == 0) v[o] = vv;
      val[i] = pv;
    }
  }
  if (dst == nullptr && dstT == nullptr) return;
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    const uint32_t pk = cvt_pk_bf16(val[i], val[i + 1]);
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

// Vectorised form for the un-permuted matrices (C % 4 == 0, Cout == C): a thread owns a 4-row x 4-column patch of the 64x64
// tile -- 16-byte loads / stores of p, g, m, v, 8-byte stores of the bf16 copy, the same LDS image for the transpose.
__device__ __forceinline__ void opt_matrix4_tile(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, int R, int C, uint16_t* __restrict__ dst, long ldd,
                                                 uint16_t* __restrict__ dstT, long lddT, const Hyper& h, int bx, int by,
                                                 uint32_t* tileT) {
  const int tid = threadIdx.x, cx = tid & 15, ry = tid >> 4;         // 16 column quads x 16 row quads
  const int c0 = bx * 64 + cx * 4, r0 = by * 64 + ry * 4;
  const float cs = h.clip ? h.clip[1] : 1.f;
  float val[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) val[i][e] = 0.f;
  if (c0 < C) {
    float4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long o = (long)min(r0 + i, R - 1) * C + c0;
      pv[i] = *reinterpret_cast<const float4*>(p + o);
      gv[i] = *reinterpret_cast<const float4*>(g + o);
      mv[i] = *reinterpret_cast<const float4*>(m + o);
      vv[i] = h.mode == 0 ? *reinterpret_cast<const float4*>(v + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (r0 + i >= R) continue;
      float* pe = reinterpret_cast<float*>(&pv[i]);
      float* ge = reinterpret_cast<float*>(&gv[i]);
      float* me = reinterpret_cast<float*>(&mv[i]);
      float* ve = reinterpret_cast<float*>(&vv[i]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        update(h, cs, pe[e], ge[e], me[e], ve[e]);
        val[i][e] = pe[e];
      }
      const long o = (long)(r0 + i) * C + c0;
      *reinterpret_cast<float4*>(p + o) = pv[i];
      *reinterpret_cast<float4*>(m + o) = mv[i];
      if (h.mode == 0) *reinterpret_cast<float4*>(v + o) = vv[i];
      if (dst != nullptr) {
        uint2 pk;
        pk.x = cvt_pk_bf16(val[i][0], val[i][1]);
        pk.y = cvt_pk_bf16(val[i][2], val[i][3]);
        *reinterpret_cast<uint2*>(dst + (long)(r0 + i) * ldd + c0) = pk;
      }
    }
  }
  if (dstT == nullptr) return;
  // LDS image: tileT[column][row pair] (bf16 x 2 packed along the rows), the layout the transposed 32-byte stores read
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    tileT[(cx * 4 + e) * 33 + ry * 2] = cvt_pk_bf16(val[0][e], val[1][e]);
    tileT[(cx * 4 + e) * 33 + ry * 2 + 1] = cvt_pk_bf16(val[2][e], val[3][e]);
  }
  __syncthreads();
  const int col = tid >> 2, part = tid & 3;
  const int jo = bx * 64 + col, ro = by * 64 + part * 16;
  if (jo >= C || ro >= R) return;
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = tileT[col * 33 + part * 8 + k];
  uint4* o = reinterpret_cast<uint4*>(dstT + (long)jo * lddT + ro);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

__global__ void __launch_bounds__(256) k_opt_matrix4(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, int R, int C, uint16_t* __restrict__ dst, long ldd,
                                                     uint16_t* __restrict__ dstT, long lddT, Hyper h) {
  __shared__ uint32_t tileT[64 * 33];
  opt_matrix4_tile(p, g, m, v, R, C, dst, ldd, dstT, lddT, h, blockIdx.x, blockIdx.y, tileT);
}

// Round 6: ALL un-permuted weight matrices of a parameter group in ONE launch (cfg3: 18 launches of ~22 us -> one; every launch
// boundary drained and refilled the memory pipeline of a purely HBM-bound pass).  Workgroup -> (matrix, 64 x 64 tile): the matrices
// own consecutive id ranges, tiles row-major (column tile fastest: consecutive workgroups stream consecutive 256-byte row segments).
constexpr int MAXM = 36;          // matrices per launch (the table is a kernel argument: 36 x 84 B + Hyper < 4 KB)
struct MatTable {
  float* p[MAXM];
  const float* g[MAXM];
  float* m[MAXM];
  float* v[MAXM];
  uint16_t* dst[MAXM];
  uint16_t* dstT[MAXM];
  long ldd[MAXM], lddT[MAXM];
  int R[MAXM], C[MAXM];
  int first_block[MAXM + 1];
  int count;
};
__global__ void __launch_bounds__(256) k_opt_matrix4_multi(MatTable t, Hyper h) {
  __shared__ uint32_t tileT[64 * 33];
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < t.count && b >= t.first_block[i + 1]) ++i;
  const int local = b - t.first_block[i], tc = (t.C[i] + 63) / 64;
  opt_matrix4_tile(t.p[i], t.g[i], t.m[i], t.v[i], t.R[i], t.C[i], t.dst[i], t.ldd[i], t.dstT[i], t.lddT[i], h, local % tc, local / tc, tileT);
}

// Permuted form (rnns.0 weight_ih: output column j = f*perm_c + c <- source column c*perm_f + f, zero columns [C, Cout)): tiles run
// over the SOURCE columns so that the 28 B/element of fp32 traffic (p, g, m, v) stays in 16-byte coalesced accesses; only the bf16
// copy is scattered (2-byte stores), the transpose leaves through LDS as 32-byte runs.  C % 4 == 0.
__global__ void __launch_bounds__(256) k_opt_matrix_perm(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, int R, int C, int perm_c, int perm_f, int Cout,
                                                         uint16_t* __restrict__ dst, long ldd, uint16_t* __restrict__ dstT, long lddT,
                                                         Hyper h) {
  __shared__ uint32_t tileT[64 * 33];
  const int tid = threadIdx.x, cx = tid & 15, ry = tid >> 4;
  const int s0 = blockIdx.x * 64 + cx * 4, r0 = blockIdx.y * 64 + ry * 4;     // source column quad, row quad
  const float cs = h.clip ? h.clip[1] : 1.f;
  float val[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) val[i][e] = 0.f;
  if (s0 < C) {
    float4 pv[4], gv[4], mv[4], vv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long o = (long)min(r0 + i, R - 1) * C + s0;
      pv[i] = *reinterpret_cast<const float4*>(p + o);
      gv[i] = *reinterpret_cast<const float4*>(g + o);
      mv[i] = *reinterpret_cast<const float4*>(m + o);
      vv[i] = h.mode == 0 ? *reinterpret_cast<const float4*>(v + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int jcol[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) jcol[e] = ((s0 + e) % perm_f) * perm_c + (s0 + e) / perm_f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (r0 + i >= R) continue;
      float* pe = reinterpret_cast<float*>(&pv[i]);
      float* ge = reinterpret_cast<float*>(&gv[i]);
      float* me = reinterpret_cast<float*>(&mv[i]);
      float* ve = reinterpret_cast<float*>(&vv[i]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        update(h, cs, pe[e], ge[e], me[e], ve[e]);
        val[i][e] = pe[e];
      }
      const long o = (long)(r0 + i) * C + s0;
      *reinterpret_cast<float4*>(p + o) = pv[i];
      *reinterpret_cast<float4*>(m + o) = mv[i];
      if (h.mode == 0) *reinterpret_cast<float4*>(v + o) = vv[i];
      if (dst != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(long)(r0 + i) * ldd + jcol[e]] = (uint16_t)cvt_pk_bf16(val[i][e], 0.f);
      }
    }
  }
  // zero pad columns [C, Cout) of the rows of this tile row (done by the first column block)
  if (blockIdx.x == 0 && Cout > C) {
    const int npad = Cout - C;
    for (int e = tid; e < 64 * npad; e += 256) {
      const int r = blockIdx.y * 64 + e / npad, j = C + e % npad;
      if (r < R) {
        if (dst != nullptr) dst[(long)r * ldd + j] = 0;
        if (dstT != nullptr) dstT[(long)j * lddT + r] = 0;
      }
    }
  }
  if (dstT == nullptr) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    tileT[(cx * 4 + e) * 33 + ry * 2] = cvt_pk_bf16(val[0][e], val[1][e]);
    tileT[(cx * 4 + e) * 33 + ry * 2 + 1] = cvt_pk_bf16(val[2][e], val[3][e]);
  }
  __syncthreads();
  const int col = tid >> 2, part = tid & 3;
  const int sc = blockIdx.x * 64 + col, ro = blockIdx.y * 64 + part * 16;
  if (sc >= C || ro >= R) return;
  const int jo = (sc % perm_f) * perm_c + sc / perm_f;
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = tileT[col * 33 + part * 8 + k];
  uint4* o = reinterpret_cast<uint4*>(dstT + (long)jo * lddT + ro);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

Hyper make_hyper(int mode, const float* hp, int first, const float* clip) {
  Hyper h{};
  h.mode = mode;
  h.decay = hp[0]; h.w1 = hp[1]; h.beta2 = hp[2]; h.w2 = hp[3]; h.bc2_sqrt = hp[4]; h.eps = hp[5]; h.neg_step = hp[6];
  h.first = first;
  h.clip = clip;
  return h;
}

}  // namespace

extern "C" {

int ds2_opt_max_tensors(void) { return MAXT; }
// floats of `partials` scratch the clip pass needs for these tensor sizes
long ds2_clip_ws_floats(int count, const long* n) {
  long b = 0;
  for (int i = 0; i < count; ++i) b += (n[i] + CHUNK - 1) / CHUNK;
  return b + 2;
}

// out[0] = ||g||_2 over ALL `count` gradient tensors, out[1] = min(1, max_norm / (out[0] + 1e-6)); device-side, no sync.
// ws: ds2_clip_ws_floats() floats.  Deterministic (fixed summation order).
int ds2_clip_coef(int count, const float* const* g, const long* n, float max_norm, float* out, float* ws, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(count > 0 && g && n && out && ws, DS2_ERR_ARG);
  int base = 0;
  for (int i0 = 0; i0 < count; i0 += MAXT) {
    SumsqTable t{};
    t.count = count - i0 < MAXT ? count - i0 : MAXT;
    int blocks = 0;
    for (int i = 0; i < t.count; ++i) {
      DS2_REQUIRE(g[i0 + i] && n[i0 + i] > 0, DS2_ERR_ARG);
      t.g[i] = g[i0 + i];
      t.n[i] = n[i0 + i];
      t.first_block[i] = blocks;
      blocks += (int)((n[i0 + i] + CHUNK - 1) / CHUNK);
    }
    t.first_block[t.count] = blocks;
    hipLaunchKernelGGL(k_sumsq, dim3(blocks), dim3(256), 0, st, t, ws, base);
    DS2_CHECK_LAUNCH();
    base += blocks;
  }
  hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(256), 0, st, (const float*)ws, base, max_norm, out);
  DS2_CHECK_LAUNCH();
  return 0;
}

// mode 0 AdamW / 1 SGD-Nesterov over `count` flat fp32 tensors (v ignored for SGD).  hp = {decay, w1, beta2, w2, bc2_sqrt, eps,
// neg_step} as documented at struct Hyper; clip = device float[2] from ds2_clip_coef or null.
int ds2_opt_multi(int mode, int count, float* const* p, const float* const* g, float* const* m, float* const* v, const long* n,
                  const float* hp, int first, const float* clip, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE((mode == 0 || mode == 1) && count > 0 && p && g && m && n && hp && (mode == 1 || v), DS2_ERR_ARG);
  const Hyper h = make_hyper(mode, hp, first, clip);
  for (int i0 = 0; i0 < count; i0 += MAXT) {
    OptTable t{};
    t.count = count - i0 < MAXT ? count - i0 : MAXT;
    int blocks = 0;
    for (int i = 0; i < t.count; ++i) {
      DS2_REQUIRE(p[i0 + i] && g[i0 + i] && m[i0 + i] && n[i0 + i] > 0, DS2_ERR_ARG);
      t.p[i] = p[i0 + i]; t.g[i] = g[i0 + i]; t.m[i] = m[i0 + i]; t.v[i] = mode == 0 ? v[i0 + i] : nullptr; t.n[i] = n[i0 + i];
      t.first_block[i] = blocks;
      blocks += (int)((n[i0 + i] + CHUNK - 1) / CHUNK);
    }
    t.first_block[t.count] = blocks;
    hipLaunchKernelGGL(k_opt_multi, dim3(blocks), dim3(256), 0, st, t, h);
    DS2_CHECK_LAUNCH();
  }
  return 0;
}

static bool opt_matrix_vec_ok(int mode, const float* p, const float* g, const float* m, const float* v, int C, int perm_c, int Cout,
                              const void* dst, long ldd) {
  return perm_c == 0 && Cout == C && C % 4 == 0 && (dst == nullptr || ldd % 4 == 0) &&
         (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)(mode == 0 ? v : p) | (uintptr_t)dst) & 15) == 0;
}

// `count` weight matrices of ONE parameter group (same hp / first): ds2_opt_matrix for each of them, with every matrix the
// vectorised kernel takes (no column permutation, Cout == C, C % 4 == 0, 16-byte aligned) in one launch per MAXM of them.
int ds2_opt_matrices(int mode, int count, float* const* p, const float* const* g, float* const* m, float* const* v, const int* R,
                     const int* C, const int* perm_c, const int* perm_f, const int* Cout, void* const* dst, const long* ldd,
                     void* const* dstT, const long* lddT, const float* hp, int first, const float* clip, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE((mode == 0 || mode == 1) && count > 0 && p && g && m && R && C && perm_c && perm_f && Cout && dst && ldd && dstT && lddT && hp &&
                  (mode == 1 || v), DS2_ERR_ARG);
  const Hyper h = make_hyper(mode, hp, first, clip);
  MatTable t{};
  int blocks = 0;
  auto flush = [&]() -> int {
    if (t.count == 0) return 0;
    t.first_block[t.count] = blocks;
    hipLaunchKernelGGL(k_opt_matrix4_multi, dim3(blocks), dim3(256), 0, st, t, h);
    DS2_CHECK_LAUNCH();
    t.count = 0;
    blocks = 0;
    return 0;
  };
  for (int i = 0; i < count; ++i) {
    float* vi = mode == 0 ? v[i] : nullptr;
    DS2_REQUIRE(p[i] && g[i] && m[i] && (mode == 1 || vi), DS2_ERR_ARG);
    DS2_REQUIRE(R[i] > 0 && C[i] > 0 && R[i] % 16 == 0 && Cout[i] >= C[i], DS2_ERR_ARG);
    DS2_REQUIRE(perm_c[i] == 0 || perm_c[i] * perm_f[i] == C[i], DS2_ERR_ARG);
    DS2_REQUIRE(dstT[i] == nullptr || (lddT[i] % 8 == 0 && (((uintptr_t)dstT[i]) & 15) == 0), DS2_ERR_ALIGN);
    if (!opt_matrix_vec_ok(mode, p[i], g[i], m[i], vi, C[i], perm_c[i], Cout[i], dst[i], ldd[i])) {
      const int rc = ds2_opt_matrix(mode, p[i], g[i], m[i], vi, R[i], C[i], perm_c[i], perm_f[i], Cout[i], dst[i], ldd[i], dstT[i], lddT[i], hp,
                                    first, clip, st_);
      if (rc != 0) return rc;
      continue;
    }
    const int k = t.count++;
    t.p[k] = p[i]; t.g[k] = g[i]; t.m[k] = m[i]; t.v[k] = vi; t.dst[k] = (uint16_t*)dst[i]; t.dstT[k] = (uint16_t*)dstT[i];
    t.ldd[k] = ldd[i]; t.lddT[k] = lddT[i]; t.R[k] = R[i]; t.C[k] = C[i];
    t.first_block[k] = blocks;
    blocks += ds2_cdiv(C[i], 64) * ds2_cdiv(R[i], 64);
    if (t.count == MAXM) {
      const int rc = flush();
      if (rc != 0) return rc;
    }
  }
  return flush();
}

// One weight matrix p[R][C] (contiguous): update + bf16 layouts (see k_opt_matrix).  R % 16 == 0.
int ds2_opt_matrix(int mode, float* p, const float* g, float* m, float* v, int R, int C, int perm_c, int perm_f, int Cout,
                   void* dst, long ldd, void* dstT, long lddT, const float* hp, int first, const float* clip, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE((mode == 0 || mode == 1) && p && g && m && hp && (mode == 1 || v), DS2_ERR_ARG);
  DS2_REQUIRE(R > 0 && C > 0 && R % 16 == 0 && Cout >= C, DS2_ERR_ARG);
  DS2_REQUIRE(perm_c == 0 || perm_c * perm_f == C, DS2_ERR_ARG);
  DS2_REQUIRE(dstT == nullptr || (lddT % 8 == 0 && (((uintptr_t)dstT) & 15) == 0), DS2_ERR_ALIGN);
  const Hyper h = make_hyper(mode, hp, first, clip);
  const bool vec = opt_matrix_vec_ok(mode, p, g, m, v, C, perm_c, Cout, dst, ldd);
  if (vec)
    hipLaunchKernelGGL(k_opt_matrix4, dim3(ds2_cdiv(C, 64), ds2_cdiv(R, 64)), dim3(256), 0, st, p, g, m, v, R, C, (uint16_t*)dst, ldd,
                       (uint16_t*)dstT, lddT, h);
  else if (perm_c > 0 && C % 4 == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)(mode == 0 ? v : p)) & 15) == 0)
    hipLaunchKernelGGL(k_opt_matrix_perm, dim3(ds2_cdiv(C, 64), ds2_cdiv(R, 64)), dim3(256), 0, st, p, g, m, v, R, C, perm_c, perm_f,
                       Cout, (uint16_t*)dst, ldd, (uint16_t*)dstT, lddT, h);
  else
    hipLaunchKernelGGL(k_opt_matrix, dim3(ds2_cdiv(Cout, 64), ds2_cdiv(R, 64)), dim3(256), 0, st, p, g, m, v, R, C, perm_c, perm_f,
                       Cout, (uint16_t*)dst, ldd, (uint16_t*)dstT, lddT, h);
  DS2_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
