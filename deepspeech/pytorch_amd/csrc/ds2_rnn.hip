// Recurrent sweep of BatchRNN (reference model.py:94-102: pack_padded_sequence -> nn.GRU/LSTM/RNN -> pad -> direction sum),
// forward and BPTT, both directions in one launch per time step.
//
// Design (SURVEY.md section 7 "option A", the correctness vehicle of round 1): the input projection for ALL time steps
// is hoisted out (one MFMA GEMM, ds2_gemm_nt); per time step one kernel does, for a slice of 16 hidden units x all
// gates x the whole minibatch, the skinny recurrent GEMM  gh = h_{t-1} * W_hh^T  on MFMA 16x16 tiles (batch rows are
// the M dimension, K split over the 4 waves of the workgroup, W_hh slice streamed from the XCD-local L2 straight
// into MFMA operand registers, cross-wave reduction through LDS) followed by the fused gate nonlinearity, the
// packed-sequence masking, the state update and all stores.  grid = (H/16, directions, batch groups of 32).
//
// Packed-sequence semantics restated with masks (oracle: rnn_dir_fwd): sample n is active at time t iff t < len[n];
// inactive samples carry their state unchanged and emit 0 (pad_packed_sequence); the reverse direction walks
// t = T'-1..0, so every sample starts at its OWN last frame from the initial state.
// Gate order follows torch: GRU r,z,n ; LSTM i,f,g,o ; RNN tanh.
#include "ds2_common.h"

namespace {

enum { CELL_GRU = 0, CELL_LSTM = 1, CELL_RNN = 2 };
template <int CELL>
struct CellInfo;
template <>
struct CellInfo<CELL_GRU> {
  static constexpr int G = 3, NS = 4;  // saved planes: r, z, n, hn
};
template <>
struct CellInfo<CELL_LSTM> {
  static constexpr int G = 4, NS = 5;  // saved planes: i, f, g, o, c
};
template <>
struct CellInfo<CELL_RNN> {
  static constexpr int G = 1, NS = 0;
};

struct StepArgs {
  int H, N, D, Tp;
  int t0, t1;            // time index handled by direction 0 / 1 in this launch
  int tp0, tp1;          // backward only: time index processed by the previous launch (source of the recurrent term)
  int first;             // backward only: 1 for the first launch of the sweep (no recurrent term yet)
  const int* lens;       // [N] device
  const void* W;         // fwd: W_hh [D][G*H][H]   bwd: W_hh^T [D][H][G*H]     (storage type T)
  const float* bhh;      // [D][G*H]
  const void* GI;        // fwd: input projection [Tp*N][D*G*H] (T)             bwd: dGI (written)
  void* dGI;
  void* dGH;             // GRU bwd: [D][Tp][N][3H] (T);  null for LSTM/RNN (dGH == dGI slice)
  const void* dOut;      // bwd: grad of the layer output [Tp][N][H] (T)
  void* Hseq;            // h_t of direction d at Hseq + d*hseq_dstride + (t*N+n)*H   (T)
  long hseq_dstride;
  void* S;               // saved planes [D][Tp][N][NS*H] (T)
  const void* hT_in;     // carried state as MFMA operand (T) [D][N][H]
  void* hT_out;
  const float* h32_in;   // carried state fp32 [D][N][H]   (bwd: elementwise part of dh)
  float* h32_out;
  const float* c32_in;   // LSTM cell state (bwd: dc)
  float* c32_out;
  const float* h0;       // bwd: initial state of the forward [D][N][H] or null (= zeros): h_{t-1} / c_{t-1} of a sample's FIRST step
  const float* c0;
  int fold;              // bwd: 1 = the launch behind the last step: only dh += dgates(last step) * W_hh (no stores): the state
                         // buffers then hold d loss / d h0 (and d c0)
};

// K-split skinny GEMM: acc[mt][nt] (16x16 tiles) += A[rows][k] * Bmat[cols][k] over this wave's chunks.
// A rows: a_base + row*lda (row < nrows else 0).  B rows: b_rows[nt] + col*ldb.
template <typename T, int MT, int NT>
__device__ __forceinline__ void skinny_gemm(ds2_f32x4 (&acc)[MT][NT], const T* a_base, long lda, int nrows,
                                            const T* const (&b_rows)[NT], long ldb, int K, int wave, int lane) {
  constexpr int V = Vec16<T>::N;
  constexpr int KC = Mma<T>::K16;  // elements of K per chunk (64 bytes per row)
  constexpr int U = 4;             // chunks whose loads are in flight together
  const int li = lane & 15, lq = lane >> 4;
  const int nch = (K + KC - 1) / KC;
  // All loads are unconditional (a predicated 16-byte load compiles to branch + s_waitcnt vmcnt(0), i.e. one L2 round
  // trip per load, serialised): rows >= nrows re-read the last valid row (their results are never used), the K tail is
  // clamped to k = 0 and zeroed with a bit mask on both operands.
  const T* a_ptr[MT];
  const T* b_ptr[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_ptr[m] = a_base + (long)min(m * 16 + li, nrows - 1) * lda + lq * V;
#pragma unroll
  for (int n = 0; n < NT; ++n) b_ptr[n] = b_rows[n] + (long)li * ldb + lq * V;
  for (int c0 = wave; c0 < nch; c0 += 4 * U) {
    uint4 a[U][MT], b[U][NT];
    uint32_t msk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
      const int k = c * KC + lq * V;
      const bool kok = (c < nch) && (k < K);
      const int kk = kok ? c * KC : 0;
      msk[u] = kok ? 0xffffffffu : 0u;
#pragma unroll
      for (int m = 0; m < MT; ++m) a[u][m] = *reinterpret_cast<const uint4*>(a_ptr[m] + kk);
#pragma unroll
      for (int n = 0; n < NT; ++n) b[u][n] = *reinterpret_cast<const uint4*>(b_ptr[n] + kk);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        a[u][m].x &= msk[u]; a[u][m].y &= msk[u]; a[u][m].z &= msk[u]; a[u][m].w &= msk[u];
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        b[u][n].x &= msk[u]; b[u][n].y &= msk[u]; b[u][n].z &= msk[u]; b[u][n].w &= msk[u];
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) Mma<T>::mma16(acc[m][n], a[u][m], b[u][n]);
    }
  }
}

template <int MT, int NT>
__device__ __forceinline__ void reduce_store(float* red, const ds2_f32x4 (&acc)[MT][NT], int wave, int lane) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[((wave * MT + m) * NT + n) * 256 + mma16_row(r, lane) * 16 + (lane & 15)] = acc[m][n][r];
}
template <int MT, int NT>
__device__ __forceinline__ float reduce_load(const float* red, int m, int n, int idx) {
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) s += red[((w * MT + m) * NT + n) * 256 + idx];
  return s;
}

// ------------------------------------------------------------------------------------------------------------
// forward step
// ------------------------------------------------------------------------------------------------------------
template <typename T, int CELL, int MT>
__global__ void __launch_bounds__(256) k_rnn_step_fwd(StepArgs a) {
  constexpr int G = CellInfo<CELL>::G, NS = CellInfo<CELL>::NS;
  __shared__ float red[4 * MT * G * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, N = a.N;
  const int d = blockIdx.y, j0 = blockIdx.x * 16, nb = blockIdx.z * (MT * 16);
  const int t = d == 0 ? a.t0 : a.t1;
  const long GH = (long)G * H;

  ds2_f32x4 acc[MT][G];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[m][g] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
  const T* Wd = (const T*)a.W + (long)d * GH * H;
  const T* brows[G];
#pragma unroll
  for (int g = 0; g < G; ++g) brows[g] = Wd + ((long)g * H + j0) * H;
  const T* hT = (const T*)a.hT_in + ((long)d * N + nb) * H;
  skinny_gemm<T, MT, G>(acc, hT, H, N - nb, brows, H, H, wave, lane);
  reduce_store<MT, G>(red, acc, wave, lane);
  __syncthreads();

  const long ldgi = (long)a.D * GH;
  for (int e = tid; e < MT * 256; e += 256) {
    const int nl = e >> 4, jj = e & 15;
    const int n = nb + nl;
    if (n >= N) continue;
    const int m = nl >> 4, idx = (nl & 15) * 16 + jj;
    const int j = j0 + jj;
    const bool act = t < a.lens[n];
    const long st_off = ((long)d * N + n) * H + j;
    const float hprev = a.h32_in[st_off];
    const long row = (long)t * N + n;
    const T* gi = (const T*)a.GI + row * ldgi + (long)d * GH + j;
    const float* bh = a.bhh + (long)d * GH + j;
    T* hs = (T*)a.Hseq + (long)d * a.hseq_dstride + ((long)t * N + n) * H + j;
    T* sv = NS ? (T*)a.S + (((long)d * a.Tp + t) * N + n) * (long)(NS ? NS : 1) * H + j : nullptr;
    float hnew = 0.f;
    if (CELL == CELL_GRU) {
      float r = 0.f, z = 0.f, nn = 0.f, hn = 0.f;
      if (act) {
        const float ghr = reduce_load<MT, G>(red, m, 0, idx) + bh[0];
        const float ghz = reduce_load<MT, G>(red, m, 1, idx) + bh[H];
        hn = reduce_load<MT, G>(red, m, 2, idx) + bh[2 * H];
        r = sigmoid_acc(ldf(gi) + ghr);
        z = sigmoid_acc(ldf(gi + H) + ghz);
        nn = tanhf_(ldf(gi + 2 * H) + r * hn);
        hnew = (1.f - z) * nn + z * hprev;
      }
      stf(sv, r);
      stf(sv + H, z);
      stf(sv + 2 * H, nn);
      stf(sv + 3 * H, hn);
    } else if (CELL == CELL_LSTM) {
      const float cprev = a.c32_in[st_off];
      float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cn = 0.f;
      if (act) {
        ig = sigmoid_acc(ldf(gi) + reduce_load<MT, G>(red, m, 0, idx) + bh[0]);
        fg = sigmoid_acc(ldf(gi + H) + reduce_load<MT, G>(red, m, 1, idx) + bh[H]);
        gg = tanhf_(ldf(gi + 2 * H) + reduce_load<MT, G>(red, m, 2, idx) + bh[2 * H]);
        og = sigmoid_acc(ldf(gi + 3 * H) + reduce_load<MT, G>(red, m, 3, idx) + bh[3 * H]);
        cn = fg * cprev + ig * gg;
        hnew = og * tanhf_(cn);
      }
      stf(sv, ig);
      stf(sv + H, fg);
      stf(sv + 2 * H, gg);
      stf(sv + 3 * H, og);
      stf(sv + 4 * H, cn);
      a.c32_out[st_off] = act ? cn : cprev;
    } else {
      if (act) hnew = tanhf_(ldf(gi) + reduce_load<MT, G>(red, m, 0, idx) + bh[0]);
    }
    stf(hs, act ? hnew : 0.f);
    const float carry = act ? hnew : hprev;
    a.h32_out[st_off] = carry;
    stf((T*)a.hT_out + st_off, carry);
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward (BPTT) step.  dh_t = dOut[t] + z_{t'}*dh_{t'} (elementwise carry, fp32 state) + dgh_{t'} * W_hh (MFMA),
// t' = the step processed by the previous launch.  Emits dGI[t] (= d loss / d input projection, consumed afterwards
// by the dgrad / wgrad GEMMs) and, for GRU, dGH[t] (differs from dGI in the n gate: dn*r).
// ------------------------------------------------------------------------------------------------------------
template <typename T, int CELL, int MT>
__global__ void __launch_bounds__(256) k_rnn_step_bwd(StepArgs a) {
  constexpr int G = CellInfo<CELL>::G, NS = CellInfo<CELL>::NS;
  __shared__ float red[4 * MT * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, N = a.N;
  const int d = blockIdx.y, j0 = blockIdx.x * 16, nb = blockIdx.z * (MT * 16);
  const int t = d == 0 ? a.t0 : a.t1;
  const int tp = d == 0 ? a.tp0 : a.tp1;
  const long GH = (long)G * H;
  const long ldgi = (long)a.D * GH;

  ds2_f32x4 acc[MT][1];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m][0] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
  if (!a.first) {
    const T* WT = (const T*)a.W + (long)d * H * GH;
    const T* brows[1] = {WT + (long)j0 * GH};
    const T* A;
    long lda;
    if (CELL == CELL_GRU) {
      A = (const T*)a.dGH + (((long)d * a.Tp + tp) * N + nb) * GH;
      lda = GH;
    } else {
      A = (const T*)a.dGI + ((long)tp * N + nb) * ldgi + (long)d * GH;
      lda = ldgi;
    }
    skinny_gemm<T, MT, 1>(acc, A, lda, N - nb, brows, GH, (int)GH, wave, lane);
  }
  reduce_store<MT, 1>(red, acc, wave, lane);
  __syncthreads();

  for (int e = tid; e < MT * 256; e += 256) {
    const int nl = e >> 4, jj = e & 15;
    const int n = nb + nl;
    if (n >= N) continue;
    const int m = nl >> 4, idx = (nl & 15) * 16 + jj;
    const int j = j0 + jj;
    const int len = a.lens[n];
    const bool act = t < len;
    const long st_off = ((long)d * N + n) * H + j;
    const float dh_in = a.h32_in[st_off] + reduce_load<MT, 1>(red, m, 0, idx);
    if (a.fold) {           // behind the sweep: the recurrent term of the last processed step completes d loss / d h0
      a.h32_out[st_off] = dh_in;
      if (CELL == CELL_LSTM) a.c32_out[st_off] = a.c32_in[st_off];
      continue;
    }
    const long row = (long)t * N + n;
    T* dgi = (T*)a.dGI + row * ldgi + (long)d * GH + j;
    // previous step IN FORWARD ORDER (source of h_{t-1}); exists iff it was an active step of this sample
    const int tprev = d == 0 ? t - 1 : t + 1;
    const bool has_prev = d == 0 ? (t > 0) : (t + 1 < len);
    const long seq_off = (((long)d * a.Tp + t) * N + n);
    const long seq_prev = (((long)d * a.Tp + tprev) * N + n);
    if (CELL == CELL_GRU) {
      T* dgh = (T*)a.dGH + seq_off * GH + j;
      float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dh_out = dh_in;
      if (act) {
        const T* sv = (const T*)a.S + seq_off * (long)(NS ? NS : 1) * H + j;
        const float r = ldf(sv), z = ldf(sv + H), nn = ldf(sv + 2 * H), hn = ldf(sv + 3 * H);
        const float hprev = has_prev ? ldf((const T*)a.Hseq + (long)d * a.hseq_dstride + ((long)tprev * N + n) * H + j)
                                     : (a.h0 ? a.h0[st_off] : 0.f);      // the sample's first step: its initial state
        const float dh = ldf((const T*)a.dOut + row * H + j) + dh_in;
        dn = dh * (1.f - z) * (1.f - nn * nn);
        dz = dh * (hprev - nn) * z * (1.f - z);
        dr = dn * hn * r * (1.f - r);
        dnr = dn * r;
        dh_out = dh * z;
      }
      stf(dgi, dr);
      stf(dgi + H, dz);
      stf(dgi + 2 * H, dn);
      stf(dgh, dr);
      stf(dgh + H, dz);
      stf(dgh + 2 * H, dnr);
      a.h32_out[st_off] = dh_out;
    } else if (CELL == CELL_LSTM) {
      const float dc_in = a.c32_in[st_off];
      float di = 0.f, df = 0.f, dg = 0.f, do_ = 0.f, dh_out = dh_in, dc_out = dc_in;
      if (act) {
        const T* sv = (const T*)a.S + seq_off * (long)(NS ? NS : 1) * H + j;
        const float ig = ldf(sv), fg = ldf(sv + H), gg = ldf(sv + 2 * H), og = ldf(sv + 3 * H), cn = ldf(sv + 4 * H);
        const float cprev = has_prev ? ldf((const T*)a.S + seq_prev * (long)(NS ? NS : 1) * H + 4 * H + j) : (a.c0 ? a.c0[st_off] : 0.f);
        const float tc = tanhf_(cn);
        const float dh = ldf((const T*)a.dOut + row * H + j) + dh_in;
        const float dcn = dc_in + dh * og * (1.f - tc * tc);
        di = dcn * gg * ig * (1.f - ig);
        df = dcn * cprev * fg * (1.f - fg);
        dg = dcn * ig * (1.f - gg * gg);
        do_ = dh * tc * og * (1.f - og);
        dh_out = 0.f;
        dc_out = dcn * fg;
      }
      stf(dgi, di);
      stf(dgi + H, df);
      stf(dgi + 2 * H, dg);
      stf(dgi + 3 * H, do_);
      a.h32_out[st_off] = dh_out;
      a.c32_out[st_off] = dc_out;
    } else {
      float dg = 0.f, dh_out = dh_in;
      if (act) {
        const float hv = ldf((const T*)a.Hseq + (long)d * a.hseq_dstride + ((long)t * N + n) * H + j);
        const float dh = ldf((const T*)a.dOut + row * H + j) + dh_in;
        dg = dh * (1.f - hv * hv);
        dh_out = 0.f;
      }
      stf(dgi, dg);
      a.h32_out[st_off] = dh_out;
    }
  }
}

// state init: h32 = h0 (or 0), hT = cast(h0), c32 = c0 (or 0)
template <typename T>
__global__ void k_rnn_init(const float* __restrict__ h0, const float* __restrict__ c0, float* __restrict__ h32,
                           T* __restrict__ hT, float* __restrict__ c32, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float h = h0 ? h0[i] : 0.f;
    h32[i] = h;
    if (hT) stf(hT + i, h);
    if (c32) c32[i] = c0 ? c0[i] : 0.f;
  }
}

template <typename T, int CELL, int MT>
void launch_step(bool bwd, const StepArgs& a, dim3 grid, hipStream_t st) {
  if (bwd)
    hipLaunchKernelGGL((k_rnn_step_bwd<T, CELL, MT>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((k_rnn_step_fwd<T, CELL, MT>), grid, dim3(256), 0, st, a);
}

template <typename T, int CELL>
int sweep(bool bwd, StepArgs a, float* st32[2], void* stT[2], float* c32[2], hipStream_t st) {
  const int N = a.N, Tp = a.Tp;
  const int MT = N <= 16 ? 1 : 2;   // batch groups of 32 rows (blockIdx.z); each group re-streams its W slice from L2
  dim3 grid(a.H / 16, a.D, ds2_cdiv(N, MT * 16));
  const int want_fold = a.fold;
  a.fold = 0;
  for (int s = 0; s < Tp; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    if (!bwd) {
      a.t0 = s;
      a.t1 = Tp - 1 - s;
    } else {
      a.t0 = Tp - 1 - s;
      a.t1 = s;
      a.tp0 = a.t0 + 1;
      a.tp1 = a.t1 - 1;
      a.first = s == 0;
    }
    a.h32_in = st32[cur];
    a.h32_out = st32[nxt];
    a.hT_in = stT[cur];
    a.hT_out = stT[nxt];
    a.c32_in = c32[cur];
    a.c32_out = c32[nxt];
    if (MT == 1)
      launch_step<T, CELL, 1>(bwd, a, grid, st);
    else
      launch_step<T, CELL, 2>(bwd, a, grid, st);
  }
  a.fold = want_fold;
  if (bwd && a.fold) {     // one launch behind the last step, see StepArgs::fold
    const int cur = Tp & 1, nxt = cur ^ 1;
    a.t0 = 0; a.t1 = Tp - 1; a.tp0 = 0; a.tp1 = Tp - 1; a.first = 0;
    a.h32_in = st32[cur]; a.h32_out = st32[nxt]; a.hT_in = stT[cur]; a.hT_out = stT[nxt]; a.c32_in = c32[cur]; a.c32_out = c32[nxt];
    if (MT == 1)
      launch_step<T, CELL, 1>(bwd, a, grid, st);
    else
      launch_step<T, CELL, 2>(bwd, a, grid, st);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <typename T>
int sweep_cell(int cell, bool bwd, const StepArgs& a, float* st32[2], void* stT[2], float* c32[2], hipStream_t st) {
  switch (cell) {
    case CELL_GRU: return sweep<T, CELL_GRU>(bwd, a, st32, stT, c32, st);
    case CELL_LSTM: return sweep<T, CELL_LSTM>(bwd, a, st32, stT, c32, st);
    case CELL_RNN: return sweep<T, CELL_RNN>(bwd, a, st32, stT, c32, st);
  }
  return DS2_ERR_ARG;
}

}  // namespace

extern "C" {

int ds2_rnn_gates(int cell) { return cell == CELL_GRU ? 3 : cell == CELL_LSTM ? 4 : cell == CELL_RNN ? 1 : -1; }
int ds2_rnn_saved_planes(int cell) { return cell == CELL_GRU ? 4 : cell == CELL_LSTM ? 5 : 0; }
// bytes of scratch state the sweeps need (fp32 + storage-type ping-pong state, fp32 cell ping-pong)
long ds2_rnn_state_bytes(int D, int N, int H) { return (long)D * N * H * (2 * 4 + 2 * 4 + 2 * 4); }

// Forward sweep over all Tp steps.
//   GI   [Tp*N][D*G*H] (T)  input projection incl. b_ih      Whh [D][G*H][H] (T)     bhh [D][G*H] f32
//   h0/c0 [D][N][H] f32 or null                               S [D][Tp][N][NS*H] (T)
//   Hseq: h_t of direction d is stored at Hseq + d*hseq_dstride + (t*N+n)*H (T); the binding allocates
//         [D][Tp+2][N][H] with zeroed guard slots and passes slot 1, so that h_{t-1} / h_{t+1} are plain views
//   hn/cn [D][N][H] f32 outputs (state after each sample's last valid step)
//   state: scratch of ds2_rnn_state_bytes()
int ds2_rnn_fwd(int dtype, int cell, int D, int N, int H, int Tp, const int* lens, const void* GI, const void* Whh,
                const float* bhh, const float* h0, const float* c0, void* Hseq, long hseq_dstride, void* S, float* hn, float* cn,
                void* state, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  DS2_REQUIRE(ds2_rnn_gates(cell) > 0, DS2_ERR_ARG);
  DS2_REQUIRE(H % 16 == 0 && (D == 1 || D == 2) && N > 0 && Tp > 0, DS2_ERR_ARG);
  const long n = (long)D * N * H;
  float* f = (float*)state;
  float* st32[2] = {f, f + n};
  float* c32[2] = {f + 2 * n, f + 3 * n};
  void* stT[2] = {(void*)(f + 4 * n), (void*)(f + 5 * n)};
  if (dtype == DS2_F32)
    hipLaunchKernelGGL(k_rnn_init<float>, dim3(ds2_cdiv(n, 256)), dim3(256), 0, st, h0, c0, st32[0], (float*)stT[0], c32[0], n);
  else
    hipLaunchKernelGGL(k_rnn_init<bf16_t>, dim3(ds2_cdiv(n, 256)), dim3(256), 0, st, h0, c0, st32[0], (bf16_t*)stT[0], c32[0], n);
  DS2_CHECK_LAUNCH();
  StepArgs a{};
  a.H = H; a.N = N; a.D = D; a.Tp = Tp; a.lens = lens; a.W = Whh; a.bhh = bhh; a.GI = GI; a.Hseq = Hseq; a.hseq_dstride = hseq_dstride; a.S = S;
  int rc = dtype == DS2_F32 ? sweep_cell<float>(cell, false, a, st32, stT, c32, st)
                            : sweep_cell<bf16_t>(cell, false, a, st32, stT, c32, st);
  if (rc) return rc;
  const int fin = Tp & 1;
  if (hn) hipMemcpyAsync(hn, st32[fin], n * 4, hipMemcpyDeviceToDevice, st);
  if (cn && cell == CELL_LSTM) hipMemcpyAsync(cn, c32[fin], n * 4, hipMemcpyDeviceToDevice, st);
  DS2_CHECK_LAUNCH();
  return 0;
}

// BPTT sweep.  dOut [Tp][N][H] (T) grad of the (direction-summed) layer output;  WhhT [D][H][G*H] (T);
// writes dGI [Tp*N][D*G*H] (T) and, GRU only, dGH [D][Tp][N][3H] (T).
// h0 / c0 [D][N][H] f32 (may be null = zeros): the initial state the forward was given (reference model.py:224-230: `hs`); a
// sample's first step then uses them as h_{t-1} / c_{t-1}.  dh0 / dc0 (may be null): d loss / d h0, d c0 -- one extra launch behind
// the sweep adds the recurrent term of the last processed step.
int ds2_rnn_bwd(int dtype, int cell, int D, int N, int H, int Tp, const int* lens, const void* dOut, const void* WhhT,
                const void* Hseq, long hseq_dstride, const void* S, void* dGI, void* dGH, const float* h0, const float* c0,
                float* dh0, float* dc0, void* state, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  DS2_REQUIRE(ds2_rnn_gates(cell) > 0, DS2_ERR_ARG);
  DS2_REQUIRE(H % 16 == 0 && (D == 1 || D == 2) && N > 0 && Tp > 0, DS2_ERR_ARG);
  DS2_REQUIRE(cell != CELL_GRU || dGH != nullptr, DS2_ERR_ARG);
  const long n = (long)D * N * H;
  float* f = (float*)state;
  float* st32[2] = {f, f + n};
  float* c32[2] = {f + 2 * n, f + 3 * n};
  void* stT[2] = {nullptr, nullptr};
  hipLaunchKernelGGL(k_rnn_init<float>, dim3(ds2_cdiv(n, 256)), dim3(256), 0, st, (const float*)nullptr,
                     (const float*)nullptr, st32[0], (float*)nullptr, c32[0], n);
  DS2_CHECK_LAUNCH();
  StepArgs a{};
  a.H = H; a.N = N; a.D = D; a.Tp = Tp; a.lens = lens; a.W = WhhT; a.dOut = dOut; a.Hseq = (void*)Hseq; a.hseq_dstride = hseq_dstride; a.S = (void*)S;
  a.dGI = dGI; a.dGH = dGH; a.h0 = h0; a.c0 = c0; a.fold = (dh0 || dc0) ? 1 : 0;
  const int rc = dtype == DS2_F32 ? sweep_cell<float>(cell, true, a, st32, stT, c32, st)
                                  : sweep_cell<bf16_t>(cell, true, a, st32, stT, c32, st);
  if (rc != 0) return rc;
  const int fin = (Tp + (a.fold ? 1 : 0)) & 1;      // the buffer the last launch wrote
  if (dh0) (void)hipMemcpyAsync(dh0, st32[fin], n * 4, hipMemcpyDeviceToDevice, st);
  if (dc0 && cell == CELL_LSTM) (void)hipMemcpyAsync(dc0, c32[fin], n * 4, hipMemcpyDeviceToDevice, st);
  return 0;
}

}  // extern "C"
