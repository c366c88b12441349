// Conv front-end of MaskConv (reference model.py:53-69 over the Sequential at model.py:157-164).
//
//   conv1  Conv2d(1,32,(41,11),s=(2,2),p=(20,5))   Cin = 1: no contraction over channels.
//          fp32 storage: VALU kernels with the 32 output channels as per-thread accumulators and the weights (fwd) / the output
//          gradient (wgrad) as wave-uniform scalar operands (s_load + v_fmac with an SGPR source), input patch in LDS.
//          bf16 storage: MFMA kernels whose contraction is the kernel's TIME tap (k_conv1_fwd_mfma / k_conv1_wgrad_mfma below).
//   conv2  Conv2d(32,32,(21,11),s=(2,1),p=(10,5))   implicit GEMM on MFMA 32x32: for every kernel tap a
//          [32 out] x [32 in] x [32 positions] product; activations are NFTC (channel fastest) so the K operand is a
//          contiguous 64-byte channel vector of one position.
//          fp32 storage: k_conv_tap (input patch of a 4-row x 32-position output tile and the 11 taps of one kernel row staged in
//          LDS, 80-byte position stride = conflict-free ds_read_b128); dgrad is the SAME kernel run once per output-row parity
//          with the flipped, parity-subsampled weights (a stride-2 transposed conv is two stride-1 convs); wgrad contracts over
//          positions with the exact-fp32 MFMA 32x32x2.
//          bf16 storage: k_conv_rtap (forward and dgrad: taps resident in registers) and k_conv2_wgrad_bf16d (several kernel
//          rows per workgroup, taps split over the waves).
// Layouts: x (N,1,161,T) f32 as given by the loader; activations NFTC [N][F][T'][32] in storage type T.
#include "ds2_common.h"

namespace {

constexpr int F0 = 161, F1 = 81, F2 = 41, CH = 32;
constexpr int K1F = 41, K1T = 11, K2F = 21, K2T = 11;

// ============================================================================================================
// conv1 forward
// ============================================================================================================
constexpr int C1_FB = 4, C1_TB = 64;                       // output tile: 4 freq rows x 64 frames
constexpr int C1_PR = (C1_FB - 1) * 2 + K1F;               // 47 input rows
constexpr int C1_PC = (C1_TB - 1) * 2 + K1T;               // 137 input cols
constexpr int C1_PCP = C1_PC + 1;

__device__ __forceinline__ void conv1_load_patch(float* patch, const float* __restrict__ xn, int T, int f_base, int t_base,
                                                 int tid, int nthreads, int f0) {
  for (int i = tid; i < C1_PR * C1_PC; i += nthreads) {
    const int pr = i / C1_PC, pc = i - pr * C1_PC;
    const int fi = f_base + pr, ti = t_base + pc;
    // unconditional load from a clamped address, zeroed by a select afterwards: keeps all of a thread's ~25 loads in flight
    const bool ok = fi >= 0 && fi < f0 && ti >= 0 && ti < T;
    const float v = xn[(long)min(max(fi, 0), f0 - 1) * T + min(max(ti, 0), T - 1)];
    patch[pr * C1_PCP + pc] = ok ? v : 0.f;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_conv1_fwd(const float* __restrict__ x, const float* __restrict__ w1k,
                                                    const float* __restrict__ b1, const int* __restrict__ lens,
                                                    T* __restrict__ y1, int N, int Tin, int Tp, int f0, int f1) {
  // f0 / f1: input frequency bins / output rows (runtime: the reference derives them from SpectConfig, model.py:166-169)
  __shared__ float patch[C1_PR * C1_PCP];
  const int tid = threadIdx.x;
  const int to0 = blockIdx.x * C1_TB, fo0 = blockIdx.y * C1_FB, n = blockIdx.z;
  conv1_load_patch(patch, x + (long)n * f0 * Tin, Tin, 2 * fo0 - 20, 2 * to0 - 5, tid, 256, f0);
  __syncthreads();
  const int r = tid >> 6, c = tid & 63;
  const int fo = fo0 + r, to = to0 + c;
  float acc[CH];
#pragma unroll
  for (int co = 0; co < CH; ++co) acc[co] = b1[co];
  const float* prow = patch + (2 * r) * C1_PCP + 2 * c;
  for (int kf = 0; kf < K1F; ++kf) {
    const float* wk = w1k + kf * K1T * CH;
#pragma unroll
    for (int kt = 0; kt < K1T; ++kt) {
      const float xv = prow[kf * C1_PCP + kt];
#pragma unroll
      for (int co = 0; co < CH; ++co) acc[co] = fmaf(xv, wk[kt * CH + co], acc[co]);
    }
  }
  if (fo < f1 && to < Tp) {
    const bool live = to < lens[n];
    T* dst = y1 + (((long)n * f1 + fo) * Tp + to) * CH;
    constexpr int V = Vec16<T>::N;
#pragma unroll
    for (int v = 0; v < CH / V; ++v) {
      float o[V];
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = live ? acc[v * V + i] : 0.f;
      Vec16<T>::store(dst + v * V, o);
    }
  }
}

// ============================================================================================================
// conv1 weight gradient: thread <-> kernel tap, 32 output-channel accumulators, dy as the wave-uniform operand
// ============================================================================================================
constexpr int C1W_THREADS = 256;
constexpr int C1W_MAXBLOCKS = 1024;

template <typename T>
__device__ __forceinline__ void load_dy32(const T* __restrict__ p, float (&d)[CH]);
template <>
__device__ __forceinline__ void load_dy32<float>(const float* __restrict__ p, float (&d)[CH]) {
#pragma unroll
  for (int i = 0; i < CH; ++i) d[i] = p[i];
}
template <>
__device__ __forceinline__ void load_dy32<bf16_t>(const bf16_t* __restrict__ p, float (&d)[CH]) {
  const uint32_t* u = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
  for (int i = 0; i < CH / 2; ++i) {
    const uint32_t w = u[i];
    d[2 * i] = __uint_as_float(w << 16);
    d[2 * i + 1] = __uint_as_float(w & 0xffff0000u);
  }
}

template <typename T>
__global__ void __launch_bounds__(C1W_THREADS) k_conv1_wgrad(const float* __restrict__ x, const T* __restrict__ dy1,
                                                             float* __restrict__ partial, int N, int Tin, int Tp,
                                                             int nchunk_t, int nchunk_f, int f0, int f1) {
  // TWO kernel taps per thread (tid and tid + 256): the output-gradient channels arrive through scalar loads and are
  // unpacked bf16 -> fp32 on the scalar ALU, one unit per CU shared by all waves -- with one tap per thread that unpack
  // (32 SALU ops per position and wave) outran the 32 v_fmac it feeds; two taps per thread halve the SALU work per FMA.
  __shared__ float patch[C1_PR * C1_PCP];
  const int tid = threadIdx.x;
  constexpr int NTAP = K1F * K1T;
  const int tap0 = tid, tap1 = tid + C1W_THREADS < NTAP ? tid + C1W_THREADS : tid;
  const bool has1 = tid + C1W_THREADS < NTAP;
  const int kf0 = tap0 / K1T, kt0 = tap0 - kf0 * K1T;
  const int kf1 = tap1 / K1T, kt1 = tap1 - kf1 * K1T;
  float acc0[CH], acc1[CH];
#pragma unroll
  for (int co = 0; co < CH; ++co) acc0[co] = acc1[co] = 0.f;
  const int total = N * nchunk_f * nchunk_t;
  for (int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
    const int ct = chunk % nchunk_t, cf = (chunk / nchunk_t) % nchunk_f, n = chunk / (nchunk_t * nchunk_f);
    const int to0 = ct * C1_TB, fo0 = cf * C1_FB;
    __syncthreads();
    conv1_load_patch(patch, x + (long)n * f0 * Tin, Tin, 2 * fo0 - 20, 2 * to0 - 5, tid, C1W_THREADS, f0);
    __syncthreads();
    const int rmax = min(C1_FB, f1 - fo0), cmax = min(C1_TB, Tp - to0);
    for (int r = 0; r < rmax; ++r) {
      const T* dyrow = dy1 + (((long)n * f1 + fo0 + r) * Tp + to0) * CH;
      const float* prow0 = patch + (2 * r + kf0) * C1_PCP + kt0;
      const float* prow1 = patch + (2 * r + kf1) * C1_PCP + kt1;
      for (int c = 0; c < cmax; ++c) {
        float d[CH];
        load_dy32<T>(dyrow + (long)c * CH, d);   // address depends only on loop counters: scalar loads
        const float xv0 = prow0[2 * c], xv1 = prow1[2 * c];
#pragma unroll
        for (int co = 0; co < CH; ++co) {
          acc0[co] = fmaf(xv0, d[co], acc0[co]);
          acc1[co] = fmaf(xv1, d[co], acc1[co]);
        }
      }
    }
  }
  float* dst0 = partial + ((long)blockIdx.x * NTAP + tap0) * CH;
#pragma unroll
  for (int v = 0; v < CH / 4; ++v)
    *reinterpret_cast<float4*>(dst0 + 4 * v) = make_float4(acc0[4 * v], acc0[4 * v + 1], acc0[4 * v + 2], acc0[4 * v + 3]);
  if (has1) {
    float* dst1 = partial + ((long)blockIdx.x * NTAP + tap1) * CH;
#pragma unroll
    for (int v = 0; v < CH / 4; ++v)
      *reinterpret_cast<float4*>(dst1 + 4 * v) = make_float4(acc1[4 * v], acc1[4 * v + 1], acc1[4 * v + 2], acc1[4 * v + 3]);
  }
}

// ============================================================================================================
// conv1 on the matrix pipes (bf16 storage): what torch.autocast does with this layer (bf16 input and weight, fp32 sums)
// ============================================================================================================
// One workgroup block = 9 output rows x 128 output frames of one sample (81 = 9 x 9 rows: no ragged row block); wave w owns
// the 32 frames [32w, 32w+32).  The input patch (57 rows x 272 columns, column u <-> input frame 2*t0 - 5 + u) is staged in
// LDS as bf16.  The contraction index of one MFMA is the kernel's time tap kt padded from 11 to 16 (zero weights), one MFMA
// per kernel row kf: the operand of input row rho = 2*fl + kf is the same for every (output row fl, kernel row kf) pair on
// that input row, so it is read from LDS ONCE and multiplied into up to 9 accumulators -- 6.5 MFMAs per 1 KB LDS read,
// which is what takes this layer off the LDS roof (one read per MFMA = LDS-bound at half rate).
constexpr int M1_FB = 9, M1_TB = 128;
constexpr int M1_PR = 2 * (M1_FB - 1) + K1F;   // 57 input rows
constexpr int M1_PC = 2 * M1_TB + 16;          // 272 input columns
constexpr int M1_RSB = M1_PC * 2;              // LDS bytes per patch row
constexpr int M1_PATCH_BYTES = M1_PR * M1_RSB; // 31008
static_assert(F1 == 9 * M1_FB, "conv1 MFMA kernels assume 81 output rows");

template <int BATCH>
__device__ __forceinline__ void conv1_stage_bf16(unsigned char* patch, const float* __restrict__ xn, int T, int f_base, int t_base,
                                                 int tid) {
  // column pairs -> one packed dword.  Branch-free straight-line batches of BATCH pairs per thread, so that all 2*BATCH loads
  // of a batch are in flight together (with a branch per element the compiler waits for every load before it issues the
  // next: one L2 round trip per element).  Every thread runs the same ITER iterations; indices past the end are clamped to the
  // last pair, which is then simply written more than once with the same value.
  constexpr int NP = M1_PC / 2, TOTAL = M1_PR * NP, ITER = (TOTAL + 255) / 256;
#pragma unroll
  for (int b0 = 0; b0 < ITER; b0 += BATCH) {
    float v0[BATCH], v1[BATCH];
#pragma unroll
    for (int it = 0; it < BATCH; ++it) {
      if (b0 + it < ITER) {
        const int i = min(tid + (b0 + it) * 256, TOTAL - 1);
        const int pr = i / NP, pc = 2 * (i - pr * NP);
        const int t0 = t_base + pc;
        const int ro = min(max(f_base + pr, 0), F0 - 1) * T;        // 32-bit offsets: one sample is 161 * T floats
        v0[it] = xn[ro + min(max(t0, 0), T - 1)];
        v1[it] = xn[ro + min(max(t0 + 1, 0), T - 1)];
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // loads above, selects and LDS stores below: the scheduler must not sink each load to its use
#pragma unroll
    for (int it = 0; it < BATCH; ++it) {
      if (b0 + it < ITER) {
        const int i = min(tid + (b0 + it) * 256, TOTAL - 1);
        const int pr = i / NP, pc = 2 * (i - pr * NP);
        const int fi = f_base + pr, t0 = t_base + pc, t1 = t0 + 1;
        const bool okf = fi >= 0 && fi < F0;
        const float a = okf && t0 >= 0 && t0 < T ? v0[it] : 0.f;
        const float b = okf && t1 >= 0 && t1 < T ? v1[it] : 0.f;
        *reinterpret_cast<uint32_t*>(patch + pr * M1_RSB + pc * 2) = cvt_pk_bf16(a, b);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ void __launch_bounds__(256, 1) k_conv1_fwd_mfma(const float* __restrict__ x, const float* __restrict__ w1k,
                                                           const float* __restrict__ b1, const int* __restrict__ lens,
                                                           bf16_t* __restrict__ y1, int N, int Tin, int Tp, int ntb) {
  __shared__ __attribute__((aligned(16))) unsigned char patch[M1_PATCH_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = lane & 31, h = lane >> 5;
  // the weights as A fragments, resident for the whole launch: lane (channel = lane & 31, k-block h) holds kt = 8h .. 8h+7 of row kf
  uint4 A[K1F];
#pragma unroll
  for (int kf = 0; kf < K1F; ++kf) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kt = 8 * h + i;
      const float wv = w1k[(kf * K1T + min(kt, K1T - 1)) * CH + pos];
      v[i] = kt < K1T ? wv : 0.f;
    }
    A[kf] = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
  }
  float bias[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias[r] = b1[mma32_row(r, lane)];
  const unsigned char* bp = patch + wave * 128 + pos * 4 + h * 16;
  const int total = N * ntb * 9;
  for (int blk = blockIdx.x; blk < total; blk += gridDim.x) {
    const int fb = blk % 9, tb = (blk / 9) % ntb, n = blk / (9 * ntb);
    __syncthreads();
    conv1_stage_bf16<8>(patch, x + (long)n * F0 * Tin, Tin, 18 * fb - 20, 256 * tb - 5, tid);
    __syncthreads();
    ds2_f32x16 acc[M1_FB];
#pragma unroll
    for (int fl = 0; fl < M1_FB; ++fl)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fl][r] = bias[r];
    auto frag = [&](int rho) {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(bp + rho * M1_RSB);
      return make_uint4(q[0], q[1], q[2], q[3]);
    };
    uint4 bn = frag(0);
#pragma unroll
    for (int rho = 0; rho < M1_PR; ++rho) {
      const uint4 b = bn;
      if (rho + 1 < M1_PR) bn = frag(rho + 1);      // the next row's operand is read under this row's MFMAs
#pragma unroll
      for (int fl = 0; fl < M1_FB; ++fl) {
        const int kf = rho - 2 * fl;
        if (kf >= 0 && kf < K1F) Mma<bf16_t>::mma32(acc[fl], A[kf], b);
      }
    }
    const int t = tb * M1_TB + wave * 32 + pos;
    if (t < Tp) {
      const bool live = t < lens[n];
#pragma unroll
      for (int fl = 0; fl < M1_FB; ++fl) {
        bf16_t* dst = y1 + (((long)n * F1 + 9 * fb + fl) * Tp + t) * CH + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 o = make_uint2(cvt_pk_bf16(acc[fl][4 * g], acc[fl][4 * g + 1]), cvt_pk_bf16(acc[fl][4 * g + 2], acc[fl][4 * g + 3]));
          if (!live) o = make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(dst + 8 * g) = o;
        }
      }
    }
  }
}

// Weight gradient: contraction over the frames (32 per MFMA 16x16x32), A = input taps [16 kt x 32 frames] gathered from the
// staged patch (stride-2 columns), B = the output gradient transposed through LDS [32 frames x 16 channels].  Wave w owns the
// 64 frames [64*(w>>1), +64) and the channel half w & 1, and keeps its 41 accumulator tiles dW[kf][kt][16 channels] in
// registers; as in the forward the tap operand of input row rho serves every (fl, kf) pair on that row.
constexpr int M1_DYS = 72;                                   // frames per channel row of the transposed tile (64 + pad: conflict-free b128)
constexpr int M1_DYT_WAVE = M1_FB * 16 * M1_DYS * 2;         // 20736 bytes per wave
constexpr int M1_PATCH_PAD = 31232;                          // patch rounded up to 256 bytes
constexpr int M1W_SMEM = M1_PATCH_PAD + 4 * M1_DYT_WAVE;     // 114176
static_assert(K1F * K1T * CH * 4 <= M1W_SMEM, "reduction buffer aliases the staging area");

__global__ void __launch_bounds__(256, 1) k_conv1_wgrad_mfma(const float* __restrict__ x, const bf16_t* __restrict__ dy1,
                                                             float* __restrict__ partial, int N, int Tin, int Tp, int ntb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem1[];
  unsigned char* patch = smem1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pair = wave >> 1, hh = wave & 1;
  uint16_t* dyT = reinterpret_cast<uint16_t*>(smem1 + M1_PATCH_PAD + wave * M1_DYT_WAVE);
  const int kt = lane & 15, q = lane >> 4;
  ds2_f32x4 acc[K1F];
#pragma unroll
  for (int kf = 0; kf < K1F; ++kf) acc[kf] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
  // byte offset of this lane's first tap inside a patch row: column u = 2 * (64*pair + 32*c + 8*q + i) + kt
  const unsigned char* xp = patch + (2 * (64 * pair + 8 * q) + kt) * 2;
  const int total = N * ntb * 9;
  for (int blk = blockIdx.x; blk < total; blk += gridDim.x) {
    const int fb = blk % 9, tb = (blk / 9) % ntb, n = blk / (9 * ntb);
    __syncthreads();
    conv1_stage_bf16<16>(patch, x + (long)n * F0 * Tin, Tin, 18 * fb - 20, 256 * tb - 5, tid);
    {  // this wave's output-gradient tile, transposed: dyT[fl][channel of the half][frame]
      const int t0 = tb * M1_TB + 64 * pair, piece = lane & 1;
      uint4 dv[M1_FB][2];
#pragma unroll
      for (int fl = 0; fl < M1_FB; ++fl)
#pragma unroll
        for (int j = 0; j < 2; ++j) {      // unconditional loads from a clamped frame, zeroed by a select: all 18 in flight together
          const int t = t0 + 32 * j + (lane >> 1);
          const uint4 v = *reinterpret_cast<const uint4*>(dy1 + (((long)n * F1 + 9 * fb + fl) * Tp + min(t, Tp - 1)) * CH + 16 * hh + 8 * piece);
          dv[fl][j] = t < Tp ? v : make_uint4(0u, 0u, 0u, 0u);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int fl = 0; fl < M1_FB; ++fl)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int tl = 32 * j + (lane >> 1);
          const uint4 v = dv[fl][j];
          uint16_t* d = dyT + (fl * 16 + 8 * piece) * M1_DYS + tl;
          d[0 * M1_DYS] = (uint16_t)v.x;
          d[1 * M1_DYS] = (uint16_t)(v.x >> 16);
          d[2 * M1_DYS] = (uint16_t)v.y;
          d[3 * M1_DYS] = (uint16_t)(v.y >> 16);
          d[4 * M1_DYS] = (uint16_t)v.z;
          d[5 * M1_DYS] = (uint16_t)(v.z >> 16);
          d[6 * M1_DYS] = (uint16_t)v.w;
          d[7 * M1_DYS] = (uint16_t)(v.w >> 16);
        }
    }
    __syncthreads();
    uint4 dyb[M1_FB][2];
#pragma unroll
    for (int fl = 0; fl < M1_FB; ++fl)
#pragma unroll
      for (int c = 0; c < 2; ++c) dyb[fl][c] = *reinterpret_cast<const uint4*>(dyT + (fl * 16 + kt) * M1_DYS + 32 * c + 8 * q);
    // the 8 strided taps of one operand as raw 16-bit reads; they are issued one operand AHEAD of the MFMAs that consume them
    // (read right before use they cost four exposed LDS round trips per operand: 5x the MFMA time of the block)
    uint16_t xn[8];
    auto taps = [&](int it) {      // it = 2 * rho + c
      const uint16_t* xr = reinterpret_cast<const uint16_t*>(xp + (it >> 1) * M1_RSB + (it & 1) * 128);
#pragma unroll
      for (int i = 0; i < 8; ++i) xn[i] = xr[2 * i];
    };
    taps(0);
#pragma unroll
    for (int it = 0; it < 2 * M1_PR; ++it) {
      const int rho = it >> 1, c = it & 1;
      uint4 xa;
      xa.x = (uint32_t)xn[0] | ((uint32_t)xn[1] << 16);
      xa.y = (uint32_t)xn[2] | ((uint32_t)xn[3] << 16);
      xa.z = (uint32_t)xn[4] | ((uint32_t)xn[5] << 16);
      xa.w = (uint32_t)xn[6] | ((uint32_t)xn[7] << 16);
      if (it + 1 < 2 * M1_PR) taps(it + 1);
#pragma unroll
      for (int fl = 0; fl < M1_FB; ++fl) {
        const int kf = rho - 2 * fl;
        if (kf >= 0 && kf < K1F) Mma<bf16_t>::mma16(acc[kf], xa, dyb[fl][c]);
      }
    }
  }
  // the two frame halves' sums in a fixed order, then one partial per workgroup: partial[blockIdx.x][451][32]
  float* red = reinterpret_cast<float*>(smem1);
  for (int w = 0; w < 2; ++w) {
    __syncthreads();
    if (pair == w) {
#pragma unroll
      for (int kf = 0; kf < K1F; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 4 * q + r;                 // 16x16 D layout: row = 4*(lane>>4) + r (tap), column = lane & 15 (channel)
          if (k < K1T) {
            float* e = red + (kf * K1T + k) * CH + 16 * hh + kt;
            *e = w == 0 ? acc[kf][r] : *e + acc[kf][r];
          }
        }
    }
  }
  __syncthreads();
  float* dst = partial + (long)blockIdx.x * (K1F * K1T * CH);
  for (int e = tid; e < K1F * K1T * CH; e += 256) dst[e] = red[e];
}

// ============================================================================================================
// conv2 forward / dgrad: MFMA tap-GEMM
// ============================================================================================================
constexpr int CT_UB = 4, CT_TB = 32;          // output tile: 4 rows (one per wave) x 32 positions
constexpr int CT_POSB = 80;                   // LDS bytes per position / per weight row: 64 data + 16 pad
constexpr int CT_MAXPR = (CT_UB - 1) * 2 + K2F;   // 27
constexpr int CT_PC = CT_TB + K2T - 1;            // 42
constexpr int CT_PATCH_BYTES = CT_MAXPR * CT_PC * CT_POSB;       // 90720
constexpr int CT_WST_BYTES = K2T * CH * CT_POSB;                  // 28160 per kernel row (one pass)
constexpr int CT_SMEM = CT_PATCH_BYTES + 2 * CT_WST_BYTES;        // 147040

struct ConvTapArgs {
  const void* X;      // [N][Fin][Tp][32]
  const void* W;      // [KF*KT][32 m][32 k]
  const float* bias;  // [32] or null
  const int* lens;    // [N] or null (time mask on the output)
  void* Y;            // [N][Fout][Tp][32]
  int N, Tp, Fin, Fout;
  int KF, SF, PF;     // kernel rows, input row stride, top padding  (kernel cols = 11, pad 5, col stride 1)
  int U, OSF, OQ;     // output rows handled: f_out = u*OSF + OQ, u in [0,U)
};

// NT = 32-position tiles per output row and workgroup.  NT = 2 (used by the stride-1 dgrad launches, whose input patch is
// only 14 rows tall) re-uses every weight fragment for two MFMAs and amortises the per-kernel-row weight staging over twice
// the work: the single-tile form is LDS-bound (2 fragment reads per MFMA + 22.5 KB of weights per kernel row).
template <typename T, int NT>
__global__ void __launch_bounds__(256, 1) k_conv_tap(ConvTapArgs a) {
  constexpr int V = Vec16<T>::N;
  constexpr int NPASS = (CH * (int)sizeof(T)) / 64;   // bf16: 1 pass of 32 channels; f32: 2 passes of 16 channels
  constexpr int CPB = 64 / (int)sizeof(T);            // channels per pass
  constexpr int TB = CT_TB * NT, PC = TB + K2T - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lq = lane >> 5;
  const int to0 = blockIdx.x * TB, u0 = blockIdx.y * CT_UB, n = blockIdx.z;
  const int PR = (CT_UB - 1) * a.SF + a.KF;
  unsigned char* patch = smem;
  unsigned char* wst = smem + PR * PC * CT_POSB;
  const int fi0 = u0 * a.SF - a.PF, ti0 = to0 - 5;
  const T* Xn = (const T*)a.X + (long)n * a.Fin * a.Tp * CH;
  const T* Wg = (const T*)a.W;

  ds2_f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int pass = 0; pass < NPASS; ++pass) {
    const int c0 = pass * CPB;
    __syncthreads();   // previous pass finished reading patch / wst
    // ---- stage the input patch: PR x PC positions x 64 bytes
    for (int i = tid; i < PR * PC * 4; i += 256) {
      const int v = i & 3, pos = i >> 2;
      const int pr = pos / PC, pc = pos - pr * PC;
      const int fi = fi0 + pr, ti = ti0 + pc;
      // unconditional load from a clamped address + bit mask (a predicated load = branch + s_waitcnt vmcnt(0) per load:
      // the ~18 loads of a thread would be 18 serial memory round trips)
      const uint32_t m = (fi >= 0 && fi < a.Fin && ti >= 0 && ti < a.Tp) ? 0xffffffffu : 0u;
      const int fc = min(max(fi, 0), a.Fin - 1), tc = min(max(ti, 0), a.Tp - 1);
      uint4 val = *reinterpret_cast<const uint4*>(Xn + ((long)fc * a.Tp + tc) * CH + c0 + v * V);
      val.x &= m; val.y &= m; val.z &= m; val.w &= m;
      *reinterpret_cast<uint4*>(patch + pos * CT_POSB + v * 16) = val;
    }
    // ---- weights of kernel row 0 of this pass
    auto wload = [&](int kf, int buf) {
      for (int i = tid; i < K2T * CH * 4; i += 256) {
        const int v = i & 3, row = i >> 2;   // row = kt*32 + m
        const uint4 val = *reinterpret_cast<const uint4*>(Wg + ((long)(kf * K2T) * CH + row) * CH + c0 + v * V);
        *reinterpret_cast<uint4*>(wst + buf * CT_WST_BYTES + row * CT_POSB + v * 16) = val;
      }
    };
    wload(0, 0);
    __syncthreads();
    for (int kf = 0; kf < a.KF; ++kf) {
      const int buf = kf & 1;
      if (kf + 1 < a.KF) wload(kf + 1, buf ^ 1);   // other buffer: last read two iterations ago (barrier in between)
      const unsigned char* wb = wst + buf * CT_WST_BYTES + li * CT_POSB + lq * 16;
      const unsigned char* pb = patch + ((wave * a.SF + kf) * PC + li) * CT_POSB + lq * 16;
#pragma unroll
      for (int kt = 0; kt < K2T; ++kt) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const uint4 af = *reinterpret_cast<const uint4*>(wb + kt * CH * CT_POSB + c * 32);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const uint4 bf = *reinterpret_cast<const uint4*>(pb + (kt + CT_TB * t) * CT_POSB + c * 32);
            Mma<T>::mma32(acc[t], af, bf);
          }
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: transpose the wave's [32 m][32 pos] tiles through LDS, then 64-byte channel-vector stores
  float* tile = reinterpret_cast<float*>(smem) + wave * (32 * 33);
  const int u = u0 + wave;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[li * 33 + mma32_row(r, lane)] = acc[t][r];   // tile[pos][m]
    __syncthreads();
    if (u < a.U) {
      const int fo = u * a.OSF + a.OQ;
      const int pos = lane >> 1, half = lane & 1;
      const int to = to0 + CT_TB * t + pos;
      if (to < a.Tp) {
        const bool live = a.lens ? (to < a.lens[n]) : true;
        T* dst = (T*)a.Y + (((long)n * a.Fout + fo) * a.Tp + to) * CH + half * 16;
#pragma unroll
        for (int v = 0; v < 16 / V; ++v) {
          float o[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const int m = half * 16 + v * V + i;
            float val = tile[pos * 33 + m] + (a.bias ? a.bias[m] : 0.f);
            o[i] = live ? val : 0.f;
          }
          Vec16<T>::store(dst + v * V, o);
        }
      }
    }
  }
}

// ============================================================================================================
// conv2 weight gradient: dW[kf][kt][co][ci] = sum_{n,fo,to} dY[n][fo][to][co] * A1[n][2fo-10+kf][to-5+kt][ci]
// grid (21 kernel rows, S position splits); exact-fp32 MFMA 32x32x2, K = position pairs
// ============================================================================================================
constexpr int CW_TB = 128;
constexpr int CW_SPLITS = 64;

template <typename T>
__global__ void __launch_bounds__(256, 1) k_conv2_wgrad(const T* __restrict__ dY, const T* __restrict__ A1,
                                                         float* __restrict__ partial, int N, int Tp, int f1, int f2) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(CW_TB + (CW_TB + K2T - 1)) * CH * sizeof(float)];
  T* sdy = reinterpret_cast<T*>(smem);                      // [CW_TB][32]
  T* sx = sdy + CW_TB * CH;                                 // [CW_TB + 10][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lq = lane >> 5;
  const int kf = blockIdx.x, split = blockIdx.y;
  constexpr int V = Vec16<T>::N;
  ds2_f32x16 acc[K2T];
#pragma unroll
  for (int k = 0; k < K2T; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  const int nrows = N * f2;
  for (int rr = split; rr < nrows; rr += gridDim.y) {
    const int n = rr / f2, fo = rr - n * f2;
    const int fi = 2 * fo - 10 + kf;
    if (fi < 0 || fi >= f1) continue;   // uniform per block
    const T* dyrow = dY + ((long)n * f2 + fo) * Tp * CH;
    const T* xrow = A1 + ((long)n * f1 + fi) * Tp * CH;
    for (int t0 = 0; t0 < Tp; t0 += CW_TB) {
      __syncthreads();
      for (int i = tid; i < CW_TB * (CH / V); i += 256) {
        const int p = i / (CH / V), v = i - p * (CH / V);
        const uint32_t mk = t0 + p < Tp ? 0xffffffffu : 0u;
        uint4 val = *reinterpret_cast<const uint4*>(dyrow + (long)min(t0 + p, Tp - 1) * CH + v * V);
        val.x &= mk; val.y &= mk; val.z &= mk; val.w &= mk;
        *reinterpret_cast<uint4*>(sdy + p * CH + v * V) = val;
      }
      for (int i = tid; i < (CW_TB + K2T - 1) * (CH / V); i += 256) {
        const int p = i / (CH / V), v = i - p * (CH / V);
        const int t = t0 - 5 + p;
        const uint32_t mk = (t >= 0 && t < Tp) ? 0xffffffffu : 0u;
        uint4 val = *reinterpret_cast<const uint4*>(xrow + (long)min(max(t, 0), Tp - 1) * CH + v * V);
        val.x &= mk; val.y &= mk; val.z &= mk; val.w &= mk;
        *reinterpret_cast<uint4*>(sx + p * CH + v * V) = val;
      }
      __syncthreads();
      const int npairs = (min(CW_TB, Tp - t0) + 1) / 2;   // positions past Tp hold dy = 0
      for (int p = wave; p < npairs; p += 4) {
        const float av = ldf(sdy + (2 * p + lq) * CH + li);
#pragma unroll
        for (int kt = 0; kt < K2T; ++kt) {
          const float bv = ldf(sx + (2 * p + lq + kt) * CH + li);
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[kt], 0, 0, 0);
        }
      }
    }
  }
  // cross-wave reduction, one tap at a time; red[wave][co][ci]
  float* red = reinterpret_cast<float*>(smem);
  float* out = partial + ((long)split * (K2F * K2T) + (long)kf * K2T) * (CH * CH);
  for (int kt = 0; kt < K2T; ++kt) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave * 1024 + mma32_row(r, lane) * 32 + li] = acc[kt][r];
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) out[(long)kt * 1024 + e] = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
  }
}

// ------------------------------------------------------------------------------------------------------------
// bf16 variant on the bf16 MFMA (16x the rate of the exact-fp32 form above; the operands ARE bf16 in this mode, so the products
// are exact and only the fp32 summation order differs): SEVERAL kernel rows per workgroup and the TAPS split over the waves.
// Wave w owns the taps kt = w, w + 4, w + 8 of R = 4 kernel rows kf0, kf0 + 2, ... of one parity -- they read the SAME input row
// fi = 2 j - 10 + kf0 against the output rows j, j - 1, ... -- and walks over all positions of a tile: 3 R accumulator tiles per wave
// (no cross-wave reduction at the end; wave 3's third tap does not exist: its products land in an accumulator that is never written
// out).  Six row groups cover the 21 kernel rows: {0,2,4,6} {8,..,14} {16,18,20,-} {1,..,7} {9,..,15} {17,19,-,-}.  Work item =
// (sample, j in [0, 40 + R - 1], position tile); an output row that does not exist is staged as zeros (branch-free).
// History (in the repository's history, not in this file): round 2 staged both operands through registers into channel-major LDS
// images (a transposing stage) and cut the tap windows out of them with v_alignbyte (`k_conv2_wgrad_bf16r`: 0.91 ms on config 3,
// 22 % of the MFMA rate); round 5 first pipelined that form (buffer loads without edge masks, window selection between the MFMAs:
// 0.71 ms) and then replaced it by the kernel below (0.49 ms).
// ------------------------------------------------------------------------------------------------------------
constexpr int CW_OOB = 0x7ffffff0;       // a buffer offset beyond any resource: the load returns zeros
constexpr int CWR_SPLITS = 85;           // position splits: 6 row groups x 85 = 510 workgroups on 512 slots (two per CU)

// ------------------------------------------------------------------------------------------------------------
// The row-group kernel WITHOUT a transposing stage (round 5).  Its predecessor moved every operand tile through
// registers to turn the position-major activations into channel-major LDS images (12 loads, 40 pack instructions and 44 LDS stores
// per thread and tile, un-overlapped: one wave per SIMD), and then cut the 11 shifted tap windows out of them on the VALU.  gfx950
// needs neither:
//   * `buffer_load_dwordx4 ... lds` copies a tile global -> LDS as it lies in memory ([position][32 channels], 64 B per position:
//     a wave-instruction = 16 positions = 1 KiB, contiguous on both sides); a position outside the clip or a row that does not exist
//     is an out-of-range offset and arrives as ZEROS (tools/probe_lds_dma_oob.py, profiles/r05zz_lds_dma_out_of_range.txt).  No
//     registers, no LDS stores: the next tile lands in the other buffer while this one is multiplied (one barrier per tile), and its
//     DMA instructions are issued one at a time BETWEEN the MFMAs (ten in one go behind the barrier, from all four waves at once,
//     wait at the full address queue while the matrix pipe idles: 0.56 -> 0.51 ms);
//   * `ds_read_b64_tr_b16` hands a lane k = 4 consecutive POSITIONS of one channel out of that layout (16 lanes pass the addresses of
//     a 4-position x 16-channel block): two reads = one MFMA operand.  A tap shift is a row offset of the read: (wave + 3) * 64 B in
//     the lane's base register, 4 i * 64 B as an immediate -- no window arithmetic at all.
// Per k-step and wave: 14 transposing reads (8 for the R dY^T fragments, 6 for the three tap windows), 12 MFMAs, no VALU; the reads
// of k-step ks + 1 are issued between the MFMAs of k-step ks (inline asm: the compiler would otherwise guard every LDS read against
// the DMA in flight with a vmcnt(0); every wait is explicit).
// KSN = k-steps (16 positions) per tile.  KSN = 7: two buffers are 72 KB and the kernel's 256 registers leave room for TWO workgroups
// per CU -- what one workgroup cannot cover (the barrier, the first fragments of a tile, the scalar bookkeeping) the other one's
// MFMAs do.  Work map: unit u = (position split, row group), units 64 x .. 64 x + 63 on XCD x (workgroup L runs on XCD L % 8), so
// that the six row groups of a split are neighbours on ONE XCD, resident together, and walk through the same tiles: a tile comes
// out of HBM / the Infinity Cache once per XCD and out of that XCD's L2 the other five times.
// ------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* cw_lds_ptr_t;
#ifndef CWD_ABL
#define CWD_ABL 0   // timing-only ablations: 1 no DMA after the first tile, 2 no LDS reads after k-step 0
#endif
#ifndef CWD_KSN
#define CWD_KSN 7
#endif
#define CWD_RDTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define CWD_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define CWD_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
constexpr int cwd_dy_bytes(int ksn) { return ksn * 16 * CH * 2; }          // one dY tile [16 KSN positions][32 channels]
constexpr int cwd_x_bytes(int ksn) { return (ksn + 1) * 16 * CH * 2; }     // X tile: 16 (KSN + 1) positions, position t0 - 8 first
constexpr int cwd_buf_bytes(int ksn) { return 4 * cwd_dy_bytes(ksn) + cwd_x_bytes(ksn); }
constexpr int cwd_smem_bytes(int ksn) { return 2 * cwd_buf_bytes(ksn); }
constexpr int cwd_wgs_per_cu(int ksn) { return 2 * cwd_smem_bytes(ksn) <= 160 * 1024 ? 2 : 1; }

struct CwdFrag {
  uint2 lo, hi;      // k = 0..3 | 4..7 of the lane's eight positions
};
__device__ __forceinline__ uint4 cwd_frag(const CwdFrag& f) { return make_uint4(f.lo.x, f.lo.y, f.hi.x, f.hi.y); }

// the 14 reads of k-step KS into fragment set (FA, FB), numbered 0..13: 2 r + h -> dY^T row r, half h; 8 + 2 i + h -> tap window i
#define CWD_READ(FA, FB, KS, IDX)                                                                             \
  do {                                                                                                        \
    constexpr int idx_ = (IDX);                                                                               \
    if constexpr ((CWD_ABL & 2) != 0 && (KS) > 0) {                                                           \
    } else if constexpr (idx_ < 8) {                                                                          \
      if constexpr (idx_ % 2 == 0) CWD_RDTR(FA[idx_ / 2].lo, aA, (idx_ / 2) * DYB + (KS) * 1024);             \
      else CWD_RDTR(FA[idx_ / 2].hi, aA, (idx_ / 2) * DYB + (KS) * 1024 + 256);                               \
    } else if constexpr (idx_ < 14) {                                                                         \
      if constexpr (idx_ % 2 == 0) CWD_RDTR(FB[(idx_ - 8) / 2].lo, aB, (KS) * 1024 + ((idx_ - 8) / 2) * 256); \
      else CWD_RDTR(FB[(idx_ - 8) / 2].hi, aB, (KS) * 1024 + ((idx_ - 8) / 2) * 256 + 256);                   \
    }                                                                                                         \
  } while (0)

// The NEXT tile's staging, cut into pieces that are issued between the MFMAs of this tile.  Branch-free: without a next tile (or for
// a row that does not exist) every lane's offset is out of range and the piece writes zeros into the idle buffer.
struct CwdNext {
  __amdgpu_buffer_rsrc_t rdy, rx;
  unsigned char* base;          // the other buffer
  int soff, sxoff;              // scalar offsets: this wave's dY row, the X row
  int t0, Tp;
  bool row_ok, any;             // dY row exists / there is a next tile
  int wave, lpos, lbyte;
};
template <int KSN, int C>
__device__ __forceinline__ void cwd_dma_dy(const CwdNext& nx) {       // 16 positions of this wave's dY^T tile
  const int pos = nx.t0 + 16 * C + nx.lpos;
  const int voff = (nx.row_ok && pos < nx.Tp) ? pos * (CH * 2) + nx.lbyte : CW_OOB;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(nx.rdy, (cw_lds_ptr_t)(nx.base + nx.wave * cwd_dy_bytes(KSN) + C * 1024), 16, voff, nx.soff, 0, 0);
}
template <int KSN, int Q>
__device__ __forceinline__ void cwd_dma_x(const CwdNext& nx) {        // chunk wave + 4 Q of the KSN + 1 X chunks
  const int c = nx.wave + 4 * Q;
  if (c < KSN + 1) {                                                  // wave-uniform
    const int pos = nx.t0 - 8 + 16 * c + nx.lpos;
    const int voff = (nx.any && (unsigned)pos < (unsigned)nx.Tp) ? pos * (CH * 2) + nx.lbyte : CW_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(nx.rx, (cw_lds_ptr_t)(nx.base + 4 * cwd_dy_bytes(KSN) + c * 1024), 16, voff, nx.sxoff, 0, 0);
  }
}

template <int KSN, int KS>
__device__ __forceinline__ void cwd_kstep(ds2_f32x16 (&acc)[4][3], CwdFrag (&fa)[2][4], CwdFrag (&fb)[2][3], uint32_t aA, uint32_t aB,
                                          const CwdNext& nx) {
  constexpr int c = KS & 1, n = c ^ 1, DYB = cwd_dy_bytes(KSN);
  CWD_WAIT_LGKM0();                            // the fragments of this k-step (issued during the previous one) have arrived
  __builtin_amdgcn_sched_barrier(0);
#define CWD_SLOT(M)                                                                          \
  Mma<bf16_t>::mma32(acc[(M) / 3][(M) % 3], cwd_frag(fa[c][(M) / 3]), cwd_frag(fb[c][(M) % 3])); \
  if constexpr (KS + 1 < KSN) {                                                              \
    CWD_READ(fa[n], fb[n], KS + 1, 2 * (M));                                                 \
    CWD_READ(fa[n], fb[n], KS + 1, 2 * (M) + 1);                                             \
  }                                                                                          \
  if constexpr (!(CWD_ABL & 1)) {                                                            \
    if constexpr ((M) == 8) cwd_dma_dy<KSN, KS>(nx);                                         \
    if constexpr ((M) == 10 && KS < 3) cwd_dma_x<KSN, KS>(nx);                               \
  }                                                                                          \
  __builtin_amdgcn_sched_barrier(0);
  CWD_SLOT(0) CWD_SLOT(1) CWD_SLOT(2) CWD_SLOT(3) CWD_SLOT(4) CWD_SLOT(5)
  CWD_SLOT(6) CWD_SLOT(7) CWD_SLOT(8) CWD_SLOT(9) CWD_SLOT(10) CWD_SLOT(11)
#undef CWD_SLOT
}

constexpr int CWD_UNITS_PER_XCD = 64;     // 85 splits x 6 row groups = 510 units on 8 x 64 workgroup slots
static_assert(CWR_SPLITS * 6 <= 8 * CWD_UNITS_PER_XCD, "every (split, row group) needs a workgroup slot");

template <int KSN>
__global__ void __launch_bounds__(256, cwd_wgs_per_cu(KSN)) k_conv2_wgrad_bf16d(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ A1,
                                                                                float* __restrict__ partial, int N, int Tp, int nsplit) {
  constexpr int R = 4;        // kernel rows per workgroup = waves: wave r stages dY^T tile r
  constexpr int TB = 16 * KSN, DYB = cwd_dy_bytes(KSN), BUF = cwd_buf_bytes(KSN);
  static_assert(KSN >= 3 && KSN <= 8, "X chunks: at most three per wave; DMA pieces: one dY chunk per k-step");
  const int unit = (blockIdx.x & 7) * CWD_UNITS_PER_XCD + (blockIdx.x >> 3);
  if (unit >= nsplit * 6) return;
  const int split = unit / 6, rg = unit - split * 6;
  const int kf0 = rg < 3 ? 8 * rg : 1 + 8 * (rg - 3);      // row groups {0,2,4,6} {8..14} {16,18,20,-} {1..7} {9..15} {17,19,-,-}
  extern __shared__ __attribute__((aligned(16))) unsigned char cwd_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  ds2_f32x16 acc[R][3];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][i][e] = 0.f;
  // work items (sample n, j in [0, 40 + R - 1], tile t) in the order wk = (n * NJ + j) * ntiles + t, every nsplit-th one is ours;
  // the cursor advances by additions and carries (a division per tile was a tenth of the tile's time in scalar instructions)
  const int ntiles = (Tp + TB - 1) / TB;
  constexpr int NJ = F2 + R - 1;
  struct Cursor {
    int n, j, t;
  };
  const int d_t = nsplit % ntiles, d_q = nsplit / ntiles, d_j = d_q % NJ, d_n = d_q / NJ;
  auto advance = [&](Cursor& c) {
    c.t += d_t;
    if (c.t >= ntiles) { c.t -= ntiles; c.j += 1; }
    c.j += d_j;
    if (c.j >= NJ) { c.j -= NJ; c.n += 1; }
    c.n += d_n;
  };
  auto next_live = [&](Cursor c) {              // skip the items whose input row does not exist; c.n >= N: no more work
    for (; c.n < N; advance(c)) {
      const int fi = 2 * c.j - 10 + kf0;
      if (fi >= 0 && fi < F1) break;
    }
    return c;
  };
  const int row_bytes = Tp * CH * 2;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, N * F2 * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)A1, 0, N * F1 * row_bytes, 0x00020000);
  const int lpos = lane >> 2, lbyte = (lane & 3) * 16;      // a wave-instruction: 16 positions x 64 B
  auto plan = [&](const Cursor& c, int b) {    // where a tile comes from and where it goes (scalar work only)
    CwdNext nx;
    nx.rdy = rdy;
    nx.rx = rx;
    nx.base = cwd_smem + b * BUF;
    nx.Tp = Tp;
    nx.wave = wave;
    nx.lpos = lpos;
    nx.lbyte = lbyte;
    nx.any = c.n < N;
    const int n = nx.any ? c.n : 0;
    const int fi = min(max(2 * c.j - 10 + kf0, 0), F1 - 1);
    const int fo = c.j - wave;
    nx.row_ok = nx.any && fo >= 0 && fo < F2 && kf0 + 2 * wave < K2F;
    nx.soff = __builtin_amdgcn_readfirstlane(nx.row_ok ? (n * F2 + fo) * row_bytes : 0);
    nx.sxoff = __builtin_amdgcn_readfirstlane((n * F1 + fi) * row_bytes);
    nx.t0 = c.t * TB;
    return nx;
  };
  // transposing reads: the 16-lane group gq = lane >> 4 covers channels 16 (gq & 1) .. + 15 and the positions 8 (gq >> 1) + {0..3 | 4..7}
  // of a k-step; lane i of the group passes the address of (position i / 4, channels 4 (i % 4) .. + 3) and receives its own channel
  const int l16 = lane & 15, gq = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)cwd_smem;
  const uint32_t la = lds0 + (8 * (gq >> 1) + (l16 >> 2)) * (CH * 2) + (16 * (gq & 1) + 4 * (l16 & 3)) * 2;
  const uint32_t lb = la + R * DYB + (wave + 3) * (CH * 2);      // tap kt = wave + 4 i reads X row p + kt + 3 (the tile starts at t0 - 8)
  CwdFrag fa[2][R], fb[2][3];

  Cursor cur;
  {
    const int q0 = split / ntiles;
    cur.t = split - q0 * ntiles;
    cur.n = q0 / NJ;
    cur.j = q0 - cur.n * NJ;
  }
  cur = next_live(cur);
  int b = 0;
  if (cur.n < N) {
    const CwdNext first = plan(cur, 0);
    cwd_dma_dy<KSN, 0>(first); cwd_dma_dy<KSN, 1>(first); cwd_dma_dy<KSN, 2>(first);
    if constexpr (KSN > 3) cwd_dma_dy<KSN, 3>(first);
    if constexpr (KSN > 4) cwd_dma_dy<KSN, 4>(first);
    if constexpr (KSN > 5) cwd_dma_dy<KSN, 5>(first);
    if constexpr (KSN > 6) cwd_dma_dy<KSN, 6>(first);
    if constexpr (KSN > 7) cwd_dma_dy<KSN, 7>(first);
    cwd_dma_x<KSN, 0>(first); cwd_dma_x<KSN, 1>(first); cwd_dma_x<KSN, 2>(first);
  }
  while (cur.n < N) {
    Cursor nc = cur;
    advance(nc);
    nc = next_live(nc);
    const CwdNext nx = plan(nc, b ^ 1);        // (scalar: before the wait, not behind it)
    CWD_WAIT_VM0();                            // this wave's share of the tile has landed ...
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();              // ... and everybody's; everybody is done with the other buffer
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t aA = la + b * BUF, aB = lb + b * BUF;
#define CWD_FIRST(IDX) CWD_READ(fa[0], fb[0], 0, IDX);
    CWD_FIRST(0) CWD_FIRST(1) CWD_FIRST(2) CWD_FIRST(3) CWD_FIRST(4) CWD_FIRST(5) CWD_FIRST(6)
    CWD_FIRST(7) CWD_FIRST(8) CWD_FIRST(9) CWD_FIRST(10) CWD_FIRST(11) CWD_FIRST(12) CWD_FIRST(13)
#undef CWD_FIRST
    cwd_kstep<KSN, 0>(acc, fa, fb, aA, aB, nx);
    cwd_kstep<KSN, 1>(acc, fa, fb, aA, aB, nx);
    cwd_kstep<KSN, 2>(acc, fa, fb, aA, aB, nx);
    if constexpr (KSN > 3) cwd_kstep<KSN, 3>(acc, fa, fb, aA, aB, nx);
    if constexpr (KSN > 4) cwd_kstep<KSN, 4>(acc, fa, fb, aA, aB, nx);
    if constexpr (KSN > 5) cwd_kstep<KSN, 5>(acc, fa, fb, aA, aB, nx);
    if constexpr (KSN > 6) cwd_kstep<KSN, 6>(acc, fa, fb, aA, aB, nx);
    if constexpr (KSN > 7) cwd_kstep<KSN, 7>(acc, fa, fb, aA, aB, nx);
    cur = nc;
    b ^= 1;
  }
  // every wave owns its taps: straight to partial[split][kf][kt][co][ci]
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int kt = wave + 4 * i;
      if (kt < K2T && kf0 + 2 * r < K2F) {
        float* out = partial + (((long)split * K2F + (kf0 + 2 * r)) * K2T + kt) * (CH * CH);
#pragma unroll
        for (int e = 0; e < 16; ++e) out[mma32_row(e, lane) * 32 + li] = acc[r][i][e];
      }
    }
}

// ============================================================================================================
// conv2 forward / dgrad for bf16 storage with the TAPS RESIDENT IN REGISTERS.
// k_conv_tap above is LDS-bound: every MFMA reads a fresh weight fragment and a fresh activation fragment (2 KB of LDS per
// 32-cycle MFMA, four waves per CU).  Here a launch is a stride-1 correlation with at most 11 x 11 taps (the forward is the sum
// of two such correlations over the even / odd input rows -- kf = 2m + q reads input row 2(u + m - 5) + q -- and the dgrad
// already is two of them), and its 121 x [32 x 32] weights are split over the four waves by contraction item (kernel column kt,
// channel half): <= 6 items x 11 kernel rows = 66 A fragments = 264 registers per wave, loaded ONCE per workgroup, which then
// walks over its tiles (4 output rows x 32 frames).  Per tile a wave reads each activation fragment of its items once per INPUT
// row and multiplies it into every output row that uses that input row (fl = rho - m): 84 LDS fragment reads for 264 MFMAs.
// The waves' partial sums over their items meet in LDS once per tile (fixed order), wave w finishing output row w; the next
// tile's input patch travels global -> LDS by DMA during the MFMAs (two patch buffers; round 5 -- until then it went through 40
// registers per lane and a commit phase behind the MFMAs).  Where a tile's time goes (config 3, forward = two launches, timing-only
// ablations RT_ABL): 0.51 ms as shipped, 0.46 without the DMA pieces, 0.40 without the partial-sum exchange and the output, 0.37
// without both; 0.28 would be the matrix pipe alone.
// ============================================================================================================
struct RTapArgs {
  const bf16_t* X;    // [N][Fin][Tp][32]
  const bf16_t* W;    // [rows*11][32 m][32 k]: kernel row m of this launch is row m*WSF + WQ of the tensor
  const float* bias;  // [32] or null
  const int* lens;    // [N] or null (time mask on the output)
  const float* Pin;   // fp32 partial sums [N][U][Tp][32] to add, or null
  float* Pout;        // write fp32 partial sums [N][U][Tp][32] instead of Y, or null
  bf16_t* Y;          // [N][Fout][Tp][32]
  int N, Tp, Fin, Fout;
  int KF, WSF, WQ;    // kernel rows (<= 11) and their place in W
  int ISF, IQ, PF;    // input row of (u, m) = (u + m - PF) * ISF + IQ
  int U, OSF, OQ;     // output rows handled: f_out = u*OSF + OQ, u in [0, U)
};
constexpr int RT_FB = 4, RT_TB = 32, RT_PC = RT_TB + K2T - 1;        // tile: 4 output rows x 32 frames; 42 patch columns
constexpr int RT_KFM = 11, RT_PR = RT_FB - 1 + RT_KFM;               // <= 11 kernel rows -> <= 14 patch rows
// (round 5) the patch travels global -> LDS by DMA (`buffer_load_dwordx4 ... lds`: no registers, no commit phase, out-of-range
// positions and rows arrive as zeros), one wave-instruction per 16 positions = 1 KiB, so a patch row is staged as 48 positions of
// 64 B (the last six are never loaded or read).  Without padding between positions a fragment read (32 positions x one 16-byte chunk)
// would hit every bank four times; the 16-byte chunks of position p are therefore stored in the order c ^ ((p >> 2) & 3) -- which the
// DMA does on its SOURCE side (lane l of a chunk fetches channel chunk (l & 3) ^ (l >> 4)): with ds_read_b128's lane groups
// {0-3, 12-15, 20-27}, ... the sixteen lanes of a group then touch sixteen different 16-byte slots of the 256-byte bank row.
constexpr int RT_PCD = 48, RT_ROWB = RT_PCD * CH * 2;                // staged positions per patch row; 3 072 bytes
constexpr int RT_PATCH = RT_PR * RT_ROWB;                            // 43 008 bytes
constexpr int RT_RED = 4 * RT_FB * 4096;                             // partial sums [wave][row][4][64 lanes] x 16 bytes
constexpr int RT_SMEM = 2 * RT_PATCH + RT_RED;                       // 151 552
#define RT_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

struct RtNext {                 // the tile whose patch is being staged (into the other buffer) while this one is multiplied
  __amdgpu_buffer_rsrc_t rx;
  unsigned char* patch;
  int n, ub, tb;                // tile coordinates; n < 0: no such tile (every lane out of range)
  int Fin, Tp, PF, ISF, IQ;
  int wave, lane;
};
// piece I of this wave: chunk I % 3 (16 positions) of patch row wave + 4 (I / 3) -- row offset and chunk are compile-time, what is left
// per piece is a handful of scalar instructions (a division of the chunk number by 3 and the row test per piece were 35: 385 SALU
// instructions per tile, issued where only one to four MFMAs are queued behind them)
template <int PR, int I>
__device__ __forceinline__ void rt_dma_piece(const RtNext& nx) {
  constexpr int k = I % 3;
  const int pr = nx.wave + 4 * (I / 3);
  if (pr < PR) {                                                     // wave-uniform
    const int fi = (nx.ub * RT_FB + pr - nx.PF) * nx.ISF + nx.IQ;
    const bool row_ok = nx.n >= 0 && fi >= 0 && fi < nx.Fin;
    const int soff = __builtin_amdgcn_readfirstlane(row_ok ? (nx.n * nx.Fin + fi) * (nx.Tp * CH * 2) : 0);
    const int pc = 16 * k + (nx.lane >> 2), ti = nx.tb * RT_TB - 5 + pc;
    const int voff = (row_ok && pc < RT_PC && (unsigned)ti < (unsigned)nx.Tp) ? ti * (CH * 2) + (((nx.lane & 3) ^ (nx.lane >> 4)) << 4) : CW_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(nx.rx, (cw_lds_ptr_t)(nx.patch + pr * RT_ROWB + k * 1024), 16, voff, soff, 0, 0);
  }
}
template <int PR, int I>
__device__ __forceinline__ void rt_dma_all(const RtNext& nx) {
  rt_dma_piece<PR, I>(nx);
  if constexpr (I + 1 < 12) rt_dma_all<PR, I + 1>(nx);
}
#ifndef RT_ABL
#define RT_ABL 0   // timing-only ablations: 1 no DMA pieces in the loop, 2 no partial-sum exchange and no output, 4 no fragment reads
#endif
constexpr int RT_AH = 4;          // steps a fragment is read ahead of its MFMAs (two: a step of one or two MFMAs is shorter than the LDS)
template <int ST>
__device__ __forceinline__ void rt_prefetch(uint4 (&bq)[RT_AH + 1], const uint32_t (&aj)[6]) {
  RT_RD128(bq[ST], aj[ST % 6], (ST / 6) * RT_ROWB);
  if constexpr (ST + 1 < RT_AH) rt_prefetch<ST + 1>(bq, aj);
}
// step ST = (input row rho, contraction item j) of a tile: the read of step ST + AH, the wait for this step's fragment, its MFMAs
// (one per output row that uses input row rho), now and then a DMA piece of the next tile.  Compile-time recursion: every LDS offset
// is an immediate and every branch below folds.
template <int KF, int ST>
__device__ __forceinline__ void rt_step(ds2_f32x16 (&acc)[RT_FB], const uint4 (&A)[KF][6], uint4 (&bq)[RT_AH + 1], const uint32_t (&aj)[6],
                                        const RtNext& nx) {
  constexpr int PR = RT_FB - 1 + KF, NS = PR * 6, rho = ST / 6, j = ST % 6;
  if constexpr (ST + RT_AH < NS) RT_RD128(bq[(ST + RT_AH) % (RT_AH + 1)], aj[(ST + RT_AH) % 6], ((ST + RT_AH) / 6) * RT_ROWB);
  constexpr int left = RT_AH < NS - 1 - ST ? RT_AH : NS - 1 - ST;      // reads that may still be in flight behind this step's fragment
  if constexpr (left >= 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
  else if constexpr (left == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
  else if constexpr (left == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
  else if constexpr (left == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int fl = 0; fl < RT_FB; ++fl) {
    const int m = rho - fl;
    if (m >= 0 && m < KF) Mma<bf16_t>::mma32(acc[fl], A[m][j], bq[ST % (RT_AH + 1)]);
  }
  if constexpr (ST % 6 == 3 && ST >= 9 && (ST - 9) / 6 < 12 && !(RT_ABL & 1)) rt_dma_piece<PR, (ST - 9) / 6>(nx);   // behind an MFMA group of the middle rows
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (ST + 1 < NS) rt_step<KF, ST + 1>(acc, A, bq, aj, nx);
}

template <int KF>   // kernel rows of the launch (11 or 10): compile-time, so that the MFMA loop is straight-line code
__global__ void __launch_bounds__(256, 1) k_conv_rtap(RTapArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
  float* red = reinterpret_cast<float*>(rsm + 2 * RT_PATCH);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lq = lane >> 5;
  // contraction items of this wave: item = 2*kt + channel half; waves 0,1 own 6, waves 2,3 own 5 and run a sixth with zero
  // weights (they would wait for waves 0,1 at the tile's barrier anyway; a wave-dependent trip count would put a branch around
  // every MFMA group)
  const int i0 = wave < 2 ? 6 * wave : 12 + 5 * (wave - 2), icnt = wave < 2 ? 6 : 5;
  uint4 A[KF][6];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)rsm;
  uint32_t boff[6];                              // LDS byte address of item j's activation fragment in patch row 0 of buffer 0
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int item = i0 + min(j, icnt - 1), kt = item >> 1, half = item & 1;
    const int p = li + kt;
    boff[j] = lds0 + p * (CH * 2) + (((half * 2 + lq) ^ ((p >> 2) & 3)) << 4);
#pragma unroll
    for (int m = 0; m < KF; ++m) {
      const uint4 wv = *reinterpret_cast<const uint4*>(a.W + (((long)(m * a.WSF + a.WQ) * K2T + kt) * CH + li) * CH + half * 16 + lq * 8);
      A[m][j] = j < icnt ? wv : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  // Round 6: 66 resident tap fragments do not fit the vector registers beside the accumulators; the allocator spilled ~30 of them into
  // accumulation registers and copied each back in front of its MFMA (127 v_accvgpr_read per tile row).  Pinned there with an "a"
  // constraint the MFMA takes them in place (srcA may be an accumulation register on gfx950) -- see pin_agpr in ds2_rnn_persist3_impl.h.
#ifndef DS2_RTAP_VKEEP
#define DS2_RTAP_VKEEP 24
#endif
#pragma unroll
  for (int m = 0; m < KF; ++m)
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (m * 6 + j >= DS2_RTAP_VKEEP) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u4v;
        u4v v = __builtin_bit_cast(u4v, A[m][j]);
        asm volatile("" : "+a"(v));
        A[m][j] = __builtin_bit_cast(uint4, v);
      }
  constexpr int PR = RT_FB - 1 + KF;
  constexpr int NPIECE = 12;                     // DMA pieces per wave and tile: up to four patch rows x three chunks
  const int nub = ds2_cdiv_dev(a.U, RT_FB), ntb = ds2_cdiv_dev(a.Tp, RT_TB);
  const int total = a.N * nub * ntb;
  RtNext nx;
  nx.rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, a.N * a.Fin * a.Tp * CH * 2, 0x00020000);
  nx.Fin = a.Fin; nx.Tp = a.Tp; nx.PF = a.PF; nx.ISF = a.ISF; nx.IQ = a.IQ;
  nx.wave = wave; nx.lane = lane;
  auto aim = [&](int tile, int buf) {            // scalar: which tile, which buffer
    const bool any = tile < total;
    const int tt = any ? tile : 0;
    nx.tb = tt % ntb;
    nx.ub = (tt / ntb) % nub;
    nx.n = any ? tt / (ntb * nub) : -1;
    nx.patch = rsm + buf * RT_PATCH;
  };

  int tile = blockIdx.x;
  if (tile < total) {
    aim(tile, 0);
    rt_dma_all<PR, 0>(nx);
  }
  __syncthreads();                               // (waits for the DMA: vmcnt(0) is part of it)
  int buf = 0;
  for (; tile < total; tile += gridDim.x, buf ^= 1) {
    const int tb = tile % ntb, ub = (tile / ntb) % nub, n = tile / (ntb * nub);
    const int next = tile + gridDim.x;
    aim(next, buf ^ 1);                          // its pieces are issued between the MFMAs below
    // the partial sums of the first pass (second launch of the forward) are needed behind the MFMAs: their loads go out now, not there
    float4 pin[4];
    {
      const int u = ub * RT_FB + wave, t = tb * RT_TB + li;
      const bool have = a.Pin != nullptr && u < a.U && t < a.Tp;
      const float* src = a.Pin + (((long)n * a.U + min(u, a.U - 1)) * a.Tp + min(t, a.Tp - 1)) * CH + 4 * lq;
#pragma unroll
      for (int g = 0; g < 4; ++g) pin[g] = have ? *reinterpret_cast<const float4*>(src + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    ds2_f32x16 acc[RT_FB];
#pragma unroll
    for (int fl = 0; fl < RT_FB; ++fl)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fl][r] = 0.f;
    // The activation fragments run TWO steps (input row rho, contraction item j) ahead of their MFMAs through three registers.  Inline
    // asm with explicit waits: the compiler would guard every LDS read it knows of against the DMA in flight with a vmcnt(0) -- and
    // left to itself it read every fragment into the same four registers and waited for it in front of its MFMAs (84 exposed LDS
    // latencies per tile, one wave per SIMD).  Same order of summation as before.
    {
      uint4 bq[RT_AH + 1];
      uint32_t aj[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) aj[j] = boff[j] + buf * RT_PATCH;
      __builtin_amdgcn_sched_barrier(0);
      rt_prefetch<0>(bq, aj);
      rt_step<KF, 0>(acc, A, bq, aj, nx);
    }
    // ---- the four waves' partial sums meet: red[wave][row][g][lane] x 16 bytes
    if (RT_ABL & 2) {
      if (acc[0][0] == 123.f && acc[1][1] == 3.f && acc[2][2] == 1.f && acc[3][3] == 7.f) red[0] = 1.f;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      continue;
    }
#pragma unroll
    for (int fl = 0; fl < RT_FB; ++fl)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(red + (((wave * RT_FB + fl) * 4 + g) * 64 + lane) * 4) =
            make_float4(acc[fl][4 * g], acc[fl][4 * g + 1], acc[fl][4 * g + 2], acc[fl][4 * g + 3]);
    // the next tile's patch (this wave's DMA pieces) must have landed before the barrier that publishes it: the compiler does not put
    // the vmcnt(0) there by itself.  Cheap: the pieces went out during the MFMA loop; the Pin loads are needed right behind anyway
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
      const int u = ub * RT_FB + wave, t = tb * RT_TB + li;       // wave w finishes output row w of the tile
      float o[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 s4 = *reinterpret_cast<const float4*>(red + (((0 * RT_FB + wave) * 4 + g) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float4 q4 = *reinterpret_cast<const float4*>(red + (((w * RT_FB + wave) * 4 + g) * 64 + lane) * 4);
          s4.x += q4.x; s4.y += q4.y; s4.z += q4.z; s4.w += q4.w;
        }
        o[4 * g] = s4.x; o[4 * g + 1] = s4.y; o[4 * g + 2] = s4.z; o[4 * g + 3] = s4.w;
      }
      if (u < a.U && t < a.Tp) {
        // lane (frame li, lq) holds channels 8g + 4lq .. +3 (the 32x32 D layout with the weights as the row operand)
        const long pidx = (((long)n * a.U + u) * a.Tp + t) * CH + 4 * lq;
        if (a.Pin) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            o[4 * g] += pin[g].x; o[4 * g + 1] += pin[g].y; o[4 * g + 2] += pin[g].z; o[4 * g + 3] += pin[g].w;
          }
        }
        if (a.Pout) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(a.Pout + pidx + 8 * g) = make_float4(o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
        } else {
          const bool live = a.lens ? (t < a.lens[n]) : true;
          bf16_t* dst = a.Y + (((long)n * a.Fout + u * a.OSF + a.OQ) * a.Tp + t) * CH + 4 * lq;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) e[c] = live ? o[4 * g + c] + (a.bias ? a.bias[8 * g + 4 * lq + c] : 0.f) : 0.f;
            *reinterpret_cast<uint2*>(dst + 8 * g) = make_uint2(cvt_pk_bf16(e[0], e[1]), cvt_pk_bf16(e[2], e[3]));
          }
        }
      }
    }
    // End of the tile: everybody is done with `red`.  NOT __syncthreads(): its vmcnt(0) would wait for the acknowledgement of the output
    // stores issued a moment ago (a memory round trip per tile with nothing to cover it); the next tile's patch has landed for every
    // wave since the barrier behind the partial sums (whose __syncthreads did wait for this wave's DMA).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int KF>
int conv_rtap_launch_kf(const RTapArgs& a, hipStream_t st) {
  static bool attr[DS2_MAX_DEVICES];
  if (ds2_first_use_on_device(attr)) (void)hipFuncSetAttribute((const void*)k_conv_rtap<KF>, hipFuncAttributeMaxDynamicSharedMemorySize, RT_SMEM);
  const long total = (long)a.N * ds2_cdiv(a.U, RT_FB) * ds2_cdiv(a.Tp, RT_TB);
  const int cus = ds2_cu_count();
  hipLaunchKernelGGL(k_conv_rtap<KF>, dim3((unsigned)(total < cus ? total : cus)), dim3(256), RT_SMEM, st, a);
  DS2_CHECK_LAUNCH();
  return 0;
}
int conv_rtap_launch(const RTapArgs& a, hipStream_t st) {
  if (a.KF == 11) return conv_rtap_launch_kf<11>(a, st);
  if (a.KF == 10) return conv_rtap_launch_kf<10>(a, st);
  return DS2_ERR_ARG;
}

template <typename T, int NT>
int conv_tap_launch_nt(const ConvTapArgs& a, hipStream_t st) {
  const int PR = (CT_UB - 1) * a.SF + a.KF, PC = CT_TB * NT + K2T - 1;
  const int shm = PR * PC * CT_POSB + 2 * CT_WST_BYTES;
  static int attr[DS2_MAX_DEVICES];              // per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = dev >= 0 && dev < DS2_MAX_DEVICES ? dev : 0;
  if (attr[dev] < shm) {
    (void)hipFuncSetAttribute((const void*)k_conv_tap<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    attr[dev] = shm;
  }
  dim3 grid(ds2_cdiv(a.Tp, CT_TB * NT), ds2_cdiv(a.U, CT_UB), a.N);
  hipLaunchKernelGGL((k_conv_tap<T, NT>), grid, dim3(256), shm, st, a);
  DS2_CHECK_LAUNCH();
  return 0;
}
template <typename T>
int conv_tap_launch(const ConvTapArgs& a, hipStream_t st) {
  // two position tiles per row whenever the taller patch still fits the 160 KiB of LDS (the stride-1 dgrad launches)
  const int PR = (CT_UB - 1) * a.SF + a.KF;
  if (PR * (2 * CT_TB + K2T - 1) * CT_POSB + 2 * CT_WST_BYTES <= 150 * 1024) return conv_tap_launch_nt<T, 2>(a, st);
  return conv_tap_launch_nt<T, 1>(a, st);
}

}  // namespace

// Frequency geometry (reference model.py:166-169): F0 = sample_rate * window_size / 2 + 1 input bins (161 at 16 kHz / 20 ms), F1 and
// F2 the rows after the two convolutions.  The matrix-pipe kernels of the bf16 path (k_conv1_*_mfma, k_conv_rtap, k_conv2_wgrad_bf16d)
// are specialised for 161 / 81 / 41; any other geometry runs the general kernels (k_conv1_fwd / _wgrad, k_conv_tap, k_conv2_wgrad)
// in the same storage type: same results, 2-3x the time of the layer.
static bool conv_geometry(int f0, int& f1, int& f2) {
  f1 = (f0 + 2 * 20 - K1F) / 2 + 1;
  f2 = (f1 + 2 * 10 - K2F) / 2 + 1;
  return f0 >= 1 && f1 >= 1 && f2 >= 1;
}

extern "C" {

int ds2_conv_rows(int F0, int* F1_out, int* F2_out) {
  int f1, f2;
  DS2_REQUIRE(conv_geometry(F0, f1, f2), DS2_ERR_ARG);
  if (F1_out) *F1_out = f1;
  if (F2_out) *F2_out = f2;
  return 0;
}

int ds2_conv1_fwd(int dtype, const float* x, const float* w1k, const float* b1, const int* lens, void* y1, int N, int F0_, int T,
                  int Tp, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  int f1, f2;
  DS2_REQUIRE(N > 0 && T > 0 && Tp == (T + 2 * 5 - 10 - 1) / 2 + 1 && conv_geometry(F0_, f1, f2), DS2_ERR_ARG);
  dim3 grid(ds2_cdiv(Tp, C1_TB), ds2_cdiv(f1, C1_FB), N);
  if (dtype == DS2_F32) {
    hipLaunchKernelGGL(k_conv1_fwd<float>, grid, dim3(256), 0, st, x, w1k, b1, lens, (float*)y1, N, T, Tp, F0_, f1);
  } else if (F0_ != F0) {
    hipLaunchKernelGGL(k_conv1_fwd<bf16_t>, grid, dim3(256), 0, st, x, w1k, b1, lens, (bf16_t*)y1, N, T, Tp, F0_, f1);
  } else {   // matrix pipes: persistent workgroups over the (sample, frame block, row block) tiles
    const int ntb = ds2_cdiv(Tp, M1_TB);
    const long total = (long)N * ntb * 9;
    const int cus = ds2_cu_count();
    hipLaunchKernelGGL(k_conv1_fwd_mfma, dim3((unsigned)(total < cus ? total : cus)), dim3(256), 0, st, x, w1k, b1, lens, (bf16_t*)y1, N,
                       T, Tp, ntb);
  }
  DS2_CHECK_LAUNCH();
  return 0;
}

static int conv1_wgrad_blocks(int N, int f1, int Tp) {
  long total = (long)N * ds2_cdiv(f1, C1_FB) * ds2_cdiv(Tp, C1_TB);
  return (int)(total < C1W_MAXBLOCKS ? total : C1W_MAXBLOCKS);
}
static int conv1_wgrad_mfma_blocks(int N, int Tp) {   // never more than conv1_wgrad_blocks (the workspace is sized for that)
  const long total = (long)N * ds2_cdiv(Tp, M1_TB) * 9;
  const int cus = ds2_cu_count();
  return (int)(total < cus ? total : cus);
}
long ds2_conv1_wgrad_ws_floats(int N, int F0_, int Tp) {
  int f1, f2;
  if (!conv_geometry(F0_, f1, f2)) return 0;
  const long P = conv1_wgrad_blocks(N, f1, Tp);
  return P * (K1F * K1T * CH) + (long)ds2_norm_partials(P) * (K1F * K1T * CH);
}
int ds2_conv1_wgrad(int dtype, const float* x, const void* dy1, float* dw1k, int N, int F0_, int T, int Tp, float* ws,
                    ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  int f1, f2;
  DS2_REQUIRE(conv_geometry(F0_, f1, f2), DS2_ERR_ARG);
  int P = conv1_wgrad_blocks(N, f1, Tp);
  const int ncf = ds2_cdiv(f1, C1_FB), nct = ds2_cdiv(Tp, C1_TB);
  if (dtype == DS2_F32) {
    hipLaunchKernelGGL(k_conv1_wgrad<float>, dim3(P), dim3(C1W_THREADS), 0, st, x, (const float*)dy1, ws, N, T, Tp, nct, ncf, F0_, f1);
  } else if (F0_ != F0) {
    hipLaunchKernelGGL(k_conv1_wgrad<bf16_t>, dim3(P), dim3(C1W_THREADS), 0, st, x, (const bf16_t*)dy1, ws, N, T, Tp, nct, ncf, F0_, f1);
  } else {
    static bool attr[DS2_MAX_DEVICES];
    if (ds2_first_use_on_device(attr))
      (void)hipFuncSetAttribute((const void*)k_conv1_wgrad_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, M1W_SMEM);
    P = conv1_wgrad_mfma_blocks(N, Tp) < P ? conv1_wgrad_mfma_blocks(N, Tp) : P;
    hipLaunchKernelGGL(k_conv1_wgrad_mfma, dim3(P), dim3(256), M1W_SMEM, st, x, (const bf16_t*)dy1, ws, N, T, Tp, ds2_cdiv(Tp, M1_TB));
  }
  DS2_CHECK_LAUNCH();
  const int C = K1F * K1T * CH;
  return ds2_colsum(DS2_F32, ws, P, C, C, dw1k, 1.0f, ws + (long)P * C, st_);
}

// scratch of ds2_conv2_fwd: bf16 storage at 161 bins runs the layer as two register-resident tap correlations (even / odd kernel
// rows) whose fp32 partial sums [N][41][Tp][32] pass through it; fp32 storage and other geometries need none
long ds2_conv2_fwd_ws_bytes(int dtype, int N, int F0_, int Tp) { return (dtype == DS2_BF16 && F0_ == F0) ? (long)N * F2 * Tp * CH * 4 : 0; }
int ds2_conv2_fwd(int dtype, const void* a1, const void* w2t, const float* b2, const int* lens, void* y2, int N, int F0_, int Tp, void* ws,
                  ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  int f1, f2;
  DS2_REQUIRE(N > 0 && Tp > 0 && conv_geometry(F0_, f1, f2), DS2_ERR_ARG);
  // (the register-resident tap kernels address their input through 32-bit buffer offsets: beyond 2 GB the general kernel takes over)
  if (dtype == DS2_F32 || F0_ != F0 || (long)N * F1 * Tp * CH * 2 >= (long)CW_OOB) {
    ConvTapArgs a{a1, w2t, b2, lens, y2, N, Tp, f1, f2, K2F, 2, 10, f2, 1, 0};
    return dtype == DS2_F32 ? conv_tap_launch<float>(a, st) : conv_tap_launch<bf16_t>(a, st);
  }
  DS2_REQUIRE(ws != nullptr, DS2_ERR_ARG);
  // kernel rows kf = 2m (11 of them), then kf = 2m + 1 (10): input row of (u, m) = 2*(u + m - 5) + q
  RTapArgs e{(const bf16_t*)a1, (const bf16_t*)w2t, nullptr, nullptr, nullptr, (float*)ws, nullptr, N, Tp, F1, F2, 11, 2, 0, 2, 0, 5, F2, 1, 0};
  RTapArgs o{(const bf16_t*)a1, (const bf16_t*)w2t, b2, lens, (const float*)ws, nullptr, (bf16_t*)y2, N, Tp, F1, F2, 10, 2, 1, 2, 1, 5, F2, 1, 0};
  const int rc = conv_rtap_launch(e, st);
  return rc ? rc : conv_rtap_launch(o, st);
}

// da1[n][f][t][ci] = sum_{co,kf,kt} w2[co][ci][kf][kt] * dy2[n][(f+10-kf)/2][t+5-kt][co]   (only even f+10-kf)
// f = 2u+q: kf = q + 2m ->  input row u + 5 - m, i.e. a stride-1 correlation with the flipped taps m' = M_q-1-m
// (M_0 = 11, M_1 = 10), top padding M_q - 6, and kt' = 10 - kt with padding 5.
//   w2d_q [M_q*11][32 ci][32 co] = w2[co][ci][q + 2*(M_q-1-m')][10 - kt']        (prepared by the binding)
int ds2_conv2_dgrad(int dtype, const void* dy2, const void* w2d_even, const void* w2d_odd, void* da1, int N, int F0_, int Tp,
                    ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  int f1, f2;
  DS2_REQUIRE(N > 0 && Tp > 0 && conv_geometry(F0_, f1, f2), DS2_ERR_ARG);
  if (dtype == DS2_BF16 && F0_ == F0 && (long)N * F1 * Tp * CH * 2 < (long)CW_OOB) {
    RTapArgs e{(const bf16_t*)dy2, (const bf16_t*)w2d_even, nullptr, nullptr, nullptr, nullptr, (bf16_t*)da1, N, Tp, F2, F1, 11, 1, 0, 1, 0, 5, 41, 2, 0};
    RTapArgs o{(const bf16_t*)dy2, (const bf16_t*)w2d_odd, nullptr, nullptr, nullptr, nullptr, (bf16_t*)da1, N, Tp, F2, F1, 10, 1, 0, 1, 0, 4, 40, 2, 1};
    const int rc = conv_rtap_launch(e, st);
    return rc ? rc : conv_rtap_launch(o, st);
  }
  // rows f = 2u + q of da1: (f1 + 1) / 2 even ones, f1 / 2 odd ones
  ConvTapArgs e{dy2, w2d_even, nullptr, nullptr, da1, N, Tp, f2, f1, 11, 1, 5, (f1 + 1) / 2, 2, 0};
  ConvTapArgs o{dy2, w2d_odd, nullptr, nullptr, da1, N, Tp, f2, f1, 10, 1, 4, f1 / 2, 2, 1};
  if (dtype == DS2_F32) {
    const int rc = conv_tap_launch<float>(e, st);
    return (rc || f1 / 2 == 0) ? rc : conv_tap_launch<float>(o, st);
  }
  const int rc = conv_tap_launch<bf16_t>(e, st);
  return (rc || f1 / 2 == 0) ? rc : conv_tap_launch<bf16_t>(o, st);
}

long ds2_conv2_wgrad_ws_floats(int N, int Tp) {
  (void)N; (void)Tp;
  const long C = (long)K2F * K2T * CH * CH;
  static_assert(CWR_SPLITS >= CW_SPLITS, "sized for the larger split count of the two kernels");
  return (long)CWR_SPLITS * C + (long)ds2_norm_partials(CWR_SPLITS) * C;
}
int ds2_conv2_wgrad(int dtype, const void* dy2, const void* a1, float* dw2t, int N, int F0_, int Tp, float* ws, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  int f1, f2;
  DS2_REQUIRE(conv_geometry(F0_, f1, f2), DS2_ERR_ARG);
  dim3 grid(K2F, CW_SPLITS);
  // (the row-group kernel addresses both tensors through 32-bit buffer offsets: beyond 2 GB -- 64 clips of over two minutes -- the
  // general kernel takes over)
  const bool general = dtype == DS2_F32 || F0_ != F0 || (long)N * F1 * Tp * CH * 2 >= (long)CW_OOB;
  if (dtype == DS2_F32) {
    hipLaunchKernelGGL(k_conv2_wgrad<float>, grid, dim3(256), 0, st, (const float*)dy2, (const float*)a1, ws, N, Tp, f1, f2);
  } else if (general) {
    hipLaunchKernelGGL(k_conv2_wgrad<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy2, (const bf16_t*)a1, ws, N, Tp, f1, f2);
  } else {
    static bool attr[DS2_MAX_DEVICES];
    constexpr int smem = cwd_smem_bytes(CWD_KSN);
    if (ds2_first_use_on_device(attr))
      (void)hipFuncSetAttribute((const void*)k_conv2_wgrad_bf16d<CWD_KSN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(k_conv2_wgrad_bf16d<CWD_KSN>, dim3(8 * CWD_UNITS_PER_XCD), dim3(256), smem, st, (const bf16_t*)dy2, (const bf16_t*)a1, ws, N,
                       Tp, CWR_SPLITS);
  }
  DS2_CHECK_LAUNCH();
  const int C = K2F * K2T * CH * CH;
  const int splits = general ? CW_SPLITS : CWR_SPLITS;
  return ds2_colsum(DS2_F32, ws, splits, C, C, dw2t, 1.0f, ws + (long)splits * C, st_);
}

}  // extern "C"
