// NT GEMM on MFMA:  C[M][N] = sum_k A[m][k] * B[n][k]  (+ bias[n])        (both operands K-contiguous)
//
// Used for every dense contraction of the path that is not inside the recurrence:
//   i2h   gi = X * W_ih^T + b_ih            (reference model.py:97-99, torch GRU/LSTM input projection)
//   dgrad dX = dGI * W_ih   (B = W_ih^T)    wgrad dW = dG^T * X (operands pre-transposed)     head model.py:197
// Tile: 128x128 per 256-thread workgroup, K-tile = 128 bytes per row (64 bf16 / 32 f32), 4 waves as 2x2, each wave a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 fp32 regs).  Global -> registers -> LDS (row stride 144 B: the 16-B
// pad makes the 16-lane ds_read_b128 groups hit 16 distinct 16-B slots), double-buffered LDS, one barrier per K-tile,
// next tile's global loads in flight during the MFMAs.  bf16: v_mfma_f32_32x32x16_bf16; f32: v_mfma_f32_32x32x2_f32
// (exact fp32, used by the 1e-3 parity mode).  Roofline: MFMA (2.5 PF bf16 / 157 TF f32); algorithmic flops 2*M*N*K.
#include <stdlib.h>

#include "ds2_common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int KTB = 128;             // bytes of K per row per LDS stage
constexpr int ROWB = KTB + 16;       // padded LDS row stride in bytes
constexpr int STAGE_BYTES = (BM + BN) * ROWB;

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  int M, N, K;
  long lda, ldb, ldc;
  long sA, sB, sC, sBias;  // batch strides (elements)
  int splitk;              // >1: K is split over blockIdx.z % splitk and results are atomically added (f32 C only)
  int out_is_f32;
  const void* A2;          // optional second block of A rows: rows m >= m_split are A2[m - m_split][.] (same lda); null = none
  int m_split;
};

template <typename T>
__global__ void __launch_bounds__(256, 2) k_gemm_nt(GemmArgs g) {
  constexpr int V = Vec16<T>::N;              // elements per 16 bytes
  constexpr int KT = KTB / (int)sizeof(T);    // elements of K per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int batch = blockIdx.z, ks = 0;
  if (g.splitk > 1) {
    ks = blockIdx.z % g.splitk;
    batch = blockIdx.z / g.splitk;
  }
  const T* A = (const T*)g.A + (long)batch * g.sA;
  const T* B = (const T*)g.B + (long)batch * g.sB;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  // K range of this block
  int nkt_total = (g.K + KT - 1) / KT;
  int kt_begin = 0, kt_end = nkt_total;
  if (g.splitk > 1) {
    int per = (nkt_total + g.splitk - 1) / g.splitk;
    kt_begin = ks * per;
    kt_end = min(nkt_total, kt_begin + per);
  }

  ds2_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Staging registers: 4 x 16 B of A and of B per thread and K-tile.  Loads are UNCONDITIONAL so that all eight stay in
  // flight together (a predicated load compiles to a branch + s_waitcnt vmcnt(0) per load and serialises the pipeline):
  // row addresses are clamped into the matrix (rows >= M / N just repeat the last row; their results are never stored)
  // and the K tail is zeroed with a bit mask.
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  uint32_t kmask = 0;   // K-tail mask of the tile currently held in the staging registers (applied at LDS-write time)
  const int ld_row = tid >> 3, ld_kc = (tid & 7) * V;
  const T* a_row[4];
  const T* b_row[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = ld_row + i * 32;
    const int ar = min(m0 + row, g.M - 1);
    a_row[i] = (g.A2 && ar >= g.m_split) ? (const T*)g.A2 + (long)batch * g.sA + (long)(ar - g.m_split) * g.lda : A + (long)ar * g.lda;
    b_row[i] = B + (long)min(n0 + row, g.N - 1) * g.ldb;
  }
#define DS2_MASK4(r, m) r.x &= m; r.y &= m; r.z &= m; r.w &= m;
#define DS2_GLOAD(kt)                                                              \
  {                                                                                \
    const int k_ = (kt) * KT + ld_kc;                                              \
    const bool kok_ = k_ < g.K;                                                    \
    const int kc_ = kok_ ? k_ : 0;                                                 \
    ra0 = *reinterpret_cast<const uint4*>(a_row[0] + kc_);                         \
    ra1 = *reinterpret_cast<const uint4*>(a_row[1] + kc_);                         \
    ra2 = *reinterpret_cast<const uint4*>(a_row[2] + kc_);                         \
    ra3 = *reinterpret_cast<const uint4*>(a_row[3] + kc_);                         \
    rb0 = *reinterpret_cast<const uint4*>(b_row[0] + kc_);                         \
    rb1 = *reinterpret_cast<const uint4*>(b_row[1] + kc_);                         \
    rb2 = *reinterpret_cast<const uint4*>(b_row[2] + kc_);                         \
    rb3 = *reinterpret_cast<const uint4*>(b_row[3] + kc_);                         \
    kmask = kok_ ? 0xffffffffu : 0u;                                               \
  }
#define DS2_SWRITE(stage)                                                          \
  {                                                                                \
    unsigned char* sa_ = smem + (stage) * STAGE_BYTES + ld_row * ROWB + (tid & 7) * 16; \
    unsigned char* sb_ = sa_ + BM * ROWB;                                          \
    DS2_MASK4(ra0, kmask) DS2_MASK4(ra1, kmask) DS2_MASK4(ra2, kmask) DS2_MASK4(ra3, kmask) \
    DS2_MASK4(rb0, kmask) DS2_MASK4(rb1, kmask) DS2_MASK4(rb2, kmask) DS2_MASK4(rb3, kmask) \
    *reinterpret_cast<uint4*>(sa_) = ra0;                                          \
    *reinterpret_cast<uint4*>(sa_ + 32 * ROWB) = ra1;                              \
    *reinterpret_cast<uint4*>(sa_ + 64 * ROWB) = ra2;                              \
    *reinterpret_cast<uint4*>(sa_ + 96 * ROWB) = ra3;                              \
    *reinterpret_cast<uint4*>(sb_) = rb0;                                          \
    *reinterpret_cast<uint4*>(sb_ + 32 * ROWB) = rb1;                              \
    *reinterpret_cast<uint4*>(sb_ + 64 * ROWB) = rb2;                              \
    *reinterpret_cast<uint4*>(sb_ + 96 * ROWB) = rb3;                              \
  }

  if (kt_begin < kt_end) {
    DS2_GLOAD(kt_begin);
    DS2_SWRITE(0);
  }
  __syncthreads();
  const int li = lane & 31, lq = lane >> 5;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int stage = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    if (more) DS2_GLOAD(kt + 1);
    const unsigned char* sa = smem + stage * STAGE_BYTES + (wm * 64 + li) * ROWB + lq * 16;
    const unsigned char* sb = smem + stage * STAGE_BYTES + BM * ROWB + (wn * 64 + li) * ROWB + lq * 16;
#pragma unroll
    for (int c = 0; c < KTB / 32; ++c) {
      uint4 a0 = *reinterpret_cast<const uint4*>(sa + c * 32);
      uint4 a1 = *reinterpret_cast<const uint4*>(sa + 32 * ROWB + c * 32);
      uint4 b0 = *reinterpret_cast<const uint4*>(sb + c * 32);
      uint4 b1 = *reinterpret_cast<const uint4*>(sb + 32 * ROWB + c * 32);
      Mma<T>::mma32(acc[0][0], a0, b0);
      Mma<T>::mma32(acc[0][1], a0, b1);
      Mma<T>::mma32(acc[1][0], a1, b0);
      Mma<T>::mma32(acc[1][1], a1, b1);
    }
    if (more) DS2_SWRITE(stage ^ 1);
    __syncthreads();
  }

  // epilogue
  const float* bias = g.bias ? g.bias + (long)batch * g.sBias : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
      if (col >= g.N) continue;
      const float bv = (bias && ks == 0) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + mma32_row(r, lane);
        if (row >= g.M) continue;
        const float v = acc[i][j][r] + bv;
        const long off = (long)batch * g.sC + (long)row * g.ldc + col;
        if (g.out_is_f32) {
          float* cp = (float*)g.C + off;
          if (g.splitk > 1)
            atomicAdd(cp, v);
          else
            *cp = v;
        } else {
          stf((T*)g.C + off, v);
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// bf16 fast path: the same 128x128x64 tile, staged with direct global -> LDS DMA (global_load_lds_dwordx4): no staging
// VGPRs, no ds_write pass.  The DMA writes lane-linear (wave-uniform base + lane*16 B), so an LDS row is the plain 128-byte
// K-slice of a matrix row; the bank-conflict fix is an XOR swizzle applied to the per-lane SOURCE address and to the
// fragment read address (both sides or neither, guide rule 21): 16-byte chunk c of tile row r lives in slot
// c ^ ((r >> 1) & 7) -- the 16 rows of every ds_read_b128 lane group then hit 16 distinct slots of the 256-byte bank row.
// Two LDS stages; iteration kt: wait for stage kt (vmcnt(0) + barrier), issue the DMA of tile kt+1, MFMA on tile kt.
// Requires K % 64 == 0 and 16-byte aligned rows (the binding pads K); rows past M / N re-read the last row.
// Workgroup ids are remapped so that each XCD (own L2) works on a contiguous band of the tile grid.
// (Measured and rejected: four stages at one workgroup per CU with asm-issued DMA and counted vmcnt waits -- 456-695
// TFLOP/s on the step's shapes against 566-798 for this two-stage, two-workgroups-per-CU form.)
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <bool LOWREG>
__global__ void __launch_bounds__(256, 2) k_gemm_nt_bf16_glds(GemmArgs g) {
  constexpr int KT = 64;                         // elements of K per stage (128 bytes per row)
  constexpr int ROW = 128;                       // LDS bytes per tile row
  constexpr int STAGE = (BM + BN) * ROW;         // 32 KiB
  // !LOWREG: TWO separate LDS objects, one per stage, and a K loop unrolled by two so that every access names its stage
  // statically: the compiler can then prove that the DMA into one stage does not alias the fragment reads of the other and
  // leaves the DMA in flight under the MFMAs (with one array + a run-time stage index it drains vmcnt(0) before the first
  // ds_read).  LOWREG: that one-array form -- slower in isolation (no in-block overlap) but ~115 instead of ~196 VGPRs, which
  // lets its waves share a SIMD with the persistent recurrent kernels (used for the weight-gradient GEMMs on the second stream).
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware remap of the (x, y) tile id (bijective for any grid size)
  const int gx = gridDim.x, nblk = gridDim.x * gridDim.y;
  int id = blockIdx.y * gx + blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = id % 8, k = id / 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int bx = id % gx, by = id / gx;
  int batch = blockIdx.z, ks = 0;
  if (g.splitk > 1) {
    ks = blockIdx.z % g.splitk;
    batch = blockIdx.z / g.splitk;
  }
  const bf16_t* A = (const bf16_t*)g.A + (long)batch * g.sA;
  const bf16_t* B = (const bf16_t*)g.B + (long)batch * g.sB;
  const int m0 = by * BM, n0 = bx * BN;
  const int nkt_total = g.K / KT;
  int kt_begin = 0, kt_end = nkt_total;
  if (g.splitk > 1) {
    const int per = (nkt_total + g.splitk - 1) / g.splitk;
    kt_begin = ks * per;
    kt_end = min(nkt_total, kt_begin + per);
  }

  ds2_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA identity: wave w stages tile rows [32w, 32w+32) of A and of B, 8 rows (1 KiB) per instruction
  const int r_in = lane >> 3, slot = lane & 7;
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 32 + j * 8 + r_in;
    const int chunk = slot ^ ((r >> 1) & 7);
    const int ar = min(m0 + r, g.M - 1);
    a_src[j] = ((g.A2 && ar >= g.m_split) ? (const bf16_t*)g.A2 + (long)batch * g.sA + (long)(ar - g.m_split) * g.lda : A + (long)ar * g.lda) +
               chunk * 8;
    b_src[j] = B + (long)min(n0 + r, g.N - 1) * g.ldb + chunk * 8;
  }
  auto dma = [&](unsigned char* st, int kt) {
    unsigned char* sa = st + wave * 32 * ROW;
    unsigned char* sb = sa + BM * ROW;
    const int k0 = kt * KT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(a_src[j] + k0), (lds_ptr_t)(sa + j * 8 * ROW), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(b_src[j] + k0), (lds_ptr_t)(sb + j * 8 * ROW), 16, 0, 0);
    }
  };

  const int li = lane & 31, lq = lane >> 5;
  const int key = (li >> 1) & 7;                 // rows wm*64 + i*32 + li: the swizzle key depends on li only
  auto compute = [&](const unsigned char* st) {
    const unsigned char* sa = st + (wm * 64 + li) * ROW;
    const unsigned char* sb = st + BM * ROW + (wn * 64 + li) * ROW;
#pragma unroll
    for (int c = 0; c < 4; ++c) {                 // k-step of 16 elements = chunks 2c (lq = 0) and 2c+1 (lq = 1)
      const int off = ((2 * c + lq) ^ key) * 16;
      const uint4 a0 = *reinterpret_cast<const uint4*>(sa + off);
      const uint4 a1 = *reinterpret_cast<const uint4*>(sa + 32 * ROW + off);
      const uint4 b0 = *reinterpret_cast<const uint4*>(sb + off);
      const uint4 b1 = *reinterpret_cast<const uint4*>(sb + 32 * ROW + off);
      // operands swapped on purpose: D rows = n (the B matrix' rows), D cols = m, so that a lane holds 4 CONSECUTIVE n
      // of one output row m per register group -> 8-byte (bf16) / 16-byte (f32) stores instead of 2-byte ones
      Mma<bf16_t>::mma32(acc[0][0], b0, a0);
      Mma<bf16_t>::mma32(acc[0][1], b1, a0);
      Mma<bf16_t>::mma32(acc[1][0], b0, a1);
      Mma<bf16_t>::mma32(acc[1][1], b1, a1);
    }
  };
  if constexpr (LOWREG) {
    if (kt_begin < kt_end) dma(smem_dyn, kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int stage = (kt - kt_begin) & 1;
      __syncthreads();
      if (kt + 1 < kt_end) dma(smem_dyn + (stage ^ 1) * STAGE, kt + 1);
      compute(smem_dyn + stage * STAGE);
    }
  } else {
    __shared__ __attribute__((aligned(16))) unsigned char stage0[STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char stage1[STAGE];
    if (kt_begin < kt_end) dma(stage0, kt_begin);
    for (int kt = kt_begin; kt < kt_end;) {
      __syncthreads();                            // stage0 landed (DMA queue drained ahead of the barrier); stage1 free
      if (kt + 1 < kt_end) dma(stage1, kt + 1);
      compute(stage0);
      if (++kt >= kt_end) break;
      __syncthreads();
      if (kt + 1 < kt_end) dma(stage0, kt + 1);
      compute(stage1);
      ++kt;
    }
  }

  const float* bias = g.bias ? g.bias + (long)batch * g.sBias : nullptr;
  const bool n_vec_ok = (g.ldc % 4 == 0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wm * 64 + i * 32 + li;                 // output row m of this lane
    if (row >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {                              // register group: columns n = base + 8q + 4lq + {0..3}
        const int col = n0 + wn * 64 + j * 32 + 8 * q + 4 * lq;
        if (col >= g.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] + ((bias && ks == 0 && col + e < g.N) ? bias[col + e] : 0.f);
        const long off = (long)batch * g.sC + (long)row * g.ldc + col;
        if (g.out_is_f32) {
          float* cp = (float*)g.C + off;
          if (g.splitk > 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < g.N) atomicAdd(cp + e, v[e]);
          } else if (col + 3 < g.N && n_vec_ok) {
            *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < g.N) cp[e] = v[e];
          }
        } else {
          bf16_t* cp = (bf16_t*)g.C + off;
          if (col + 3 < g.N && n_vec_ok) {
            uint2 pk;
            pk.x = cvt_pk_bf16(v[0], v[1]);
            pk.y = cvt_pk_bf16(v[2], v[3]);
            *reinterpret_cast<uint2*>(cp) = pk;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < g.N) stf(cp + e, v[e]);
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Large-tile variant of the DMA-staged kernel for the big GEMMs of the step (input projections, dX): TBM x TBN tile per
// workgroup of 8 or 16 waves (WM x WN, each wave (TBM/WM) x (TBN/WN) = MI x NJ MFMA 32x32 accumulators), ONE workgroup
// per CU, same two-stage global_load_lds pipeline and XOR-swizzled 128-byte rows.  Against the 128x128 tile: 2-4x the MFMA work
// per staged byte (L2 -> LDS traffic per flop halves) and per barrier.  Same contract (K % 64 == 0, 16-byte aligned rows).
// ------------------------------------------------------------------------------------------------------------------
template <int TBM, int TBN, int WM, int WN>
__global__ void __launch_bounds__(WM * WN * 64, WM * WN / 4) k_gemm_nt_bf16_big(GemmArgs g) {
  constexpr int KT = 64, ROW = 128, NW = WM * WN;
  constexpr int STAGE = (TBM + TBN) * ROW;
  constexpr int MI = TBM / WM / 32, NJ = TBN / WN / 32;
  constexpr int RA = TBM / NW, RB = TBN / NW;            // tile rows each wave stages (8 rows = 1 KiB per DMA instruction)
  static_assert((NW == 8 || NW == 16) && RA % 8 == 0 && RB % 8 == 0 && MI >= 1 && NJ >= 1, "tile / wave decomposition");
  __shared__ __attribute__((aligned(16))) unsigned char stage0[STAGE];
  __shared__ __attribute__((aligned(16))) unsigned char stage1[STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int gx = gridDim.x, nblk = gridDim.x * gridDim.y;
  int id = blockIdx.y * gx + blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = id % 8, k = id / 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int bx = id % gx, by = id / gx;
  const int batch = blockIdx.z;
  const bf16_t* A = (const bf16_t*)g.A + (long)batch * g.sA;
  const bf16_t* B = (const bf16_t*)g.B + (long)batch * g.sB;
  const int m0 = by * TBM, n0 = bx * TBN;
  const int kt_end = g.K / KT;

  ds2_f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int r_in = lane >> 3, slot = lane & 7;
  // per-lane source offsets in ELEMENTS from the (uniform) operand base: 32-bit, half the registers of pointers (the host
  // checks that the operands span < 2^31 elements)
  int a_off[RA / 8], b_off[RB / 8];
#pragma unroll
  for (int j = 0; j < RA / 8; ++j) {
    const int r = wave * RA + j * 8 + r_in;
    a_off[j] = (int)((long)min(m0 + r, g.M - 1) * g.lda) + (slot ^ ((r >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int j = 0; j < RB / 8; ++j) {
    const int r = wave * RB + j * 8 + r_in;
    b_off[j] = (int)((long)min(n0 + r, g.N - 1) * g.ldb) + (slot ^ ((r >> 1) & 7)) * 8;
  }
  auto dma = [&](unsigned char* st, int kt) {
    unsigned char* sa = st + wave * RA * ROW;
    unsigned char* sb = st + TBM * ROW + wave * RB * ROW;
    const bf16_t* Ak = A + kt * KT;
    const bf16_t* Bk = B + kt * KT;
#pragma unroll
    for (int j = 0; j < RA / 8; ++j) __builtin_amdgcn_global_load_lds((glb_ptr_t)(Ak + a_off[j]), (lds_ptr_t)(sa + j * 8 * ROW), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < RB / 8; ++j) __builtin_amdgcn_global_load_lds((glb_ptr_t)(Bk + b_off[j]), (lds_ptr_t)(sb + j * 8 * ROW), 16, 0, 0);
  };
  const int li = lane & 31, lq = lane >> 5;
  const int key = (li >> 1) & 7;
  auto compute = [&](const unsigned char* st) {
    const unsigned char* sa = st + (wm * (TBM / WM) + li) * ROW;
    const unsigned char* sb = st + TBM * ROW + (wn * (TBN / WN) + li) * ROW;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int off = ((2 * c + lq) ^ key) * 16;
      uint4 a[MI], b[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const uint4*>(sa + i * 32 * ROW + off);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const uint4*>(sb + j * 32 * ROW + off);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) Mma<bf16_t>::mma32(acc[i][j], b[j], a[i]);   // operands swapped: see k_gemm_nt_bf16_glds
    }
  };
  if (kt_end > 0) dma(stage0, 0);
  for (int kt = 0; kt < kt_end;) {
    __syncthreads();
    if (kt + 1 < kt_end) dma(stage1, kt + 1);
    compute(stage0);
    if (++kt >= kt_end) break;
    __syncthreads();
    if (kt + 1 < kt_end) dma(stage0, kt + 1);
    compute(stage1);
    ++kt;
  }

  const float* bias = g.bias ? g.bias + (long)batch * g.sBias : nullptr;
  const bool n_vec_ok = (g.ldc % 4 == 0);
  // the epilogue's address arithmetic must not be hoisted above the K loop (it would sit in registers the accumulators need):
  // route the lane coordinates through an opaque move after the loop
  int eli = li, elq = lq;
  asm volatile("" : "+v"(eli), "+v"(elq));
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = m0 + wm * (TBM / WM) + i * 32 + eli;
    if (row >= g.M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn * (TBN / WN) + j * 32 + 8 * q + 4 * elq;
        if (col >= g.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] + ((bias && col + e < g.N) ? bias[col + e] : 0.f);
        const long off = (long)batch * g.sC + (long)row * g.ldc + col;
        if (g.out_is_f32) {
          float* cp = (float*)g.C + off;
          if (col + 3 < g.N && n_vec_ok) {
            *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < g.N) cp[e] = v[e];
          }
        } else {
          bf16_t* cp = (bf16_t*)g.C + off;
          if (col + 3 < g.N && n_vec_ok) {
            uint2 pk;
            pk.x = cvt_pk_bf16(v[0], v[1]);
            pk.y = cvt_pk_bf16(v[2], v[3]);
            *reinterpret_cast<uint2*>(cp) = pk;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < g.N) stf(cp + e, v[e]);
          }
        }
      }
    }
  }
}

// 1 = use the 256x128 tile.  Measured on the step's shapes (tools/bench_gemm.py, profiles/r02c_gemm_tiles.txt): input projections
// (M = 24032, N = 6144) 616-660 vs 573-597 TFLOP/s with the 128x128 tile; narrow outputs (N = 1024: dX, weight gradients) are
// faster on the 128x128 tile (more workgroups in the tail); a 256x256 tile needs 128 accumulator registers per wave, which hipcc
// duplicates across the two-stage loop (spills) -- rejected.
int big_tile_variant(int M, int N, int K) {
  if (M < 4096 || N < 2048 || K < 256 || (long)M * K >= (1L << 31) || (long)N * K >= (1L << 31)) return 0;
  return 1;
}

}  // namespace

extern "C" {

// C (f32 if out_f32 else `dtype`) [M][ldc] = A[M][lda] * B[N][ldb]^T (+bias[N]); batched over `batch` with element strides.
// K, lda, ldb must be multiples of 16 bytes / sizeof(T); A/B 16-byte aligned.  splitk>1 requires out_f32; C is zeroed here
// (memset nodes on the stream) and the K-slices are atomically accumulated.
static int gemm_nt_impl(int dtype, const void* A, const void* A2, int m_split, const void* B, void* C, const float* bias, int M, int N, int K, long lda,
                        long ldb, long ldc, int out_f32, int batch, long strideA, long strideB, long strideC, long strideBias,
                        int splitk, bool coresident, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(dtype == DS2_F32 || dtype == DS2_BF16, DS2_ERR_DTYPE);
  const int V = dtype == DS2_F32 ? 4 : 8;
  DS2_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && splitk >= 1, DS2_ERR_ARG);
  DS2_REQUIRE(K % V == 0 && lda % V == 0 && ldb % V == 0 && strideA % V == 0 && strideB % V == 0, DS2_ERR_ALIGN);
  DS2_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0, DS2_ERR_ALIGN);
  DS2_REQUIRE(splitk == 1 || out_f32, DS2_ERR_ARG);
  GemmArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, strideBias, splitk, out_f32 || dtype == DS2_F32, A2, m_split};
  DS2_REQUIRE(A2 == nullptr || (m_split > 0 && m_split < M && (((uintptr_t)A2) & 15) == 0), DS2_ERR_ARG);
  dim3 grid(ds2_cdiv(N, BN), ds2_cdiv(M, BM), batch * splitk), blk(256);
  const size_t shm = 2 * STAGE_BYTES;
  if (splitk > 1) {
    for (int b = 0; b < batch; ++b) {
      hipError_t e = hipMemset2DAsync((float*)C + (long)b * strideC, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, st);
      if (e != hipSuccess) return (int)e;
    }
  }
  if (dtype == DS2_F32) {
    static bool attr_f[DS2_MAX_DEVICES];
    if (ds2_first_use_on_device(attr_f)) (void)hipFuncSetAttribute((const void*)k_gemm_nt<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(k_gemm_nt<float>, grid, blk, shm, st, g);
  } else if (splitk == 1 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0) {
    // (split-K accumulates with atomics: the register-staged kernel below keeps them coalesced along n)
    g.out_is_f32 = out_f32;
    const int big = (coresident || A2) ? 0 : big_tile_variant(M, N, K);
    if (big == 1) {
      hipLaunchKernelGGL((k_gemm_nt_bf16_big<256, 128, 4, 2>), dim3(ds2_cdiv(N, 128), ds2_cdiv(M, 256), batch), dim3(512), 0, st, g);
    } else if (coresident) {
      static bool attr_lr[DS2_MAX_DEVICES];
      const size_t shm_lr = 2 * (BM + BN) * 128;
      if (ds2_first_use_on_device(attr_lr))
        (void)hipFuncSetAttribute((const void*)k_gemm_nt_bf16_glds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_lr);
      hipLaunchKernelGGL(k_gemm_nt_bf16_glds<true>, grid, blk, shm_lr, st, g);
    } else {
      hipLaunchKernelGGL(k_gemm_nt_bf16_glds<false>, grid, blk, 0, st, g);
    }
  } else {
    static bool attr_b[DS2_MAX_DEVICES];
    if (ds2_first_use_on_device(attr_b)) (void)hipFuncSetAttribute((const void*)k_gemm_nt<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(k_gemm_nt<bf16_t>, grid, blk, shm, st, g);
  }
  DS2_CHECK_LAUNCH();
  return 0;
}

int ds2_gemm_nt(int dtype, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb,
                long ldc, int out_f32, int batch, long strideA, long strideB, long strideC, long strideBias, int splitk,
                ds2_stream_t st) {
  return gemm_nt_impl(dtype, A, nullptr, 0, B, C, bias, M, N, K, lda, ldb, ldc, out_f32, batch, strideA, strideB, strideC, strideBias, splitk,
                      false, st);
}

// Same contract; picks the low-register kernel variant whose waves can share a SIMD with the persistent recurrent kernels
// (for GEMMs issued on a second stream while a recurrent sweep owns the CUs).
int ds2_gemm_nt_coresident(int dtype, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda,
                           long ldb, long ldc, int out_f32, int batch, long strideA, long strideB, long strideC, long strideBias,
                           int splitk, ds2_stream_t st) {
  return gemm_nt_impl(dtype, A, nullptr, 0, B, C, bias, M, N, K, lda, ldb, ldc, out_f32, batch, strideA, strideB, strideC, strideBias, splitk,
                      true, st);
}

// A given as two row blocks: rows [0, m_split) from A, rows [m_split, M) from A2 (same lda) -- the GRU's hidden-side gate gradient
// [dr, dz | dQ] lives in two buffers (ds2_rnn_persist_bwd).  coresident != 0 selects the low-register kernel.
int ds2_gemm_nt_rows2(int dtype, const void* A, const void* A2, int m_split, const void* B, void* C, int M, int N, int K, long lda,
                      long ldb, long ldc, int out_f32, int splitk, int coresident, ds2_stream_t st) {
  return gemm_nt_impl(dtype, A, A2, m_split, B, C, nullptr, M, N, K, lda, ldb, ldc, out_f32, 1, 0, 0, 0, 0, splitk, coresident != 0, st);
}

}  // extern "C"
