// 256x256 bf16 GEMM with a phase-split, role-alternating schedule (the large dense contractions of the step):
//   NT:  C[M][N] = sum_k A[m][k] * B[n][k] (+ bias[n])      both operands K-contiguous   (input projections, dX;  model.py:97-99)
//   TN:  C[M][N] = sum_k At[k][m] * Bt[k][n]                both operands K-major        (weight gradients: contraction over the
//        T'*N rows of dGI / X / h -- no operand transposes; grouped: several products per launch)
//
// Geometry: 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the tile = 4 x 2 accumulators of
// v_mfma_f32_32x32x16_bf16 (128 registers).  K-tile 64.  LDS: two K-tile buffers of 64 KiB (A 32 KiB | B 32 KiB), filled by
// global_load_lds_dwordx4 (lane-linear destination; the bank-conflict XOR lives in the per-lane SOURCE address and in the
// fragment read address -- both or neither).
//
// Schedule (one K-tile = four phases, one quadrant 64 x 32 of the wave's tile each = 8 MFMAs = 256 matrix-pipe cycles):
//   phase 0   read A(m-half 0) [8 x ds_read_b128] + B(n-half 0) [4]     | MFMA (0,0)
//   phase 1   read B(n-half 1) [4]                                      | MFMA (0,1)      B is dead in LDS after this phase
//   phase 2   read A(m-half 1) [8];  DMA B of K-tile t+2 [4 x 1 KiB]    | MFMA (1,1)      A is dead in LDS after this phase
//   phase 3   DMA A of K-tile t+2 [4];  s_waitcnt vmcnt(8)              | MFMA (1,0)      (B(n-half 0) kept in registers)
// Every phase is  { loads ; s_barrier ; MFMAs ; s_barrier }.  The waves of M-half 1 (waves 4-7: the SIMD partners of waves 0-3)
// run ONE barrier behind, so between two consecutive barriers one wave of every SIMD is in its MFMA section while its partner
// issues LDS reads / DMA: the matrix pipe of a SIMD is fed by one wave at a time, back to back.
// Hazards, by counters and barriers only (never by timing):
//   RAW  a wave waits for ITS OWN DMA pieces (vmcnt(8) in phase 3 leaves exactly the 8 pieces of K-tile t+2 in flight, so K-tile
//        t+1 has landed); both barriers of phase 3 lie between every wave's wait and the first read of K-tile t+1.
//   WAR  the reads of phases 1 / 2 are retired (lgkmcnt(0)) BEFORE the phase's first barrier; the DMA that overwrites B / A is
//        issued one phase later, i.e. behind a barrier that every wave -- of either half -- reaches only after that wait.
// Requirements: K % 64 == 0, 16-byte aligned rows, operands < 2^31 elements; edge tiles clamp their source rows.
#include "ds2_common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int G8_MAX_GROUPS = 6;

struct G8Problem {
  const bf16_t* A;    // NT: [M][lda]   TN: [K][lda] (m contiguous)
  const bf16_t* A2;   // TN only: rows (of C) >= m_split read their A columns from A2 (column index m - m_split, row stride lda2); null = none
  const bf16_t* B;    // NT: [N][ldb]   TN: [K][ldb]
  void* C;
  const float* bias;
  int M, N, m_split;
  long lda, lda2, ldb, ldc;   // lda2: leading dimension of A2
  int tiles_n, tile_end;   // tiles of this problem are [previous tile_end, tile_end)
  int out_is_f32;
  int K, tn;               // contraction length; tn != 0: both operands K-major
  const int* rows;         // row list (null = every row): NT: tile row r is row rows[r] of A and of C (M = length of the list);
                           // TN: contraction index k is row rows[k] of both operands (K = length of the list)
};
struct G8Args {
  G8Problem p[G8_MAX_GROUPS];
  int n_problems;
};

#define G8_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define G8_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define G8_BARRIER()                        \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)

// One output tile.  TN is a template parameter of the body; a launch that mixes NT and TN problems (k_gemm8<2>) selects the body
// per workgroup with a wave-uniform branch.
template <bool TN, bool SPREAD>
__device__ __forceinline__ void g8_tile(const G8Problem& P, int t_id, unsigned char* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // Tile order inside an XCD's run of ids.  Row-major (n fastest) puts the 32 tiles that an XCD's CUs work on at one time on
  // 32 / tiles_n rows: with the input projection's 24 column tiles that is 1.3 rows -- 2 A panels + 24 B panels through a 4 MB L2
  // per 32 tiles.  DS2_G8_SWZ = GM > 1 walks stripes of GM tile rows column by column: 32 consecutive ids = GM x (32 / GM) tiles
  // (4 x 8: 12 panels), and the stripe's A panels stay in L2 for its later columns (round 6 A/B: profiles/r06h_gemm8_swizzle.txt).
#ifndef DS2_G8_SWZ
#define DS2_G8_SWZ 1
#endif
  int tm_ = t_id / P.tiles_n, tn_ = t_id % P.tiles_n;
  if (DS2_G8_SWZ > 1 && P.tiles_n > 32 / DS2_G8_SWZ) {
    const int tiles_m = ((TN ? P.M : P.M) + 255) / 256;
    const int per = DS2_G8_SWZ * P.tiles_n, st = t_id / per, wi = t_id - st * per;
    const int rows = min(DS2_G8_SWZ, tiles_m - st * DS2_G8_SWZ);
    tm_ = st * DS2_G8_SWZ + wi % rows;
    tn_ = wi / rows;
  }
  const int m0 = tm_ * 256, n0 = tn_ * 256;
  const int K_ = P.K;
  const int nkt = (K_ + 63) / 64;                       // NT: K % 64 == 0 (host check); TN: a ragged last K-tile is zero-filled in LDS
  const int krem = K_ - (nkt - 1) * 64;                 // valid k-rows of the last K-tile (1..64)
  const bool tail = TN && krem < 64;

  ds2_f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- DMA identity.  One instruction of the whole workgroup moves 8 KiB; wave w owns the 1 KiB piece w of it.
  //  NT: piece = 8 tile rows x 128 B (64 k): lane -> row (lane >> 3), 16-byte slot (lane & 7); the slot holds source chunk
  //      slot ^ ((row >> 1) & 7).                                     4 instructions = the 256 rows of an operand tile.
  //  TN: LDS rows are k (64 per tile), 512 B = 256 m each: piece = 2 k-rows: lane -> k-row (lane >> 5), slot (lane & 31); the
  //      slot holds source chunk slot ^ ((k & 3) << 2) (a 64-byte rotation: the four k-rows of a transposing read then sit in
  //      four different bank quarters).                               4 instructions = the 64 k-rows of an operand tile.
  uint32_t a_off[4], b_off[4];                 // BYTE offsets from the (wave-uniform) operand base: saddr + 32-bit voffset addressing
  const bf16_t* a_base = P.A;
  const bf16_t* b_base = P.B;
  if constexpr (!TN) {
    const int rloc = wave * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rloc >> 1) & 7);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      int ar = min(m0 + rb * 64 + rloc, P.M - 1);
      if (P.rows) ar = P.rows[ar];
      a_off[rb] = (uint32_t)(((long)ar * P.lda + chunk * 8) * 2);
      b_off[rb] = (uint32_t)(((long)min(n0 + rb * 64 + rloc, P.N - 1) * P.ldb + chunk * 8) * 2);
    }
  } else {
    // every lane's m / n range (8 elements) must be inside the operand row: M, N multiples of 8 (checked on the host); ranges
    // past the edge re-read the last 8 columns (their products are never stored)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int kr = rb * 16 + wave * 2 + (lane >> 5);
      const int chunk = (lane & 31) ^ ((kr & 3) << 2);
      int mcol = m0 + chunk * 8, ncol = min(n0 + chunk * 8, P.N - 8);
      const int mlim = (P.A2 && m0 >= P.m_split) ? P.M - P.m_split : (P.A2 ? P.m_split : P.M);
      if (P.A2 && m0 >= P.m_split) mcol -= P.m_split;
      mcol = min(mcol, mlim - 8);
      a_off[rb] = (uint32_t)(((long)kr * ((P.A2 && m0 >= P.m_split) ? P.lda2 : P.lda) + mcol) * 2);
      b_off[rb] = (uint32_t)(((long)kr * P.ldb + ncol) * 2);
    }
    if (P.A2 && m0 >= P.m_split) a_base = P.A2;
  }
  const long lda_eff = (TN && P.A2 && m0 >= P.m_split) ? P.lda2 : P.lda;
  // TN, ragged last K-tile: k-rows past K re-read row K - 1 (never out of bounds; they are zeroed in LDS before use)
  const int t_kr0 = wave * 2 + (lane >> 5);
  const uint32_t t_acol = TN ? a_off[0] - (uint32_t)((long)t_kr0 * ((P.A2 && m0 >= P.m_split) ? P.lda2 : P.lda) * 2) : 0;
  const uint32_t t_bcol = TN ? b_off[0] - (uint32_t)((long)t_kr0 * P.ldb * 2) : 0;
  // LDS map: A of buffer b at b * 32 KiB, B of buffer b at 64 KiB + b * 32 KiB (every fragment read = one per-lane base register +
  // a 16-bit immediate)
  const long a_kstep = TN ? 128 * lda_eff : 128, b_kstep = TN ? 128 * P.ldb : 128;      // bytes per K-tile
  // TN over a row list: the physical rows of this lane's four k-rows of the K-tile that is staged next (scalar loads: two list
  // entries per piece and wave, selected per half-wave; indices past the list re-read its last entry and are zeroed in LDS)
  const bool listed = TN && P.rows != nullptr;
  typedef const __attribute__((address_space(4))) int* const_int_ptr_t;      // constant address space: wave-uniform reads become
  const const_int_ptr_t crows = (const_int_ptr_t)(uintptr_t)P.rows;           // s_load (lgkmcnt), not vector loads behind the DMA
  // fetch_rows issues the scalar loads (phase 0 of the K-tile two ahead of the one they belong to); select_rows consumes them two
  // phases later, right in front of the DMA that needs them -- so the loads' latency never sits in front of a phase's LDS reads
  uint32_t rk[4] = {0, 0, 0, 0};
  int sr0[4] = {0, 0, 0, 0}, sr1[4] = {0, 0, 0, 0};
  auto fetch_rows = [&](int kt) {
    if (listed && kt < nkt) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const int k = kt * 64 + rb * 16 + wave * 2;
        sr0[rb] = crows[min(k, K_ - 1)];
        sr1[rb] = crows[min(k + 1, K_ - 1)];
      }
    }
  };
  auto select_rows = [&]() {
    if (listed) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) rk[rb] = (uint32_t)((lane >> 5) ? sr1[rb] : sr0[rb]);
    }
  };
  const uint32_t lda2b = (uint32_t)(lda_eff * 2), ldb2b = (uint32_t)(P.ldb * 2);
  auto stage_a = [&](int b, int kt, int r0 = 0, int r1 = 4) {
    const unsigned char* src = (const unsigned char*)a_base + (long)kt * a_kstep;
    unsigned char* d = lds + b * 32768 + wave * 1024;
    if (listed) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        if (rb >= r0 && rb < r1)
          __builtin_amdgcn_global_load_lds((glb_ptr_t)((const unsigned char*)a_base + (size_t)(rk[rb] * lda2b + t_acol)), (lds_ptr_t)(d + rb * 8192), 16, 0, 0);
    } else if (tail && kt == nkt - 1) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        if (rb >= r0 && rb < r1) {
          const uint32_t off = (uint32_t)((long)min(t_kr0 + rb * 16, krem - 1) * lda_eff * 2) + t_acol;
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)off), (lds_ptr_t)(d + rb * 8192), 16, 0, 0);
        }
    } else {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        if (rb >= r0 && rb < r1) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)a_off[rb]), (lds_ptr_t)(d + rb * 8192), 16, 0, 0);
    }
  };
  auto stage_b = [&](int b, int kt, int r0 = 0, int r1 = 4) {
    const unsigned char* src = (const unsigned char*)b_base + (long)kt * b_kstep;
    unsigned char* d = lds + 65536 + b * 32768 + wave * 1024;
    if (listed) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        if (rb >= r0 && rb < r1)
          __builtin_amdgcn_global_load_lds((glb_ptr_t)((const unsigned char*)b_base + (size_t)(rk[rb] * ldb2b + t_bcol)), (lds_ptr_t)(d + rb * 8192), 16, 0, 0);
    } else if (tail && kt == nkt - 1) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        if (rb >= r0 && rb < r1) {
          const uint32_t off = (uint32_t)((long)min(t_kr0 + rb * 16, krem - 1) * P.ldb * 2) + t_bcol;
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)off), (lds_ptr_t)(d + rb * 8192), 16, 0, 0);
        }
    } else {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        if (rb >= r0 && rb < r1) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)b_off[rb]), (lds_ptr_t)(d + rb * 8192), 16, 0, 0);
    }
  };
  // zero the k-rows [krem, 64) of both operand tiles of buffer b (waves 0-3; the DMA of that K-tile has landed for every wave)
  auto zero_tail = [&](int b) {
    if (wm == 0) {
      const int nbytes = (64 - krem) * 512;
      for (int o = tid * 16; o < nbytes; o += 256 * 16) {
        *reinterpret_cast<uint4*>(lds + b * 32768 + krem * 512 + o) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(lds + 65536 + b * 32768 + krem * 512 + o) = make_uint4(0, 0, 0, 0);
      }
      G8_WAIT_LGKM0();
    }
    G8_BARRIER();
  };

  // ---- fragment reads: inline asm (the compiler then neither counts them nor guards them against the DMA in flight: every wait
  //      below is explicit)
  const int li = lane & 31, lq = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  struct Frag {
    uint4 v;        // NT: one ds_read_b128
    uint2 lo, hi;   // TN: two ds_read_b64_tr_b16 (k = 0..3 | 4..7 of the lane's eight)
  };
  Frag fa[2][4], fbx[4], fby[4];
  // NT: operand row (wave base + li), 16-byte chunk (2c + lq) ^ key of its 128-byte LDS row:  ra[c] / rb_[c] per k-step c
  // TN: 16-lane group gq = lane >> 4 covers m = 16 (gq & 1) .. +15, k = 8 (gq >> 1) + {0..3 | 4..7}; lane i of a group passes the
  //     address of (k-row i / 4, m-block 4 (i % 4)) and receives k = 0..3 of m = its own index.  Byte address of element (k, m)
  //     of a tile: k * 512 + (((m >> 5) ^ (k & 3)) << 6) + (m & 31) * 2:  ra[j] = the wave's j-th 32-column quarter of A
  //     (j = 2 mh + i), rb_[nh] the same for B.
  uint32_t ra[4], rb_[4];
  if constexpr (!TN) {
    const int key = (li >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ra[c] = lds0 + (wm * 128 + li) * 128 + (((2 * c + lq) ^ key) << 4);
      rb_[c] = lds0 + 65536 + (wn * 64 + li) * 128 + (((2 * c + lq) ^ key) << 4);
    }
  } else {
    const int l16 = lane & 15, gq = lane >> 4;
    const int krow = 8 * (gq >> 1) + (l16 >> 2);          // (krow & 3) == l16 >> 2, also after + 4 / + 16 c
    const int mloc = 16 * (gq & 1) + 4 * (l16 & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = lds0 + krow * 512 + mloc * 2 + ((((wm * 4 + j) ^ (l16 >> 2))) << 6);
#pragma unroll
    for (int j = 0; j < 2; ++j) rb_[j] = lds0 + 65536 + krow * 512 + mloc * 2 + ((((wn * 2 + j) ^ (l16 >> 2))) << 6);
    rb_[2] = rb_[3] = 0;
  }
#define G8_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define G8_RDTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
  // A fragments of m-half MH (2 MFMA tiles x 4 k-steps) / B fragments of n-half NH (4 k-steps) out of buffer BUF
#define G8_READ_A(BUF, MH)                                                                      \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int c = 0; c < 4; ++c) { \
    if constexpr (!TN) {                                                                        \
      G8_RD128(fa[i][c].v, ra[c], (BUF) * 32768 + ((MH) * 64 + i * 32) * 128);                  \
    } else {                                                                                    \
      G8_RDTR(fa[i][c].lo, ra[(MH) * 2 + i], (BUF) * 32768 + c * 8192);                         \
      G8_RDTR(fa[i][c].hi, ra[(MH) * 2 + i], (BUF) * 32768 + c * 8192 + 2048);                  \
    }                                                                                           \
  }
#define G8_READ_B(BUF, NH, FB)                                           \
  _Pragma("unroll") for (int c = 0; c < 4; ++c) {                        \
    if constexpr (!TN) {                                                 \
      G8_RD128(FB[c].v, rb_[c], (BUF) * 32768 + (NH) * 32 * 128);        \
    } else {                                                             \
      G8_RDTR(FB[c].lo, rb_[NH], (BUF) * 32768 + c * 8192);              \
      G8_RDTR(FB[c].hi, rb_[NH], (BUF) * 32768 + c * 8192 + 2048);       \
    }                                                                    \
  }
  auto frag = [&](const Frag& f) {
    if constexpr (!TN)
      return f.v;
    else
      return make_uint4(f.lo.x, f.lo.y, f.hi.x, f.hi.y);
  };
  auto mma = [&](int mh, int nh, const Frag (&fb)[4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        // operands swapped on purpose: D rows = n, D columns = m -> a lane holds 4 consecutive n of one output row m
        Mma<bf16_t>::mma32(acc[mh * 2 + i][nh], frag(fb[c]), frag(fa[i][c]));
    __builtin_amdgcn_s_setprio(0);
  };

  // one K-tile: buffer BUF holds K-tile KT (landed and visible).
  //  !SPREAD: the DMA of K-tile KT + 1 into the other buffer is in flight; K-tile KT + 2 is staged into BUF in phases 2 (B) and 3 (A).
  //   SPREAD (experiment, not the default): two DMA pieces per phase -- B of K-tile KT + 2 goes into BUF in phases 2 and 3, A of
  //           K-tile KT + 1 into the other buffer in phases 0 and 1 (its A rows were last read in phase 2 of the K-tile before).  In
  //           flight at the phase-3 wait: the four B pieces just issued -> s_waitcnt vmcnt(4) retires everything K-tile KT + 1 needs.
#define G8_KTILE(BUF, KT)                                      \
  {                                                            \
    const bool more = (KT) + 2 < nkt;                          \
    const bool more_a = SPREAD && (KT) >= 1 && (KT) + 1 < nkt; \
    if (tail && (KT) == nkt - 1) zero_tail(BUF);               \
    /* phase 0 */                                              \
    fetch_rows((KT) + 2);                                      \
    G8_READ_A(BUF, 0)                                          \
    G8_READ_B(BUF, 0, fbx)                                     \
    if (more_a) stage_a((BUF) ^ 1, (KT) + 1, 0, 2);            \
    G8_BARRIER();                                              \
    G8_WAIT_LGKM0();                                           \
    __builtin_amdgcn_sched_barrier(0);                         \
    mma(0, 0, fbx);                                            \
    G8_BARRIER();                                              \
    /* phase 1 */                                              \
    G8_READ_B(BUF, 1, fby)                                     \
    if (more_a) stage_a((BUF) ^ 1, (KT) + 1, 2, 4);            \
    G8_WAIT_LGKM0();                                           \
    G8_BARRIER();                                              \
    mma(0, 1, fby);                                            \
    G8_BARRIER();                                              \
    /* phase 2 */                                              \
    G8_READ_A(BUF, 1)                                          \
    select_rows();                                             \
    if (more) stage_b(BUF, (KT) + 2, 0, SPREAD ? 2 : 4);       \
    G8_WAIT_LGKM0();                                           \
    G8_BARRIER();                                              \
    mma(1, 1, fby);                                            \
    G8_BARRIER();                                              \
    /* phase 3 */                                              \
    if (SPREAD) {                                              \
      if (more) {                                              \
        stage_b(BUF, (KT) + 2, 2, 4);                          \
        G8_WAIT_VM(4);                                         \
      } else {                                                 \
        G8_WAIT_VM(0);                                         \
      }                                                        \
    } else if (more) {                                         \
      stage_a(BUF, (KT) + 2);                                  \
      G8_WAIT_VM(8);                                           \
    } else {                                                   \
      G8_WAIT_VM(0);                                           \
    }                                                          \
    G8_BARRIER();                                              \
    mma(1, 0, fbx);                                            \
    G8_BARRIER();                                              \
  }

  // prologue: K-tiles 0 and 1
  fetch_rows(0);
  select_rows();
  stage_a(0, 0);
  stage_b(0, 0);
  if (nkt > 1) {
    fetch_rows(1);
    select_rows();
    stage_a(1, 1);
    stage_b(1, 1);
    G8_WAIT_VM(8);
  } else {
    G8_WAIT_VM(0);
  }
  G8_BARRIER();
  if (wm == 1) G8_BARRIER();        // M-half 1 runs one barrier behind
  for (int kt = 0; kt < nkt; kt += 2) {
    G8_KTILE(0, kt)
    if (kt + 1 < nkt) G8_KTILE(1, kt + 1)
  }
  if (wm == 0) G8_BARRIER();        // balance the barrier count

  // ---- epilogue
  int eli = li, elq = lq;
  asm volatile("" : "+v"(eli), "+v"(elq));
  const float* bias = P.bias;
  const int M_ = P.M, N_ = P.N;
  const long ldc = P.ldc;
  if (!P.out_is_f32 && (ldc % 8 == 0) && ((((uintptr_t)P.C) & 15) == 0)) {
    // bf16 output: through LDS (free now: every wave is past the last barrier, nothing reads the operand tiles any more), so that
    // a store instruction writes whole 128-byte row segments (8 rows x 128 B per instruction) instead of 16-byte pieces of 32
    // rows.  A wave stages its 128 x 64 tile in two halves of 64 rows in its own 9 KiB (row stride 144 B: 16-byte aligned
    // reads, two-way write conflicts at most); wave-private, so no barrier.
    unsigned char* ep = lds + wave * 9216;
    bf16_t* Cb = (bf16_t*)P.C;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = j * 32 + 8 * q + 4 * elq;
            const int gcol = n0 + wn * 64 + col;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
#pragma unroll
              for (int e = 0; e < 4; ++e) bv[e] = gcol + e < N_ ? bias[gcol + e] : 0.f;
            }
            uint2 pk;
            pk.x = cvt_pk_bf16(acc[half * 2 + i2][j][4 * q + 0] + bv[0], acc[half * 2 + i2][j][4 * q + 1] + bv[1]);
            pk.y = cvt_pk_bf16(acc[half * 2 + i2][j][4 * q + 2] + bv[2], acc[half * 2 + i2][j][4 * q + 3] + bv[3]);
            *reinterpret_cast<uint2*>(ep + (i2 * 32 + eli) * 144 + col * 2) = pk;
          }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row_l = r * 8 + (lane >> 3), seg = lane & 7;
        const uint4 d = *reinterpret_cast<const uint4*>(ep + row_l * 144 + seg * 16);
        const int grow = m0 + wm * 128 + half * 64 + row_l, gcol = n0 + wn * 64 + seg * 8;
        if (grow < M_) {
          bf16_t* cp = Cb + (long)(P.rows ? P.rows[grow] : grow) * ldc + gcol;
          if (gcol + 7 < N_) {
            *reinterpret_cast<uint4*>(cp) = d;
          } else {
            const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (gcol + e < N_) cp[e].v = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
          }
        }
      }
    }
    return;
  }
  const bool n_vec_ok = (ldc % 4 == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + wm * 128 + i * 32 + eli;
    if (row >= M_) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn * 64 + j * 32 + 8 * q + 4 * elq;
        if (col >= N_) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] + ((bias && col + e < N_) ? bias[col + e] : 0.f);
        const long off = (long)((!TN && P.rows) ? P.rows[row] : row) * ldc + col;
        if (P.out_is_f32) {
          float* cp = (float*)P.C + off;
          if (col + 3 < N_ && n_vec_ok) {
            *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < N_) cp[e] = v[e];
          }
        } else {
          bf16_t* cp = (bf16_t*)P.C + off;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < N_) stf(cp + e, v[e]);
        }
      }
    }
  }
}


// MODE 0: every problem NT, 1: every problem TN, 2: mixed.  Workgroup -> (problem, tile): problems own consecutive id ranges (the
// host orders them by decreasing K, so the long tiles are dispatched first); inside a problem's range the ids that the dispatcher
// places on one XCD (id % 8) get a contiguous run of its tiles (neighbouring tiles share operand panels in that XCD's L2).
template <int MODE, bool SPREAD>
__global__ void __launch_bounds__(512, 2) k_gemm8(G8Args g) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[131072];
  const int id = blockIdx.x;
  int pi = 0, first = 0;
#pragma unroll
  for (int i = 0; i < G8_MAX_GROUPS - 1; ++i)
    if (i + 1 < g.n_problems && id >= g.p[i].tile_end) {
      pi = i + 1;
      first = g.p[i].tile_end;
    }
  const G8Problem& P = g.p[pi];
  const int count = P.tile_end - first, local = id - first, xcd = id & 7;
  int base = 0;
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const int l0 = (y - first) & 7;                     // first local id of this problem that lands on XCD y
    const int n_y = l0 < count ? (count - l0 - 1) / 8 + 1 : 0;
    if (y < xcd) base += n_y;
  }
  const int t_id = base + (local - ((xcd - first) & 7)) / 8;
  if (MODE == 0)
    g8_tile<false, SPREAD>(P, t_id, lds);
  else if (MODE == 1)
    g8_tile<true, SPREAD>(P, t_id, lds);
  else if (P.tn)
    g8_tile<true, SPREAD>(P, t_id, lds);
  else
    g8_tile<false, SPREAD>(P, t_id, lds);
}

// The SPREAD staging schedule (two DMA pieces per phase) stays in the source as a measured-and-rejected experiment: no gain on the NT
// shapes of the step (0.289 vs 0.290 ms), slower on the grouped TN launch (0.555 vs 0.589 ms), profiles/r03d_gemm8_ab.txt.  It is not
// instantiated (round 6: the process-wide ds2_gemm8_set_variant hook that selected it is gone from the ABI).
template <int MODE>
void g8_launch(int total, const G8Args& g, hipStream_t st) {
  hipLaunchKernelGGL((k_gemm8<MODE, false>), dim3(total), dim3(512), 0, st, g);
}

}  // namespace


// The `_rows` entries take a row list (device, int32, strictly inside [0, n_phys)) and visit only the listed rows of the row dimension
// shared by the activations of a padded [T' x batch] sequence matrix -- the frames t < length of every clip (model.py:96,100: the
// reference packs them with pack_padded_sequence; here the matrices stay in place and the products skip the padding):
//   NT: C[rows[r]][:] = A[rows[r]][:] * B^T (+ bias) for r < n_rows; other rows of C are NOT written.
//   TN: C = sum over k < n_rows of At[rows[k]][m] * Bt[rows[k]][n].
// rows == null: every row (n_rows = n_phys).  n_phys bounds the operands (n_phys * ld < 2^31 elements).
static int g8_nt(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb, long ldc, int out_f32,
                 const int* rows, int n_phys, hipStream_t st) {
  DS2_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && n_phys >= M, DS2_ERR_ARG);
  DS2_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0, DS2_ERR_ALIGN);
  DS2_REQUIRE((long)n_phys * lda < (1L << 31) && (long)N * ldb < (1L << 31), DS2_ERR_ARG);
  G8Args g{};
  g.n_problems = 1;
  const int tm = ds2_cdiv(M, 256), tn = ds2_cdiv(N, 256);
  g.p[0] = G8Problem{(const bf16_t*)A, nullptr, (const bf16_t*)B, C, bias, M, N, 0, lda, 0, ldb, ldc, tn, tm * tn, out_f32, K, 0, rows};
  g8_launch<0>(tm * tn, g, st);
  DS2_CHECK_LAUNCH();
  return 0;
}

static int g8_tn_fill(G8Args& g, int& total, int n, const void* const* At, const void* const* At2, const int* m_split, const void* const* Bt,
                      void* const* C, const int* M, const int* N, const long* lda, const long* lda2, const long* ldb, const long* ldc, int K,
                      const int* rows, int n_phys) {
  DS2_REQUIRE(K > 0 && n_phys >= K, DS2_ERR_ARG);
  for (int i = 0; i < n; ++i) {
    DS2_REQUIRE(M[i] > 0 && N[i] > 0 && M[i] % 8 == 0 && N[i] % 8 == 0 && lda[i] % 8 == 0 && ldb[i] % 8 == 0, DS2_ERR_ARG);
    DS2_REQUIRE((((uintptr_t)At[i]) & 15) == 0 && (((uintptr_t)Bt[i]) & 15) == 0, DS2_ERR_ALIGN);
    DS2_REQUIRE((long)n_phys * lda[i] < (1L << 31) && (long)n_phys * ldb[i] < (1L << 31), DS2_ERR_ARG);
    const void* a2 = At2 ? At2[i] : nullptr;
    const int ms = a2 ? m_split[i] : 0;
    DS2_REQUIRE(a2 == nullptr || (ms > 0 && ms < M[i] && ms % 256 == 0 && (((uintptr_t)a2) & 15) == 0 && lda2 && lda2[i] % 8 == 0 &&
                                  (long)n_phys * lda2[i] < (1L << 31)), DS2_ERR_ARG);
    const int tm = ds2_cdiv(M[i], 256), tn = ds2_cdiv(N[i], 256);
    total += tm * tn;
    g.p[i] = G8Problem{(const bf16_t*)At[i], (const bf16_t*)a2, (const bf16_t*)Bt[i], C[i], nullptr, M[i], N[i], ms, lda[i], a2 ? lda2[i] : 0, ldb[i], ldc[i], tn, total, 1, K, 1, rows};
  }
  return 0;
}

extern "C" {

// C (f32 if out_f32 else bf16) [M][ldc] = A[M][lda] * B[N][ldb]^T (+ bias[N]) on the 256x256 phase-split kernel.
// K % 64 == 0, lda/ldb % 8 == 0, 16-byte aligned operands of < 2^31 elements.
int ds2_gemm8_nt(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb, long ldc, int out_f32,
                 ds2_stream_t st) {
  return g8_nt(A, B, C, bias, M, N, K, lda, ldb, ldc, out_f32, nullptr, M, (hipStream_t)st);
}
int ds2_gemm8_nt_rows(const void* A, const void* B, void* C, const float* bias, int n_phys, int N, int K, long lda, long ldb, long ldc,
                      int out_f32, const int* rows, int n_rows, ds2_stream_t st) {
  return g8_nt(A, B, C, bias, rows ? n_rows : n_phys, N, K, lda, ldb, ldc, out_f32, rows, n_phys, (hipStream_t)st);
}

// Grouped TN products, one launch:  for every problem i   C_i[M_i][N_i] f32 = sum_k At_i[k][m] * Bt_i[k][n],  k < K
// (At_i [K][lda_i], Bt_i [K][ldb_i]: the contraction index is the ROW of both operands).  A problem's A operand may be two
// column blocks: output rows >= m_split_i take their columns from A2_i (leading dimension lda2_i), m_split_i a multiple of 256.
// Any K (a ragged last K-tile is zero-filled on chip; no row past K - 1 is read), M_i, N_i % 8 == 0, 16-byte aligned rows.
int ds2_gemm8_tn_grouped_rows(int n_problems, const void* const* At, const void* const* At2, const int* m_split, const void* const* Bt,
                              void* const* C, const int* M, const int* N, const long* lda, const long* lda2, const long* ldb, const long* ldc,
                              int n_phys, const int* rows, int n_rows, ds2_stream_t st) {
  DS2_REQUIRE(n_problems >= 1 && n_problems <= G8_MAX_GROUPS, DS2_ERR_ARG);
  G8Args g{};
  g.n_problems = n_problems;
  int total = 0;
  const int rc = g8_tn_fill(g, total, n_problems, At, At2, m_split, Bt, C, M, N, lda, lda2, ldb, ldc, rows ? n_rows : n_phys, rows, n_phys);
  if (rc) return rc;
  g8_launch<1>(total, g, (hipStream_t)st);
  DS2_CHECK_LAUNCH();
  return 0;
}
int ds2_gemm8_tn_grouped(int n_problems, const void* const* At, const void* const* At2, const int* m_split, const void* const* Bt, void* const* C,
                         const int* M, const int* N, const long* lda, const long* lda2, const long* ldb, const long* ldc, int K, ds2_stream_t st) {
  return ds2_gemm8_tn_grouped_rows(n_problems, At, At2, m_split, Bt, C, M, N, lda, lda2, ldb, ldc, K, nullptr, K, st);
}

// The weight gradients of a recurrent layer AND its dX product in one launch: the TN problems of ds2_gemm8_tn_grouped (contraction
// over K_tn rows) followed by ONE NT problem (dX [M_nt][ldc_nt] bf16 = A_nt[M_nt][K_nt] * B_nt[N_nt][K_nt]^T).  Alone, the weight
// gradients leave a quarter of the CUs idle (192 tiles of 376 K-tiles on cfg3) and the dX product half of its second round (376
// tiles of 96 K-tiles); in one grid the long tiles start first and the short ones fill in behind them.  With a row list
// (n_phys = K_tn = M_nt physical rows) the TN problems contract over the listed rows and the NT problem computes the listed rows.
int ds2_gemm8_wgrad_dx_rows(int n_tn, const void* const* At, const void* const* At2, const int* m_split, const void* const* Bt, void* const* C,
                            const int* M, const int* N, const long* lda, const long* lda2, const long* ldb, const long* ldc, int K_tn,
                            const void* A_nt, const void* B_nt, void* C_nt, int M_nt, int N_nt, int K_nt, long lda_nt, long ldb_nt, long ldc_nt,
                            const int* rows, int n_rows, ds2_stream_t st) {
  DS2_REQUIRE(n_tn >= 1 && n_tn < G8_MAX_GROUPS && K_tn > 0, DS2_ERR_ARG);
  DS2_REQUIRE(rows == nullptr || (K_tn == M_nt && n_rows > 0 && n_rows <= K_tn), DS2_ERR_ARG);
  DS2_REQUIRE(M_nt > 0 && N_nt > 0 && K_nt > 0 && K_nt % 64 == 0 && lda_nt % 8 == 0 && ldb_nt % 8 == 0, DS2_ERR_ARG);
  DS2_REQUIRE((((uintptr_t)A_nt) & 15) == 0 && (((uintptr_t)B_nt) & 15) == 0, DS2_ERR_ALIGN);
  DS2_REQUIRE((long)M_nt * lda_nt < (1L << 31) && (long)N_nt * ldb_nt < (1L << 31), DS2_ERR_ARG);
  G8Args g{};
  g.n_problems = n_tn + 1;
  int total = 0;
  const int rc = g8_tn_fill(g, total, n_tn, At, At2, m_split, Bt, C, M, N, lda, lda2, ldb, ldc, rows ? n_rows : K_tn, rows, K_tn);
  if (rc) return rc;
  const int m_eff = rows ? n_rows : M_nt;
  const int tm = ds2_cdiv(m_eff, 256), tn = ds2_cdiv(N_nt, 256);
  total += tm * tn;
  g.p[n_tn] = G8Problem{(const bf16_t*)A_nt, nullptr, (const bf16_t*)B_nt, C_nt, nullptr, m_eff, N_nt, 0, lda_nt, 0, ldb_nt, ldc_nt, tn, total, 0, K_nt, 0, rows};
  g8_launch<2>(total, g, (hipStream_t)st);
  DS2_CHECK_LAUNCH();
  return 0;
}
int ds2_gemm8_wgrad_dx(int n_tn, const void* const* At, const void* const* At2, const int* m_split, const void* const* Bt, void* const* C,
                       const int* M, const int* N, const long* lda, const long* lda2, const long* ldb, const long* ldc, int K_tn,
                       const void* A_nt, const void* B_nt, void* C_nt, int M_nt, int N_nt, int K_nt, long lda_nt, long ldb_nt, long ldc_nt,
                       ds2_stream_t st) {
  return ds2_gemm8_wgrad_dx_rows(n_tn, At, At2, m_split, Bt, C, M, N, lda, lda2, ldb, ldc, K_tn, A_nt, B_nt, C_nt, M_nt, N_nt, K_nt, lda_nt,
                                 ldb_nt, ldc_nt, nullptr, 0, st);
}

}  // extern "C"
