// EXPERIMENT (round 6, tools/probe_gemm9.py): is a 128 x 128 wave tile worth a new GEMM kernel?
// NT bf16 GEMM, 256 x 256 tile per workgroup of FOUR waves (2 x 2), one wave per SIMD, v_mfma_f32_32x32x16_bf16 with 256 accumulator
// registers per lane, K-tile 64, two 64 KiB LDS stages filled by global_load_lds_dwordx4 (the layout and swizzle of ds2_gemm8.hip).
// Per K-tile and wave: 32 fragment reads (ds_read_b128) feed 64 MFMAs -- 128 KB of LDS reads per workgroup against ds2_gemm8's
// 192 KB -- issued one k16-step ahead and interleaved with the MFMAs by source order (sched_barrier between mini-groups).
// Not part of libds2hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct G9Args {
  const uint16_t* A;   // [M][lda] bf16
  const uint16_t* B;   // [N][ldb] bf16
  uint16_t* C;         // [M][ldc] bf16
  const float* bias;   // [N] or null
  int M, N, K;
  long lda, ldb, ldc;
  int tiles_n, tiles;
};

#define G9_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define G9_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define G9_SB() __builtin_amdgcn_sched_barrier(0)
#define G9_BARRIER()                  \
  do {                                \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();     \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)
#define G9_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__global__ void __launch_bounds__(256, 1) k_gemm9_nt(G9Args g) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[131072];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile id: the ids the dispatcher places on one XCD (id % 8) get a contiguous run of tiles
  const int orig = blockIdx.x, xcd = orig & 7, loc = orig >> 3;
  const int q8 = g.tiles >> 3, r8 = g.tiles & 7;
  const int t_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int m0 = (t_id / g.tiles_n) * 256, n0 = (t_id % g.tiles_n) * 256;
  const int nkt = g.K / 64;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA identity: one wave instruction moves 8 tile rows x 128 B; wave w owns rows w*8 .. w*8+7 of every 32-row band (8 bands)
  uint32_t a_off[8], b_off[8];
  {
    const int rloc = wave * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rloc >> 1) & 7);
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
      const int ar = min(m0 + rb * 32 + rloc, g.M - 1), br = min(n0 + rb * 32 + rloc, g.N - 1);
      a_off[rb] = (uint32_t)(((long)ar * g.lda + chunk * 8) * 2);
      b_off[rb] = (uint32_t)(((long)br * g.ldb + chunk * 8) * 2);
    }
  }
  const unsigned char* a_base = (const unsigned char*)g.A;
  const unsigned char* b_base = (const unsigned char*)g.B;
  auto stage_a = [&](int b, int kt, int r0, int r1) {
    const unsigned char* src = a_base + (long)kt * 128;
    unsigned char* d = lds + b * 32768 + wave * 1024;
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
      if (rb >= r0 && rb < r1) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)a_off[rb]), (lds_ptr_t)(d + rb * 4096), 16, 0, 0);
  };
  auto stage_b = [&](int b, int kt, int r0, int r1) {
    const unsigned char* src = b_base + (long)kt * 128;
    unsigned char* d = lds + 65536 + b * 32768 + wave * 1024;
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
      if (rb >= r0 && rb < r1) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)b_off[rb]), (lds_ptr_t)(d + rb * 4096), 16, 0, 0);
  };

  // fragment read addresses: operand row (wave base + li), 16-byte chunk (2c + lq) ^ key of its 128-byte LDS row
  const int li = lane & 31, lq = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  uint32_t ra[4], rbk[4];
  {
    const int key = (li >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ra[c] = lds0 + (wm * 128 + li) * 128 + (((2 * c + lq) ^ key) << 4);
      rbk[c] = lds0 + 65536 + (wn * 128 + li) * 128 + (((2 * c + lq) ^ key) << 4);
    }
  }
  uint4 fa[2][4], fb[2][4];
#define G9_READ_ONE(S, BUF, C, X)                                                   \
  do {                                                                              \
    if ((X) < 4) G9_RD128(fa[S][(X)&3], ra[C], (BUF) * 32768 + ((X)&3) * 4096);     \
    else G9_RD128(fb[S][(X)&3], rbk[C], (BUF) * 32768 + ((X)&3) * 4096);            \
  } while (0)
#define G9_MMA(S, I, J)                                                                                                            \
  acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[S][J]), __builtin_bit_cast(bf16x8, fa[S][I]), \
                                                      acc[I][J], 0, 0, 0)
  // 16 MFMAs of fragment set S with the 8 reads of (set S ^ 1, buffer RBUF, k16-step RC) between them (DOREAD: 0 / 1)
#ifndef G9_FRONT
#define G9_FRONT 8      // reads issued one per MFMA from the start of the step (8: the first eight MFMAs carry them all)
#endif
#define G9_STEP(S, DOREAD, RBUF, RC)                              \
  do {                                                            \
    _Pragma("unroll") for (int x_ = 0; x_ < 16; ++x_) {           \
      if (DOREAD && x_ < 8) G9_READ_ONE((S) ^ 1, RBUF, RC, x_);   \
      G9_SB();                                                    \
      G9_MMA(S, x_ >> 2, x_ & 3);                                 \
      G9_SB();                                                    \
    }                                                             \
  } while (0)

  // ---- prologue
  stage_a(0, 0, 0, 8);
  stage_b(0, 0, 0, 8);
  if (nkt > 1) {
    stage_a(1, 1, 0, 8);
    stage_b(1, 1, 0, 8);
    G9_WAIT_VM(16);
  } else {
    G9_WAIT_VM(0);
  }
  G9_BARRIER();
#pragma unroll
  for (int x = 0; x < 8; ++x) G9_READ_ONE(0, 0, 0, x);
  G9_WAIT_LGKM0();
  G9_SB();

#define G9_KTILE(BUF, KT)                                                                       \
  {                                                                                             \
    const bool more2 = (KT) + 2 < nkt, more1 = (KT) + 1 < nkt;                                  \
    G9_STEP(0, 1, BUF, 1);                                                                      \
    G9_WAIT_LGKM0();                                                                            \
    G9_SB();                                                                                    \
    G9_STEP(1, 1, BUF, 2);                                                                      \
    G9_WAIT_LGKM0();                                                                            \
    G9_SB();                                                                                    \
    G9_STEP(0, 1, BUF, 3);                                                                      \
    G9_WAIT_LGKM0(); /* every read of BUF has returned */                                       \
    G9_BARRIER();    /* ... on every wave: BUF may be overwritten */                            \
    /* step 3: set 1; the DMA of K-tile KT + 2 into BUF between the first MFMAs */              \
    _Pragma("unroll") for (int x_ = 0; x_ < 4; ++x_) {                                          \
      G9_MMA(1, x_ >> 2, x_ & 3);                                                               \
      G9_SB();                                                                                  \
      if (more2) {                                                                              \
        stage_a(BUF, (KT) + 2, 2 * x_, 2 * x_ + 2);                                             \
        stage_b(BUF, (KT) + 2, 2 * x_, 2 * x_ + 2);                                             \
      }                                                                                         \
      G9_SB();                                                                                  \
    }                                                                                           \
    if (more1) {                                                                                \
      if (more2) G9_WAIT_VM(16); else G9_WAIT_VM(0);   /* K-tile KT + 1 has landed (this wave's pieces) */ \
      G9_BARRIER();                                    /* ... every wave's */                   \
    }                                                                                           \
    /* (the MFMAs stay outside every branch: a join of two paths that both update 256 accumulator registers costs copies) */ \
    _Pragma("unroll") for (int x_ = 4; x_ < 12; ++x_) {                                         \
      if (more1) G9_READ_ONE(0, (BUF) ^ 1, 0, x_ - 4);                                          \
      G9_SB();                                                                                  \
      G9_MMA(1, x_ >> 2, x_ & 3);                                                               \
      G9_SB();                                                                                  \
    }                                                                                           \
    _Pragma("unroll") for (int x_ = 12; x_ < 16; ++x_) G9_MMA(1, x_ >> 2, x_ & 3);              \
    G9_WAIT_LGKM0();                                                                            \
    G9_SB();                                                                                    \
  }
  // (probe: an even number of K-tiles, so that the loop body is straight-line code over both buffers)
  for (int kt = 0; kt < nkt; kt += 2) {
    G9_KTILE(0, kt)
    G9_KTILE(1, kt + 1)
  }

  // ---- epilogue (first version: straight from the registers; D rows = n, D columns = m: a lane holds 4 consecutive n of row m)
  const float* bias = g.bias;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + wm * 128 + i * 32 + li;
    if (row >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn * 128 + j * 32 + 8 * q + 4 * lq;
        if (col + 3 >= g.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] + (bias ? bias[col + e] : 0.f);
        uint2 pk;
        pk.x = pk_bf16(v[0], v[1]);
        pk.y = pk_bf16(v[2], v[3]);
        *reinterpret_cast<uint2*>(g.C + (long)row * g.ldc + col) = pk;
      }
  }
}

__global__ void __launch_bounds__(256, 1) k_gemm9r_nt(G9Args g) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[131072];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile id: the ids the dispatcher places on one XCD (id % 8) get a contiguous run of tiles
  const int orig = blockIdx.x, xcd = orig & 7, loc = orig >> 3;
  const int q8 = g.tiles >> 3, r8 = g.tiles & 7;
  const int t_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int m0 = (t_id / g.tiles_n) * 256, n0 = (t_id % g.tiles_n) * 256;
  const int nkt = g.K / 64;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA identity: one wave instruction moves 8 tile rows x 128 B; wave w owns rows w*8 .. w*8+7 of every 32-row band (8 bands)
  uint32_t a_off[8], b_off[8];
  {
    const int rloc = wave * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rloc >> 1) & 7);
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
      const int ar = min(m0 + rb * 32 + rloc, g.M - 1), br = min(n0 + rb * 32 + rloc, g.N - 1);
      a_off[rb] = (uint32_t)(((long)ar * g.lda + chunk * 8) * 2);
      b_off[rb] = (uint32_t)(((long)br * g.ldb + chunk * 8) * 2);
    }
  }
  const unsigned char* a_base = (const unsigned char*)g.A;
  const unsigned char* b_base = (const unsigned char*)g.B;
  // register staging: the wave's 8 + 8 pieces of a K-tile (8 tile rows x 128 B each) travel global -> VGPR -> ds_write_b128, in the
  // LDS image the DMA variant produces (piece rb of a 32-row band at band * 4096 + wave * 1024 + lane * 16)
  u32x4 stg[16];
  const uint32_t lds0w = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const uint32_t wa = lds0w + wave * 1024 + lane * 16, wb = wa + 65536;
#define G9R_LOAD(KT, X)                                                                                                   \
  do {                                                                                                                    \
    if ((X) < 8) stg[X] = *reinterpret_cast<const u32x4*>(a_base + (long)(KT) * 128 + (size_t)a_off[(X)&7]);              \
    else stg[X] = *reinterpret_cast<const u32x4*>(b_base + (long)(KT) * 128 + (size_t)b_off[(X)&7]);                      \
  } while (0)
#define G9R_WRITE(BUF, X)                                                                                                 \
  do {                                                                                                                    \
    if ((X) < 8) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wa), "v"(stg[X]), "i"((BUF) * 32768 + ((X)&7) * 4096) : "memory"); \
    else asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wb), "v"(stg[X]), "i"((BUF) * 32768 + ((X)&7) * 4096) : "memory");         \
  } while (0)

  // fragment read addresses: operand row (wave base + li), 16-byte chunk (2c + lq) ^ key of its 128-byte LDS row
  const int li = lane & 31, lq = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  uint32_t ra[4], rbk[4];
  {
    const int key = (li >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ra[c] = lds0 + (wm * 128 + li) * 128 + (((2 * c + lq) ^ key) << 4);
      rbk[c] = lds0 + 65536 + (wn * 128 + li) * 128 + (((2 * c + lq) ^ key) << 4);
    }
  }
  uint4 fa[2][4], fb[2][4];
#define G9_READ_ONE(S, BUF, C, X)                                                   \
  do {                                                                              \
    if ((X) < 4) G9_RD128(fa[S][(X)&3], ra[C], (BUF) * 32768 + ((X)&3) * 4096);     \
    else G9_RD128(fb[S][(X)&3], rbk[C], (BUF) * 32768 + ((X)&3) * 4096);            \
  } while (0)
#define G9_MMA(S, I, J)                                                                                                            \
  acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[S][J]), __builtin_bit_cast(bf16x8, fa[S][I]), \
                                                      acc[I][J], 0, 0, 0)
  // 16 MFMAs of fragment set S with the 8 reads of (set S ^ 1, buffer RBUF, k16-step RC) between them (DOREAD: 0 / 1)
#ifndef G9_FRONT
#define G9_FRONT 8      // reads issued one per MFMA from the start of the step (8: the first eight MFMAs carry them all)
#endif
#define G9_STEP(S, DOREAD, RBUF, RC)                              \
  do {                                                            \
    _Pragma("unroll") for (int x_ = 0; x_ < 16; ++x_) {           \
      if (DOREAD && x_ < 8) G9_READ_ONE((S) ^ 1, RBUF, RC, x_);   \
      G9_SB();                                                    \
      G9_MMA(S, x_ >> 2, x_ & 3);                                 \
      G9_SB();                                                    \
    }                                                             \
  } while (0)

  // ---- prologue: K-tile 0 into LDS buffer 0, K-tile 1 staged in registers
#pragma unroll
  for (int x = 0; x < 16; ++x) G9R_LOAD(0, x);
#pragma unroll
  for (int x = 0; x < 16; ++x) G9R_WRITE(0, x);
  if (nkt > 1) {
#pragma unroll
    for (int x = 0; x < 16; ++x) G9R_LOAD(1, x);
  }
  G9_WAIT_LGKM0();
  G9_BARRIER();
#pragma unroll
  for (int x = 0; x < 8; ++x) G9_READ_ONE(0, 0, 0, x);
  G9_WAIT_LGKM0();
  G9_SB();

  // One K-tile: 64 MFMAs in four steps of 16.  Steps 0-2 read the rest of BUF's fragments; step 2 also writes the staged K-tile
  // KT + 1 into the other buffer (free since the barrier of the K-tile before); ONE barrier; step 3 reads K-tile KT + 1's first
  // fragments and issues the global loads of K-tile KT + 2 into the staging registers (consumed a K-tile later, in step 2).
#define G9R_KTILE(BUF, KT)                                                                      \
  {                                                                                             \
    const bool more2 = (KT) + 2 < nkt, more1 = (KT) + 1 < nkt;                                  \
    G9_STEP(0, 1, BUF, 1);                                                                      \
    G9_WAIT_LGKM0();                                                                            \
    G9_SB();                                                                                    \
    G9_STEP(1, 1, BUF, 2);                                                                      \
    G9_WAIT_LGKM0();                                                                            \
    G9_SB();                                                                                    \
    _Pragma("unroll") for (int x_ = 0; x_ < 16; ++x_) {                                         \
      if (x_ < 8) G9_READ_ONE(1, BUF, 3, x_);                                                   \
      if (more1) G9R_WRITE((BUF) ^ 1, x_);                                                      \
      G9_SB();                                                                                  \
      G9_MMA(0, x_ >> 2, x_ & 3);                                                               \
      G9_SB();                                                                                  \
    }                                                                                           \
    G9_WAIT_LGKM0();                                                                            \
    G9_BARRIER();                                                                               \
    _Pragma("unroll") for (int x_ = 0; x_ < 16; ++x_) {                                         \
      if (more1 && x_ < 8) G9_READ_ONE(0, (BUF) ^ 1, 0, x_);                                    \
      if (more2) G9R_LOAD((KT) + 2, x_);                                                        \
      G9_SB();                                                                                  \
      G9_MMA(1, x_ >> 2, x_ & 3);                                                               \
      G9_SB();                                                                                  \
    }                                                                                           \
    G9_WAIT_LGKM0();                                                                            \
    G9_SB();                                                                                    \
  }
  for (int kt = 0; kt < nkt; kt += 2) {
    G9R_KTILE(0, kt)
    G9R_KTILE(1, kt + 1)
  }

  // ---- epilogue (first version: straight from the registers; D rows = n, D columns = m: a lane holds 4 consecutive n of row m)
  const float* bias = g.bias;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + wm * 128 + i * 32 + li;
    if (row >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn * 128 + j * 32 + 8 * q + 4 * lq;
        if (col + 3 >= g.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] + (bias ? bias[col + e] : 0.f);
        uint2 pk;
        pk.x = pk_bf16(v[0], v[1]);
        pk.y = pk_bf16(v[2], v[3]);
        *reinterpret_cast<uint2*>(g.C + (long)row * g.ldc + col) = pk;
      }
  }
}

extern "C" int gemm9_nt(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb, long ldc, void* st) {
  if (K % 128 != 0 || N % 4 != 0) return 1;
  G9Args g;
  g.A = (const uint16_t*)A; g.B = (const uint16_t*)B; g.C = (uint16_t*)C; g.bias = bias; g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.tiles_n = (N + 255) / 256;
  g.tiles = g.tiles_n * ((M + 255) / 256);
  hipLaunchKernelGGL(k_gemm9_nt, dim3(g.tiles), dim3(256), 0, (hipStream_t)st, g);
  return (int)hipGetLastError();
}

extern "C" int gemm9r_nt(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb, long ldc, void* st) {
  if (K % 128 != 0 || N % 4 != 0) return 1;
  G9Args g;
  g.A = (const uint16_t*)A; g.B = (const uint16_t*)B; g.C = (uint16_t*)C; g.bias = bias; g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.tiles_n = (N + 255) / 256;
  g.tiles = g.tiles_n * ((M + 255) / 256);
  hipLaunchKernelGGL(k_gemm9r_nt, dim3(g.tiles), dim3(256), 0, (hipStream_t)st, g);
  return (int)hipGetLastError();
}
