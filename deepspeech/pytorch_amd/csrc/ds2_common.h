// ds2hip -- MI355X (gfx950 / CDNA4) kernels for the DeepSpeech2 train-step hot path.
// Common device helpers: storage types, bf16 conversion, 16-byte fragment MFMA wrappers, wave reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/ds2hip.h"

#define DS2_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 ds2_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ds2_f32x16;
typedef __attribute__((ext_vector_type(4))) float ds2_f32x4;

// Activation storage type tags.  bf16 is stored as raw uint16 (no header API dependence).
struct bf16_t {
  uint16_t v;
};

__device__ __host__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  union {
    uint32_t u;
    float f;
  } c;
  c.u = ((uint32_t)b) << 16;
  return c.f;
}

// round-to-nearest-even, NaN preserved (quiet)
__device__ __host__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  union {
    uint32_t u;
    float f;
  } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// two fp32 -> packed bf16x2 (lo in bits 15:0) in ONE instruction: gfx950's v_cvt_pk_bf16_f32 (round-to-nearest-even,
// NaN stays NaN).  The portable bit-twiddling version above costs ~10 instructions and an exec-mask branch per value
// (the NaN test), which dominated the epilogues of latency-bound kernels.
// (Round 5, measured and not adopted: `__builtin_convertvector(float2 -> bf16x2)` lowers to the same single instruction and lets the
// compiler schedule and if-convert it, but the sweeps' forward kernels came out 1.5-4 % slower with it (profiles/r05h_ab_sweep_timing.txt).
// With the asm form, never write `cond ? pack(a, b) : pack(b, a)`: the statement cannot be if-converted and becomes a divergent branch
// diamond -- select the operands, then pack once.)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

template <typename T>
struct Store;
template <>
struct Store<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <>
struct Store<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_bits_to_f32(p->v); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { p->v = (uint16_t)cvt_pk_bf16(v, 0.f); }
};

template <typename T>
__device__ __forceinline__ float ldf(const T* p) {
  return Store<T>::ld(p);
}
template <typename T>
__device__ __forceinline__ void stf(T* p, float v) {
  Store<T>::st(p, v);
}

// ---- 16-byte vector access: VEC<T>::N elements per 16 bytes ---------------------------------------------------
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <>
struct Vec16<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&o)[8]) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&o)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = cvt_pk_bf16(o[2 * i], o[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// ---- MFMA on 16-byte operand fragments --------------------------------------------------------------------
// One "chunk" = 16 bytes of K per lane for both operands.
//   bf16: 8 k-values per lane; 32x32x16 covers K=16 per instruction (lane-group q=lane>>5 holds k=8q..8q+7),
//         16x16x32 covers K=32 (lane-group q=lane>>4 holds k=8q..8q+7).
//   f32 : 4 k-values per lane; issued as 4 instructions of 32x32x2 (or 16x16x4); instruction c uses element c of
//         both fragments, i.e. the k-pairing is (q, c) on both sides -- consistent, so the sum over the chunk is exact
//         regardless of the order.  K covered per chunk: 8 (32x32) / 16 (16x16).
template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  static constexpr int K32 = 16;  // K per chunk with 32x32 tiles
  static constexpr int K16 = 32;  // K per chunk with 16x16 tiles
  static __device__ __forceinline__ void mma32(ds2_f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ds2_bf16x8, a), __builtin_bit_cast(ds2_bf16x8, b), acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void mma16(ds2_f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ds2_bf16x8, a), __builtin_bit_cast(ds2_bf16x8, b), acc, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  static constexpr int K32 = 8;
  static constexpr int K16 = 16;
  static __device__ __forceinline__ void mma32(ds2_f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void mma16(ds2_f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};
// C/D layouts (dtype independent, cdna guide section 3):
//   32x32: acc[r] of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
//   16x16: acc[r] of lane l is D[row = 4*(l>>4) + r][col = l&15]
// rows index the A operand's lane index (l&31 / l&15), cols the B operand's.
__device__ __forceinline__ int mma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int mma16_row(int r, int lane) { return 4 * (lane >> 4) + r; }

// ---- wave / block reductions ---------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); accurate to ~1e-7 relative with full-precision expf
  float e = expf(2.0f * x);
  return 1.0f - 2.0f / (e + 1.0f);
}
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

#define DS2_CHECK_LAUNCH()                             \
  do {                                                 \
    hipError_t e__ = hipGetLastError();                \
    if (e__ != hipSuccess) return (int)e__;            \
  } while (0)

// compute units of the CURRENT device (cached per device; 256 when the query fails)
static inline int ds2_cu_count() {
  static int n[64];
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (n[dev] == 0) n[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : 256;
  return n[dev];
}

#define DS2_REQUIRE(cond, code) \
  do {                          \
    if (!(cond)) return (code); \
  } while (0)

static inline int ds2_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
__device__ __forceinline__ int ds2_cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// Per-DEVICE one-time set-up (kernel attributes are per device, a process may drive several): returns true the first time
// it is called with `flags` on the current device.  A benign race (two threads both see "first") only repeats an idempotent call.
#define DS2_MAX_DEVICES 64
static inline bool ds2_first_use_on_device(bool* flags) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DS2_MAX_DEVICES) return true;
  if (flags[dev]) return false;
  flags[dev] = true;
  return true;
}
