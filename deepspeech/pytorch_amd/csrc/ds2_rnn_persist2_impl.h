// Generalised persistent recurrent sweeps (forward and BPTT) of BatchRNN (reference model.py:94-102): the shapes the tuned
// H = 1024 / bf16 kernels of ds2_rnn_persist_impl.h do not cover -- hidden sizes 800 / 1024 / 1280 (BASELINE.json configs 2
// and 5), fp32 storage (the 1e-3 parity mode) and up to 64 samples per group.
//
// Same idea: ONE launch per layer and sweep, W_hh resident in registers as MFMA B fragments, h exchanged between the
// workgroups of a group as 8-byte {tag = step, payload} granules (the data is the flag; MI355X guide section 6 G16, recipe R2).
// What is generalised:
//   * every workgroup owns U = 16 hidden units of every gate, so a group is P = H/16 workgroups (50 / 64 / 80) and the chip
//     holds NG = floor(256/P) groups (5 / 4 / 3); a group owns one direction and a slice of the minibatch (up to 16*MT
//     samples, MT in {1,2,4} M-tiles of the 16x16 MFMA);
//   * storage type T: bf16 (granule = 2 x bf16, v_mfma_f32_16x16x32_bf16) or fp32 (granule = 1 x f32, v_mfma_f32_16x16x4_f32,
//     full-precision gate math) -- the exchange layout is the MFMA A-fragment order for both, so a gather instruction of a
//     wave always reads 1 KiB of contiguous memory;
//   * K (= H forward, G*H BPTT) is split over the 4 waves in whole k-steps; when the k-steps do not divide by 4 (H = 800) the
//     last wave simply owns fewer (wave-uniform predicate, zero weights in the unused fragment registers);
//   * groups span XCDs (P > 32), so publishes are always the write-through (sc1) form; results never depend on placement.
// Gathers are NOT speculative here (tags are checked before the MFMAs; the loads of the next chunk are in flight under the
// MFMAs of the current one): with MT > 1 a second set of accumulators would not fit beside the resident weights.
// Safety protocol as in the tuned kernels: grid <= CU count, bounded spins, *err / per-launch word, NaN poisoning.
#pragma once
#include "ds2_rnn_persist_impl.h"

namespace ds2q {
using namespace ds2p;

struct QArgs {
  int N, Tp, D, gpd, NG;    // gpd = groups per direction, NG = D * gpd groups in the grid
  const int* lens;          // [N]
  const void* W;            // fwd: W_hh [D][G*H][H]        bwd: W_hh^T [D][H][G*H]            (type T)
  const float* bhh;         // [D][G*H]
  const void* GI;           // fwd: input projection [Tp*N][D*G*H]                             (T)
  void* Hseq;               // h_t of direction d at Hseq + d*hseq_dstride + (t*N+n)*H; zero guard slots at t=-1, t=Tp (T)
  long hseq_dstride;
  void* S;                  // saved planes [D][Tp][N][NS*H]                                   (T)
  const float* h0;          // [D][N][H] or null
  const float* c0;
  float* hn;                // [D][N][H] or null
  float* cn;
  const void* dOut;         // bwd: [Tp][N][H]                                                 (T)
  void* dGI;                // bwd: [Tp*N][D*G*H]                                              (T)
  void* dGH;                // bwd, GRU only: dQ [D][Tp][N][H] (the n-gate slot of the hidden-side gate gradient) (T)
  float* dBacc;             // bwd: [D][N][NB*H] per-sample sums over time of the gate gradients (NB = 4 GRU: dr,dz,dn,dq; else G)
  char* xbuf;               // [NG][2 parities or 4 slots][PAR_BYTES], filled with 0xFF bytes before the launch
  long xgroup_bytes;        // 2 * PAR_BYTES
  int* err;                 // sticky device word (the host reads it)
  int* lerr;                // per-launch word in the scratch (reset before the launch; raised == 1)
  unsigned startup_ms;      // per-launch budget of the start-up wait (ds2_persist_opts.startup_ms, never 0 here)
};

// ---- storage-type traits: pair access, granule packing, gate math --------------------------------------------------
template <typename T>
struct XT;
template <>
struct XT<bf16_t> {
  static constexpr int KSZ = 32;   // K elements per k-step (one 16-byte fragment per lane and operand)
  static constexpr int EPL = 8;    // elements per lane and k-step
  static constexpr int EPG = 2;    // elements per granule
  typedef uint32_t raw;            // two adjacent elements as stored
  static __device__ __forceinline__ raw ld(const bf16_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
  static __device__ __forceinline__ void st(bf16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = cvt_pk_bf16(a, b); }
  static __device__ __forceinline__ raw zero() { return 0u; }
  static __device__ __forceinline__ float lo(raw r) { return bf_lo(r); }
  static __device__ __forceinline__ float hi(raw r) { return bf_hi(r); }
  static __device__ __forceinline__ float sig(float x) { return fsigmoid(x); }
  static __device__ __forceinline__ float tnh(float x) { return ftanh(x); }
  static __device__ __forceinline__ float rnd(float x) { return bf_lo(cvt_pk_bf16(x, 0.f)); }   // x as stored
  // publish the pair (a, b) = elements (k, k+1), k even, tagged `tag`, at byte offset `off` of the parity buffer
  static __device__ __forceinline__ void publish(char* xpar, int off, unsigned tag, float a, float b) {
    g_store(reinterpret_cast<u64*>(xpar + off), ((u64)tag << 32) | cvt_pk_bf16(a, b));
  }
};
template <>
struct XT<float> {
  static constexpr int KSZ = 16;
  static constexpr int EPL = 4;
  static constexpr int EPG = 1;
  typedef float2 raw;
  static __device__ __forceinline__ raw ld(const float* p) { return *reinterpret_cast<const float2*>(p); }
  static __device__ __forceinline__ void st(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
  static __device__ __forceinline__ raw zero() { return make_float2(0.f, 0.f); }
  static __device__ __forceinline__ float lo(raw r) { return r.x; }
  static __device__ __forceinline__ float hi(raw r) { return r.y; }
  static __device__ __forceinline__ float sig(float x) { return sigmoid_acc(x); }   // full precision: the 1e-3 parity mode
  static __device__ __forceinline__ float tnh(float x) { return tanhf_(x); }
  static __device__ __forceinline__ float rnd(float x) { return x; }
  static __device__ __forceinline__ void publish(char* xpar, int off, unsigned tag, float a, float b) {
    g_store(reinterpret_cast<u64*>(xpar + off), ((u64)tag << 32) | __float_as_uint(a));
    g_store(reinterpret_cast<u64*>(xpar + off + 8), ((u64)tag << 32) | __float_as_uint(b));
  }
};

// Exchange buffer of one group and parity: [k-step][m-tile (MT)][q (2)][lq (4)][row (16)] x 16 bytes.  A 16-byte unit = two
// granules = the elements {e, .., e + EPL/2 - 1} of sample row (mt*16 + row) with e = kstep*KSZ + lq*EPL + q*EPL/2.
template <int MT>
__device__ __forceinline__ int xunit2(int kstep, int mt, int q, int lq, int row) {
  return ((((kstep * MT + mt) * 2 + q) * 4 + lq) * 16 + row) * 16;
}
// byte offset of the first granule of the pair (k, k+1), k even
template <typename T, int MT>
__device__ __forceinline__ int xpair2(int k, int mt, int row) {
  constexpr int KSZ = XT<T>::KSZ, EPL = XT<T>::EPL, HPL = EPL / 2, EPG = XT<T>::EPG;
  const int kk = k % KSZ, e = kk % EPL;
  return xunit2<MT>(k / KSZ, mt, e / HPL, kk / EPL, row) + ((e % HPL) / EPG) * 8;
}

// gate work item of a thread -> (m-tile, row, offset of the unit pair inside the workgroup's 16 units).  The bit order follows
// the exchange layout so that the publishes of a wave are contiguous runs.
template <typename T>
__device__ __forceinline__ void item_decode(int item, int p, int& mt, int& row, int& jo);
template <>
__device__ __forceinline__ void item_decode<bf16_t>(int item, int p, int& mt, int& row, int& jo) {
  // bits: g(1) row(4) lq_local(1) q(1) mt ; the 16 units of workgroup p are half a k-step: lq = 2*(p&1) + lq_local
  (void)p;
  const int g = item & 1, lql = (item >> 5) & 1, q = (item >> 6) & 1;
  row = (item >> 1) & 15;
  mt = item >> 7;
  jo = lql * 8 + q * 4 + g * 2;
}
template <>
__device__ __forceinline__ void item_decode<float>(int item, int p, int& mt, int& row, int& jo) {
  // bits: row(4) lq(2) q(1) mt ; the 16 units of workgroup p are exactly k-step p
  (void)p;
  const int lq = (item >> 4) & 3, q = (item >> 6) & 1;
  row = item & 15;
  mt = item >> 7;
  jo = lq * 4 + q * 2;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// lane i of every 16-lane row <- lane (i + N) mod 16 (DPP row_ror:(16-N): data moves to higher lanes by 16-N)
template <int N>
__device__ __forceinline__ uint32_t row_from_plus(uint32_t v) {
  static_assert(N > 0 && N < 16, "rotation");
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x120 + (16 - N), 0xf, 0xf, true);
}
__device__ __forceinline__ uint4 row_from_plus4(const uint4& a, int n) {   // n folds to a constant after unrolling
  switch (n) {
    case 4: return make_uint4(row_from_plus<4>(a.x), row_from_plus<4>(a.y), row_from_plus<4>(a.z), row_from_plus<4>(a.w));
    case 8: return make_uint4(row_from_plus<8>(a.x), row_from_plus<8>(a.y), row_from_plus<8>(a.z), row_from_plus<8>(a.w));
    case 12: return make_uint4(row_from_plus<12>(a.x), row_from_plus<12>(a.y), row_from_plus<12>(a.z), row_from_plus<12>(a.w));
    default: return a;
  }
}

// k-steps per gather chunk.  The loads of a chunk are all in flight together (the sweep is latency-bound: the deeper the
// better) within a register budget: resident weights + accumulators + staging (8 registers per k-step, m-tile and buffer).
constexpr int chunk2(int KSW, int MT, int RT, int SP) {
  const int kpad = (KSW + SP - 1) / SP * SP;
  const int budget = 440 - RT * KSW * 4 - MT * RT * 4 - 90;
  int per_one = budget / (MT * 8);                       // k-steps per lane if ONE chunk covers the wave's slice
  if (per_one * SP >= kpad) return kpad;                 // single chunk, no double buffer
  int per = budget / (2 * MT * 8);                       // two buffers: the next chunk's loads fly under this chunk's MFMAs
  per = per < 1 ? 1 : (per > 12 ? 12 : per);
  return per * SP;
}

// Gathers this wave's k-steps [ks0, ks0 + cnt) of the exchanged vector (granules tagged `epoch`) for all MT m-tiles and
// multiplies: acc[mt][rt] += A(mt: 16 samples x K-slice) * w[rt](16 rows x K-slice)^T.  Chunks of CH k-steps; the loads of
// chunk c+1 are issued before the products of chunk c.
// SP > 1 (MT == 1, at most 16/SP samples in the group): the otherwise idle lanes of every 16-lane row share the loads --
// lane (part, srow) = (li / (16/SP), li % (16/SP)) fetches k-steps part, part + SP, ... of each chunk for sample srow and
// the fragment is rotated into place (DPP) before its MFMA: 1/SP of the load instructions per lane.
template <typename T, int MT, int RT, int KSW, int SP, bool RAGGED>
__device__ __forceinline__ void gather_mma2(ds2_f32x4 (&acc)[MT][RT], const uint4 (&w)[RT][KSW], __amdgpu_buffer_rsrc_t rsrc,
                                            int par_off, int ks0, int cnt, int lq, int li, int Ns, unsigned epoch, int* err,
                                            int* lerr, bool& dead) {
  static_assert(SP == 1 || MT == 1, "lane sharing needs a single m-tile");
  constexpr int CH = chunk2(KSW, MT, RT, SP);
  constexpr int PER = CH / SP;
  constexpr int NCH = (KSW + CH - 1) / CH;
  constexpr int NB = NCH > 1 ? 2 : 1;
  constexpr int W16 = 16 / SP;
  u32x4_t v[NB][PER][MT][2];
  const int srow = SP > 1 ? (li & (W16 - 1)) : li, part = SP > 1 ? li / W16 : 0;
  bool need[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) need[m] = m * 16 + srow < Ns;
#define DS2Q_LOAD(c, b)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                                                     \
    const int k_ = (c) * CH + part + SP * i;                                                                             \
    const bool kok = k_ < KSW && (!RAGGED || k_ < cnt);                                                                  \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int q = 0; q < 2; ++q)                        \
        v[b][i][m][q] = (kok && need[m]) ? __builtin_amdgcn_raw_buffer_load_b128(rsrc, par_off + xunit2<MT>(ks0 + k_, m, q, lq, srow), 0, 16 /* sc1 */) \
                                         : u32x4_t{0u, epoch, 0u, epoch};                                                \
  }
#define DS2Q_CHECK(b, bad)                                                                                               \
  bool bad = false;                                                                                                      \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) _Pragma("unroll") for (int m = 0; m < MT; ++m)                        \
      _Pragma("unroll") for (int q = 0; q < 2; ++q) bad |= (v[b][i][m][q][1] != epoch) | (v[b][i][m][q][3] != epoch);
  DS2Q_LOAD(0, 0)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int b = NB > 1 ? (c & 1) : 0;
    DS2Q_CHECK(b, bad0)
    if (__any(bad0) && !dead) {        // not all there yet: poll this chunk (bounded)
      unsigned spins = 0;
      for (;;) {
        __builtin_amdgcn_s_sleep(1);
        DS2Q_LOAD(c, b)
        DS2Q_CHECK(b, bad1)
        if (!__any(bad1)) break;
        if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && spin_check(lerr, spins))) {
          dead = true;
          raise_err(err, lerr);
          break;
        }
      }
    }
    if (c + 1 < NCH) { DS2Q_LOAD(c + 1, b ^ 1) }
#pragma unroll
    for (int kk = 0; kk < CH; ++kk) {
      const int k_ = c * CH + kk;                                  // compile-time after unrolling
      if (k_ < KSW && (!RAGGED || k_ < cnt)) {
        const int i = kk / SP;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          uint4 a = make_uint4(v[b][i][m][0][0], v[b][i][m][0][2], v[b][i][m][1][0], v[b][i][m][1][2]);
          if (SP > 1 && kk % SP) a = row_from_plus4(a, (kk % SP) * W16);   // the lanes of part kk % SP hold this k-step
#pragma unroll
          for (int r = 0; r < RT; ++r) Mma<T>::mma16(acc[m][r], a, w[r][k_]);
        }
      }
    }
  }
#undef DS2Q_LOAD
#undef DS2Q_CHECK
}

// ---- fp32 storage, <= 4 samples per group (config 2: the 1e-3 parity mode): 4x4x1 MFMA with A-block broadcast (round 6) ----------
// The fp32 matrix pipe is the floor of these sweeps, and with 4 clips a 16x16x4 instruction spends 12 of its 16 rows on padding
// (156 instructions x 32 cycles per wave and forward step at GRU-800).  v_mfma_f32_4x4x1_16b_f32 computes SIXTEEN independent 4x4
// blocks of K = 1: lane l = (block l / 4, index l % 4), A = row l % 4, B = column l % 4, D = the four rows of column l % 4.  With
// CBSZ = c, ABID = a the A rows of block (g * 2^c + a) are broadcast to the 2^c blocks of group g (tools/probe_mfma4x4.py checks
// this model against the hardware for every (c, a) and times it: 10 cycles per instruction against 32.5 -- with 4 clips 25.5
// instead of 7.9 useful MACs per cycle and SIMD, profiles/r06i_probe_mfma4x4_broadcast.txt).  The SP = 4 gather already holds the
// operand in exactly that shape: lane (lq, li = part * 4 + srow) = block 4 lq + part, row srow carries sample srow's values of
// k-step (chunk + part + 4 i), k = 4 lq + e -- no DPP rotation any more, the broadcast selects the block.
//   forward (Q4F): CBSZ 4 -- one instruction = 4 clips x 64 unit slots (gate g = slot / 16, unit slot % 16 of the workgroup) x ONE k;
//                  the weights are one float per lane and k: wq[k-step][16] (GRU: 48 of the 64 slots used);
//   BPTT (Q4B):    CBSZ 2 -- the four block groups (= lq) take four different k (their own lq), a group's four blocks the
//                  workgroup's 16 units: one instruction = 4 clips x 16 units x FOUR k; the weights stay the uint4 fragments of the
//                  16x16x4 form (element e of w[0][k-step]); the four groups' partial sums are added with the waves' in the gate phase.
// Exact fp32 products as before: only the summation order changes.
template <int KSW, int SP, bool RAGGED, bool BWD, typename WQ, typename WF>
__device__ __forceinline__ void gather_mma2_q4(ds2_f32x4& out, const WQ& wq /* forward: float [KSW][16] */, const WF& w /* BPTT: uint4 [1][KSW] */,
                                               __amdgpu_buffer_rsrc_t rsrc, int par_off, int ks0, int cnt, int lq, int li, int Ns,
                                               unsigned epoch, int* err, int* lerr, bool& dead) {
  static_assert(SP == 4, "4x4x1 blocks: four samples per group");
  constexpr int MT = 1, RT = BWD ? 1 : 4;        // (RT only sizes the gather chunk: resident weight registers / (4 KSW))
  constexpr int CH = chunk2(KSW, MT, RT, SP);
  constexpr int PER = CH / SP;
  constexpr int NCH = (KSW + CH - 1) / CH;
  constexpr int NB = NCH > 1 ? 2 : 1;
  u32x4_t v[NB][PER][MT][2];
  const int srow = li & 3, part = li >> 2;
  bool need[MT];
  need[0] = srow < Ns;
  ds2_f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
#define DS2Q_LOAD(c, b)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                                                     \
    const int k_ = (c) * CH + part + SP * i;                                                                             \
    const bool kok = k_ < KSW && (!RAGGED || k_ < cnt);                                                                  \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int q = 0; q < 2; ++q)                        \
        v[b][i][m][q] = (kok && need[m]) ? __builtin_amdgcn_raw_buffer_load_b128(rsrc, par_off + xunit2<MT>(ks0 + k_, m, q, lq, srow), 0, 16 /* sc1 */) \
                                         : u32x4_t{0u, epoch, 0u, epoch};                                                \
  }
#define DS2Q_CHECK(b, bad)                                                                                               \
  bool bad = false;                                                                                                      \
  _Pragma("unroll") for (int i = 0; i < PER; ++i) _Pragma("unroll") for (int m = 0; m < MT; ++m)                        \
      _Pragma("unroll") for (int q = 0; q < 2; ++q) bad |= (v[b][i][m][q][1] != epoch) | (v[b][i][m][q][3] != epoch);
  DS2Q_LOAD(0, 0)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int b = NB > 1 ? (c & 1) : 0;
    DS2Q_CHECK(b, bad0)
    if (__any(bad0) && !dead) {        // not all there yet: poll this chunk (bounded)
      unsigned spins = 0;
      for (;;) {
        __builtin_amdgcn_s_sleep(1);
        DS2Q_LOAD(c, b)
        DS2Q_CHECK(b, bad1)
        if (!__any(bad1)) break;
        if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && spin_check(lerr, spins))) {
          dead = true;
          raise_err(err, lerr);
          break;
        }
      }
    }
    if (c + 1 < NCH) { DS2Q_LOAD(c + 1, b ^ 1) }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int kb = c * CH + SP * i;                              // k-steps kb .. kb + 3 (lane part p holds kb + p)
      const float ae[4] = {__uint_as_float(v[b][i][0][0][0]), __uint_as_float(v[b][i][0][0][2]), __uint_as_float(v[b][i][0][1][0]),
                           __uint_as_float(v[b][i][0][1][2])};
      if constexpr (!BWD) {
        // block beta = 4 lq' + part' broadcasts h[sample][16 (kb + part') + 4 lq' + e] to all 16 blocks
#define DS2Q_M4(E, B_)                                                                                                   \
        if (kb + ((B_) & 3) < KSW)                                                                                       \
          acc[(B_) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(ae[E], wq[kb + ((B_) & 3) < KSW ? kb + ((B_) & 3) : 0][4 * ((B_) >> 2) + (E)], acc[(B_) & 3], 4, B_, 0);
#define DS2Q_M4E(E) DS2Q_M4(E, 0) DS2Q_M4(E, 1) DS2Q_M4(E, 2) DS2Q_M4(E, 3) DS2Q_M4(E, 4) DS2Q_M4(E, 5) DS2Q_M4(E, 6) DS2Q_M4(E, 7) \
                    DS2Q_M4(E, 8) DS2Q_M4(E, 9) DS2Q_M4(E, 10) DS2Q_M4(E, 11) DS2Q_M4(E, 12) DS2Q_M4(E, 13) DS2Q_M4(E, 14) DS2Q_M4(E, 15)
        DS2Q_M4E(0) DS2Q_M4E(1) DS2Q_M4E(2) DS2Q_M4E(3)
#undef DS2Q_M4E
#undef DS2Q_M4
      } else {
        // group lq' takes block 4 lq' + alpha: dG[sample][16 (kb + alpha) + 4 lq' + e]; this lane's weight = its own fragment's element e
#define DS2Q_M2(E, A_)                                                                                                   \
        if (kb + (A_) < KSW) {                                                                                           \
          const uint4& wf = w[0][kb + (A_) < KSW ? kb + (A_) : 0];                                                       \
          const uint32_t wb = (E) == 0 ? wf.x : (E) == 1 ? wf.y : (E) == 2 ? wf.z : wf.w;                                \
          acc[A_] = __builtin_amdgcn_mfma_f32_4x4x1f32(ae[E], __uint_as_float(wb), acc[A_], 2, A_, 0);                   \
        }
#define DS2Q_M2E(E) DS2Q_M2(E, 0) DS2Q_M2(E, 1) DS2Q_M2(E, 2) DS2Q_M2(E, 3)
        DS2Q_M2E(0) DS2Q_M2E(1) DS2Q_M2E(2) DS2Q_M2E(3)
#undef DS2Q_M2E
#undef DS2Q_M2
      }
    }
  }
#undef DS2Q_LOAD
#undef DS2Q_CHECK
  out = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// partial sums of the Q4 forms.  Forward: part[wave][sample (4)][unit slot (64)]; BPTT: part[wave * 4 + lq][sample (4)][unit (16)]
__device__ __forceinline__ void store_partials_q4f(float* part, const ds2_f32x4& acc, int wave, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) part[(wave * 4 + r) * 64 + lane] = acc[r];
}
__device__ __forceinline__ float2 load_partials_q4f(const float* part, int g, int row, int col) {
  float2 s = make_float2(0.f, 0.f);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float2 v = *reinterpret_cast<const float2*>(part + (w * 4 + row) * 64 + g * 16 + col);
    s.x += v.x;
    s.y += v.y;
  }
  return s;
}
__device__ __forceinline__ void store_partials_q4b(float* part, const ds2_f32x4& acc, int wave, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) part[((wave * 4 + (lane >> 4)) * 4 + r) * 16 + (lane & 15)] = acc[r];
}
__device__ __forceinline__ float2 load_partials_q4b(const float* part, int row, int col) {
  float2 s = make_float2(0.f, 0.f);
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const float2 v = *reinterpret_cast<const float2*>(part + (w * 4 + row) * 16 + col);
    s.x += v.x;
    s.y += v.y;
  }
  return s;
}
#ifndef DS2Q_Q4
#define DS2Q_Q4 1        // 0: the 16x16x4 form everywhere (round 5), for A/B runs
#endif

// ---- payload-only exchange (MT >= 2, bf16): the large-batch regime is bound by the gathered BYTES (a workgroup reads N_s x K
// values per step through one CU's vector-memory path), so the granule tags are dropped: payload = plain bf16 in A-fragment order
// [k-step][m-tile][lq (4)][row (16)] x 16 bytes (ONE 16-byte load = one complete MFMA A fragment: half the bytes and half the load
// instructions of the tagged form).  Completion is carried by the data itself: a dword that still holds the all-ones sentinel
// has not been published (gather_mma2f).  Until round 2d this form used one flag word per producing workgroup instead (payload
// stores -> vmcnt(0) -> barrier -> flag store; consumers polled the flags, then loaded the payload): three dependent round trips
// and an extra workgroup barrier per step.
template <int MT>
__device__ __forceinline__ int xunitf(int kstep, int mt, int lq, int row) { return (((kstep * MT + mt) * 4 + lq) * 16 + row) * 16; }
template <int MT>
__device__ __forceinline__ int xpayf(int k, int mt, int row) {   // byte offset of the bf16 pair (k, k+1), k even
  const int kk = k % 32;
  return xunitf<MT>(k / 32, mt, kk / 8, row) + (kk % 8) * 2;
}
// A buffer offset beyond every exchange resource: the range check returns zeros.  Rows without a sample are loaded from there
// instead of being skipped by a branch (round 3: BPTT 11.8 -> 10.2 us per time step at config 5a, 6.45 -> 5.39 at 5b; the same
// change in the tagged gather_mma2 cost config 2's forward sweep 1.3 us per step and was not kept there).
constexpr int XOOB = 0x7ffffff0;
constexpr uint32_t XSENT2 = 0xffffffffu;   // "not published yet": see gather_mma_tf in ds2_rnn_persist_impl.h (same protocol)
__device__ __forceinline__ void publish_pay(char* xslot, int off, float a, float b) {
  const uint32_t pk = cvt_pk_bf16(a, b);
  __hip_atomic_store(reinterpret_cast<unsigned*>(xslot + off), pk == XSENT2 ? 0x7fc07fc0u : pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rearm_pay(char* xslot, int off) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(xslot + off), XSENT2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KSW, int MT, int RT>
constexpr int chunkf(int) {
  const int budget = 440 - RT * KSW * 4 - MT * RT * 4 - 90;
  int per_one = budget / (MT * 4);
  if (per_one >= KSW) return KSW;
  int per = budget / (2 * MT * 4);
  return per < 1 ? 1 : (per > 16 ? 16 : per);
}

// Payload-only gather: FOUR slots (step e publishes into slot e & 3), the all-ones dword marks "not published yet", every
// publisher re-arms its dwords of slot (s + 2) & 3 at step s (the protocol and its ordering argument: gather_mma_tf in
// ds2_rnn_persist_impl.h).  One round trip per step where the flag form needed three (payload acknowledged -> barrier -> flag
// store -> flag poll -> payload load).  The loads of chunk c+1 are in flight while chunk c is checked and multiplied; a chunk
// with a sentinel in it is re-polled (bounded) before its MFMAs -- no speculative products, nothing to undo.
template <typename T, int MT, int RT, int KSW, bool RAGGED>
__device__ __forceinline__ void gather_mma2f(ds2_f32x4 (&acc)[MT][RT], const uint4 (&w)[RT][KSW], __amdgpu_buffer_rsrc_t rsrc,
                                             int slot_off, int ks0, int cnt, int lq, int li, int Ns, int* err, int* lerr, bool& dead) {
  constexpr int CH = chunkf<KSW, MT, RT>(0);
  constexpr int NCH = (KSW + CH - 1) / CH;
  constexpr int NB = NCH > 1 ? 2 : 1;
  u32x4_t v[NB][CH][MT];
  bool need[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) need[m] = m * 16 + li < Ns;
#define DS2Q_LOADF(c, b)                                                                                                 \
  _Pragma("unroll") for (int i = 0; i < CH; ++i) {                                                                      \
    const int k_ = (c) * CH + i;                                                                                         \
    if (k_ < KSW && (!RAGGED || k_ < cnt)) {                                                                             \
      _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                     \
          v[b][i][m] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, need[m] ? slot_off + xunitf<MT>(ks0 + k_, m, lq, li) : XOOB, 0, \
                                                             16 /* sc1 */);   /* beyond the range: zeros, no branch */   \
    }                                                                                                                    \
  }
#define DS2Q_CHECKF(c, b, bad)                                                                                           \
  bool bad;                                                                                                              \
  {                                                                                                                      \
    uint32_t mx = 0;                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < CH; ++i) {                                                                    \
      const int k_ = (c) * CH + i;                                                                                       \
      if (k_ < KSW && (!RAGGED || k_ < cnt)) {                                                                           \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                   \
            mx = max(max(mx, max(v[b][i][m][0], v[b][i][m][1])), max(v[b][i][m][2], v[b][i][m][3]));                     \
      }                                                                                                                  \
    }                                                                                                                    \
    bad = mx == XSENT2;   /* rows that are not loaded hold zeros */                                                      \
  }
  DS2Q_LOADF(0, 0)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int b = NB > 1 ? (c & 1) : 0;
    if (c + 1 < NCH) { DS2Q_LOADF(c + 1, b ^ 1) }
    DS2Q_CHECKF(c, b, bad0)
    if (__any(bad0) && !dead) {
      unsigned spins = 0;
      for (;;) {
        __builtin_amdgcn_s_sleep(1);
        DS2Q_LOADF(c, b)
        DS2Q_CHECKF(c, b, bad1)
        if (!__any(bad1)) break;
        if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && spin_check(lerr, spins))) {
          dead = true;
          raise_err(err, lerr);
          break;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int k_ = c * CH + i;
      if (k_ < KSW && (!RAGGED || k_ < cnt)) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const uint4 a = make_uint4(v[b][i][m][0], v[b][i][m][1], v[b][i][m][2], v[b][i][m][3]);
#pragma unroll
          for (int r = 0; r < RT; ++r) Mma<T>::mma16(acc[m][r], a, w[r][k_]);
        }
      }
    }
  }
#undef DS2Q_LOADF
#undef DS2Q_CHECKF
}

// partial sums of the 4 K-slices: part[wave][mt][rt][sample row][unit col]
template <int MT, int RT>
__device__ __forceinline__ void store_partials2(float* part, const ds2_f32x4 (&acc)[MT][RT], int wave, int lane) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(((wave * MT + m) * RT + t) * 16 + mma16_row(r, lane)) * 16 + (lane & 15)] = acc[m][t][r];
}
template <int MT, int RT>
__device__ __forceinline__ float2 load_partials2(const float* part, int m, int t, int row, int col) {
  float2 s = make_float2(0.f, 0.f);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float2 v = *reinterpret_cast<const float2*>(part + (((w * MT + m) * RT + t) * 16 + row) * 16 + col);
    s.x += v.x;
    s.y += v.y;
  }
  return s;
}

template <int MT>
struct Items {
  static constexpr int TOTAL = MT * 128;                       // (sample row, unit pair) items of a workgroup
  static constexpr int PER_THREAD = (TOTAL + 255) / 256;
  static constexpr int FIRST_TID = TOTAL >= 256 ? 0 : 256 - TOTAL;   // MT = 1: the LAST 128 threads (waves 2-3)
};

constexpr float QNAN = __builtin_nanf("");

// ------------------------------------------------------------------------------------------------------------------
// forward sweep
// ------------------------------------------------------------------------------------------------------------------
template <int CELL, typename T, int H, int MT, int SP>
__global__ void __launch_bounds__(256, 1) k_rnn_persist2_fwd(QArgs a) {
  typedef XT<T> X;
  typedef typename X::raw raw_t;
  constexpr int G = CellInfo<CELL>::G, NS = CellInfo<CELL>::NS;
  constexpr int RT = G;                                  // 16-row tiles of the resident slice: tile g = gate g, 16 units
  constexpr int KSZ = X::KSZ, KT = H / KSZ, KSW = (KT + 3) / 4;
  constexpr bool RAGGED = KT % 4 != 0;
  constexpr int IT = Items<MT>::PER_THREAD;
  constexpr bool FLAGS = sizeof(T) == 2 && MT >= 2;        // > 16 samples per group, bf16: payload-only exchange in four slots
  constexpr int PAR_BYTES = KT * MT * (FLAGS ? 1024 : 2048);   // one slot (payload-only) / one parity (tagged granules)
  static_assert(H % KSZ == 0 && H % 16 == 0, "unsupported hidden size");
  // fp32, <= 4 samples per group, GRU / LSTM: the 4x4x1 broadcast form (gather_mma2_q4); tanh cells would use 16 of its 64 unit slots
  constexpr bool Q4 = DS2Q_Q4 && std::is_same<T, float>::value && MT == 1 && SP == 4 && G >= 3;
  extern __shared__ __attribute__((aligned(16))) float part[];       // [2][4][MT][RT][256]
  constexpr int PART_FLOATS = 4 * MT * RT * 256;
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-uniform predicates
  const int grp = blockIdx.x % a.NG, p = blockIdx.x / a.NG;
  const int d = grp / a.gpd, slice = grp % a.gpd;
  const int N = a.N, Tp = a.Tp;
  const int Ns = (N - slice + a.gpd - 1) / a.gpd;          // samples n = slice + gpd*i, i < Ns
  const int li = lane & 15, lq = lane >> 4;
  constexpr long GH = (long)G * H;
  const long ldgi = (long)a.D * GH;
  const int ks0 = wave * KSW;
  const int cnt = RAGGED ? max(0, min(KSW, KT - ks0)) : KSW;

  uint4 w[Q4 ? 1 : RT][Q4 ? 1 : KSW];
  float wq[Q4 ? KSW : 1][16];          // Q4: lane = unit slot (gate lane / 16, unit lane % 16), one float per k of the wave's K-quarter
  if constexpr (Q4) {
    const int g_ = lane >> 4;
    const float* row = (const float*)a.W + (long)d * GH * H + ((long)min(g_, G - 1) * H + p * 16 + (lane & 15)) * H;
#pragma unroll
    for (int k = 0; k < KSW; ++k) {
      const bool ok = g_ < G && (!RAGGED || k < cnt);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 f = ok ? *reinterpret_cast<const float4*>(row + (long)(ks0 + k) * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        wq[k][4 * q] = f.x; wq[k][4 * q + 1] = f.y; wq[k][4 * q + 2] = f.z; wq[k][4 * q + 3] = f.w;
      }
    }
  } else {
    const T* Wd = (const T*)a.W + (long)d * GH * H;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const T* row = Wd + ((long)t * H + p * 16 + li) * H + lq * X::EPL;
#pragma unroll
      for (int k = 0; k < KSW; ++k)
        w[t][k] = (!RAGGED || k < cnt) ? *reinterpret_cast<const uint4*>(row + (long)(ks0 + k) * KSZ) : make_uint4(0, 0, 0, 0);
    }
  }
  char* xg = a.xbuf + (long)grp * a.xgroup_bytes;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, (FLAGS ? 4 : 2) * PAR_BYTES, 0x00020000);

  // ---- gate items of this thread
  bool on[IT];
  int it_row[IT], it_mt[IT], it_xoff[IT], it_n[IT], it_j[IT], it_len[IT];
  float hprev[IT][2], cprev[IT][2], bh[IT][G][2];
  const T* gi_ptr[IT];
  T* sv_ptr[IT];
  T* hs_ptr[IT];
  const long dstep = d == 0 ? 1 : -1;
  const int t_first = d == 0 ? 0 : Tp - 1;
  constexpr long NSH_ = (long)(NS ? NS : 1) * H;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int item = tid - Items<MT>::FIRST_TID + 256 * it;
    int mt = 0, row = 0, jo = 0;
    item_decode<T>(item < 0 ? 0 : item, p, mt, row, jo);
    const int gi_i = mt * 16 + row;
    on[it] = item >= 0 && item < Items<MT>::TOTAL && gi_i < Ns;
    it_row[it] = row;
    it_mt[it] = mt;
    it_j[it] = p * 16 + jo;
    it_n[it] = on[it] ? slice + a.gpd * gi_i : 0;
    it_xoff[it] = FLAGS ? xpayf<MT>(it_j[it], mt, row) : xpair2<T, MT>(it_j[it], mt, row);
    it_len[it] = 0;
    hprev[it][0] = hprev[it][1] = cprev[it][0] = cprev[it][1] = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) bh[it][g][0] = bh[it][g][1] = 0.f;
    const int n = it_n[it], j = it_j[it];
    if (on[it]) {
      it_len[it] = a.lens[n];
      const long so = ((long)d * N + n) * H + j;
      if (a.h0) {
        hprev[it][0] = a.h0[so];
        hprev[it][1] = a.h0[so + 1];
      }
      if (CELL == CELL_LSTM && a.c0) {
        cprev[it][0] = a.c0[so];
        cprev[it][1] = a.c0[so + 1];
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        bh[it][g][0] = a.bhh[(long)d * GH + (long)g * H + j];
        bh[it][g][1] = a.bhh[(long)d * GH + (long)g * H + j + 1];
      }
    }
    gi_ptr[it] = (const T*)a.GI + ((long)t_first * N + n) * ldgi + (long)d * GH + j;
    sv_ptr[it] = NS ? (T*)a.S + (((long)d * Tp + t_first) * N + n) * NSH_ + j : nullptr;
    hs_ptr[it] = (T*)a.Hseq + (long)d * a.hseq_dstride + ((long)t_first * N + n) * H + j;
  }
  const long gi_stride = dstep * N * ldgi, sv_stride = dstep * N * NSH_, hs_stride = dstep * N * H;
  bool dead = false;
  wait_all_resident(a.lerr, tid, a.err, a.startup_ms, dead);   // no wait of the sweep before all of the launch's workgroups are resident
#pragma unroll
  for (int it = 0; it < IT; ++it)
    if (on[it]) {   // zero guard slots of the state sequence at t = -1 and t = T' ("previous h" reads are unconditional)
      T* hb = (T*)a.Hseq + (long)d * a.hseq_dstride + (long)it_n[it] * H + it_j[it];
      X::st(hb - (long)N * H, 0.f, 0.f);
      X::st(hb + (long)Tp * N * H, 0.f, 0.f);
    }
  if (a.h0) {   // initial state as "step -1": parity 1, tag TAG_INIT
#pragma unroll
    for (int it = 0; it < IT; ++it)
      if (on[it]) {
        if (FLAGS) publish_pay(xg + 3 * PAR_BYTES, it_xoff[it], hprev[it][0], hprev[it][1]);     // "step -1": slot 3
        else X::publish(xg + PAR_BYTES, it_xoff[it], TAG_INIT, hprev[it][0], hprev[it][1]);
      }
  }

  for (int s = 0; s < Tp; ++s) {
    const int t = d == 0 ? s : Tp - 1 - s;
    const int par = s & 1;
    raw_t gi[IT][G];
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int g = 0; g < G; ++g) gi[it][g] = on[it] ? X::ld(gi_ptr[it] + (long)g * H) : X::zero();
    ds2_f32x4 acc[MT][RT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int tt = 0; tt < RT; ++tt) acc[m][tt] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
    float* pp = part + par * PART_FLOATS;
    if constexpr (Q4) {
      ds2_f32x4 accq = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
      if (s > 0 || a.h0)
        gather_mma2_q4<KSW, SP, RAGGED, false>(accq, wq, w, rsrc, (par ^ 1) * PAR_BYTES, ks0, cnt, lq, li, Ns, s > 0 ? (unsigned)s : TAG_INIT,
                                               a.err, a.lerr, dead);
      store_partials_q4f(pp, accq, wave, lane);
    } else {
      if (s > 0 || a.h0) {
        if constexpr (FLAGS)
          gather_mma2f<T, MT, RT, KSW, RAGGED>(acc, w, rsrc, ((s + 3) & 3) * PAR_BYTES, ks0, cnt, lq, li, Ns, a.err, a.lerr, dead);
        else
          gather_mma2<T, MT, RT, KSW, SP, RAGGED>(acc, w, rsrc, (par ^ 1) * PAR_BYTES, ks0, cnt, lq, li, Ns, s > 0 ? (unsigned)s : TAG_INIT,
                                                  a.err, a.lerr, dead);
      }
      store_partials2<MT, RT>(pp, acc, wave, lane);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      if (on[it]) {
        const bool act = t < it_len[it];
        const int jo = it_j[it] - p * 16;
        float hn0 = 0.f, hn1 = 0.f;     // emitted h_t (0 when inactive)
        float2 gh[G];
#pragma unroll
        for (int g = 0; g < G; ++g) gh[g] = Q4 ? load_partials_q4f(pp, g, it_row[it], jo) : load_partials2<MT, RT>(pp, it_mt[it], g, it_row[it], jo);
        float pl[NS ? NS : 1][2];
#pragma unroll
        for (int q = 0; q < (NS ? NS : 1); ++q) pl[q][0] = pl[q][1] = 0.f;
        if (CELL == CELL_GRU) {
          if (act) {
            const float q0 = gh[2 % G].x + bh[it][2 % G][0], q1 = gh[2 % G].y + bh[it][2 % G][1];
            const float r0 = X::sig(X::lo(gi[it][0]) + gh[0].x + bh[it][0][0]), r1 = X::sig(X::hi(gi[it][0]) + gh[0].y + bh[it][0][1]);
            const float z0 = X::sig(X::lo(gi[it][1 % G]) + gh[1 % G].x + bh[it][1 % G][0]);
            const float z1 = X::sig(X::hi(gi[it][1 % G]) + gh[1 % G].y + bh[it][1 % G][1]);
            const float n0 = X::tnh(X::lo(gi[it][2 % G]) + r0 * q0), n1 = X::tnh(X::hi(gi[it][2 % G]) + r1 * q1);
            hn0 = (1.f - z0) * n0 + z0 * hprev[it][0];
            hn1 = (1.f - z1) * n1 + z1 * hprev[it][1];
            hprev[it][0] = hn0;
            hprev[it][1] = hn1;
            pl[0][0] = r0; pl[0][1] = r1;
            pl[1 % (NS ? NS : 1)][0] = z0; pl[1 % (NS ? NS : 1)][1] = z1;
            pl[2 % (NS ? NS : 1)][0] = n0; pl[2 % (NS ? NS : 1)][1] = n1;
            pl[3 % (NS ? NS : 1)][0] = q0; pl[3 % (NS ? NS : 1)][1] = q1;
          }
        } else if (CELL == CELL_LSTM) {
          if (act) {
            const float i0 = X::sig(X::lo(gi[it][0]) + gh[0].x + bh[it][0][0]), i1 = X::sig(X::hi(gi[it][0]) + gh[0].y + bh[it][0][1]);
            const float f0 = X::sig(X::lo(gi[it][1 % G]) + gh[1 % G].x + bh[it][1 % G][0]);
            const float f1 = X::sig(X::hi(gi[it][1 % G]) + gh[1 % G].y + bh[it][1 % G][1]);
            const float g0 = X::tnh(X::lo(gi[it][2 % G]) + gh[2 % G].x + bh[it][2 % G][0]);
            const float g1 = X::tnh(X::hi(gi[it][2 % G]) + gh[2 % G].y + bh[it][2 % G][1]);
            const float o0 = X::sig(X::lo(gi[it][3 % G]) + gh[3 % G].x + bh[it][3 % G][0]);
            const float o1 = X::sig(X::hi(gi[it][3 % G]) + gh[3 % G].y + bh[it][3 % G][1]);
            const float c0 = f0 * cprev[it][0] + i0 * g0, c1 = f1 * cprev[it][1] + i1 * g1;
            hn0 = o0 * X::tnh(c0);
            hn1 = o1 * X::tnh(c1);
            cprev[it][0] = c0;
            cprev[it][1] = c1;
            hprev[it][0] = hn0;
            hprev[it][1] = hn1;
            pl[0][0] = i0; pl[0][1] = i1;
            pl[1 % (NS ? NS : 1)][0] = f0; pl[1 % (NS ? NS : 1)][1] = f1;
            pl[2 % (NS ? NS : 1)][0] = g0; pl[2 % (NS ? NS : 1)][1] = g1;
            pl[3 % (NS ? NS : 1)][0] = o0; pl[3 % (NS ? NS : 1)][1] = o1;
            pl[4 % (NS ? NS : 1)][0] = c0; pl[4 % (NS ? NS : 1)][1] = c1;
          }
        } else {
          if (act) {
            hn0 = X::tnh(X::lo(gi[it][0]) + gh[0].x + bh[it][0][0]);
            hn1 = X::tnh(X::hi(gi[it][0]) + gh[0].y + bh[it][0][1]);
            hprev[it][0] = hn0;
            hprev[it][1] = hn1;
          }
        }
        if (dead) hn0 = hn1 = hprev[it][0] = hprev[it][1] = QNAN;   // fail loudly downstream
        // publish the carried state first (inactive samples republish their unchanged state), then the bookkeeping stores
        if (FLAGS) {
          publish_pay(xg + (s & 3) * PAR_BYTES, it_xoff[it], hprev[it][0], hprev[it][1]);
          rearm_pay(xg + ((s + 2) & 3) * PAR_BYTES, it_xoff[it]);
        } else {
          X::publish(xg + par * PAR_BYTES, it_xoff[it], (unsigned)(s + 1), hprev[it][0], hprev[it][1]);
        }
        X::st(hs_ptr[it], hn0, hn1);
#pragma unroll
        for (int q = 0; q < NS; ++q) X::st(sv_ptr[it] + (long)q * H, pl[q][0], pl[q][1]);
      }
      gi_ptr[it] += gi_stride;
      if (NS) sv_ptr[it] += sv_stride;
      hs_ptr[it] += hs_stride;
    }
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    if (on[it]) {
      const long so = ((long)d * N + it_n[it]) * H + it_j[it];
      if (a.hn) {
        a.hn[so] = hprev[it][0];
        a.hn[so + 1] = hprev[it][1];
      }
      if (CELL == CELL_LSTM && a.cn) {
        a.cn[so] = cprev[it][0];
        a.cn[so + 1] = cprev[it][1];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BPTT sweep.  dh_t = dOut[t] + carry (elementwise part of dh from the step processed before) + dgates_{t'} * W_hh.
// The workgroup owns W_hh^T rows of its 16 units (one 16-row tile), K = G*H.
// ------------------------------------------------------------------------------------------------------------------
template <int CELL, typename T, int H, int MT, int SP>
__global__ void __launch_bounds__(256, 1) k_rnn_persist2_bwd(QArgs a) {
  typedef XT<T> X;
  typedef typename X::raw raw_t;
  constexpr int G = CellInfo<CELL>::G, NS = CellInfo<CELL>::NS;
  constexpr int RT = 1;
  constexpr int KSZ = X::KSZ, KT = G * H / KSZ, KSW = (KT + 3) / 4, KTH = H / KSZ;   // KTH: k-steps per gate
  constexpr bool RAGGED = KT % 4 != 0;
  constexpr int IT = Items<MT>::PER_THREAD;
  constexpr bool FLAGS = sizeof(T) == 2 && MT >= 2;        // payload-only exchange in four slots (see the forward kernel)
  constexpr int PAR_BYTES = KT * MT * (FLAGS ? 1024 : 2048);
  constexpr int GATE_BYTES = KTH * MT * (FLAGS ? 1024 : 2048);   // exchange bytes of one gate's H elements
  static_assert(H % KSZ == 0 && H % 16 == 0, "unsupported hidden size");
  extern __shared__ __attribute__((aligned(16))) float part[];       // [2][4][MT][1][256]
  constexpr int PART_FLOATS = 4 * MT * RT * 256;
  constexpr bool Q4 = DS2Q_Q4 && std::is_same<T, float>::value && MT == 1 && SP == 4 && G >= 3;   // see gather_mma2_q4
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-uniform predicates
  const int grp = blockIdx.x % a.NG, p = blockIdx.x / a.NG;
  const int d = grp / a.gpd, slice = grp % a.gpd;
  const int N = a.N, Tp = a.Tp;
  const int Ns = (N - slice + a.gpd - 1) / a.gpd;
  const int li = lane & 15, lq = lane >> 4;
  constexpr long GH = (long)G * H;
  const long ldgi = (long)a.D * GH;
  const int ks0 = wave * KSW;
  const int cnt = RAGGED ? max(0, min(KSW, KT - ks0)) : KSW;

  uint4 w[RT][KSW];
  {
    const T* WT = (const T*)a.W + (long)d * H * GH;
    const T* row = WT + (long)(p * 16 + li) * GH + lq * X::EPL;
#pragma unroll
    for (int k = 0; k < KSW; ++k)
      w[0][k] = (!RAGGED || k < cnt) ? *reinterpret_cast<const uint4*>(row + (long)(ks0 + k) * KSZ) : make_uint4(0, 0, 0, 0);
  }
  char* xg = a.xbuf + (long)grp * a.xgroup_bytes;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, (FLAGS ? 4 : 2) * PAR_BYTES, 0x00020000);

  bool on[IT];
  int it_row[IT], it_mt[IT], it_xoff[IT], it_j[IT], it_len[IT];
  constexpr int NB = CELL == CELL_GRU ? 4 : G;
  int it_n[IT];
  float car[IT][2], dc[IT][2], bsum[IT][NB][2];
  const T* do_ptr[IT];
  const T* sv_ptr[IT];
  const T* hs_ptr[IT];
  T* dgi_ptr[IT];
  T* dgh_ptr[IT];
  const long dstep = d == 0 ? -1 : 1;                       // BPTT walks the direction's time axis backwards
  const int t_first = d == 0 ? Tp - 1 : 0;
  const long prev_off = d == 0 ? -1 : 1;                    // previous step in FORWARD order of this direction
  constexpr long NSH_ = (long)(NS ? NS : 1) * H;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int item = tid - Items<MT>::FIRST_TID + 256 * it;
    int mt = 0, row = 0, jo = 0;
    item_decode<T>(item < 0 ? 0 : item, p, mt, row, jo);
    const int gi_i = mt * 16 + row;
    on[it] = item >= 0 && item < Items<MT>::TOTAL && gi_i < Ns;
    it_row[it] = row;
    it_mt[it] = mt;
    it_j[it] = p * 16 + jo;
    const int n = on[it] ? slice + a.gpd * gi_i : 0, j = it_j[it];
    it_xoff[it] = FLAGS ? xpayf<MT>(j, mt, row) : xpair2<T, MT>(j, mt, row);
    it_len[it] = on[it] ? a.lens[n] : 0;
    car[it][0] = car[it][1] = dc[it][0] = dc[it][1] = 0.f;
    it_n[it] = n;
#pragma unroll
    for (int g = 0; g < NB; ++g) bsum[it][g][0] = bsum[it][g][1] = 0.f;
    do_ptr[it] = (const T*)a.dOut + ((long)t_first * N + n) * H + j;
    sv_ptr[it] = NS ? (const T*)a.S + (((long)d * Tp + t_first) * N + n) * NSH_ + j : nullptr;
    hs_ptr[it] = (const T*)a.Hseq + (long)d * a.hseq_dstride + ((long)t_first * N + n) * H + j;   // h_t
    dgi_ptr[it] = (T*)a.dGI + ((long)t_first * N + n) * ldgi + (long)d * GH + j;
    dgh_ptr[it] = a.dGH ? (T*)a.dGH + (((long)d * Tp + t_first) * N + n) * H + j : nullptr;     // dQ
  }
  bool dead = false;
  wait_all_resident(a.lerr, tid, a.err, a.startup_ms, dead);   // no wait of the sweep before all of the launch's workgroups are resident

  for (int s = 0; s < Tp; ++s) {
    const int t = d == 0 ? Tp - 1 - s : s;
    const int par = s & 1;
    // ---- prefetch everything the gate phase needs
    raw_t dout[IT], sp[IT][NS ? NS : 1], hp[IT], cp[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      dout[it] = hp[it] = cp[it] = X::zero();
#pragma unroll
      for (int q = 0; q < (NS ? NS : 1); ++q) sp[it][q] = X::zero();
      if (on[it]) {
        dout[it] = X::ld(do_ptr[it]);
#pragma unroll
        for (int q = 0; q < NS; ++q) sp[it][q] = X::ld(sv_ptr[it] + (long)q * H);
        // h_{prev}: guard slots / inactive frames hold zeros, so the read is unconditional (tprev in [-1, Tp])
        hp[it] = X::ld(hs_ptr[it] + prev_off * N * H);
        if (CELL == CELL_LSTM) {
          const bool has_prev = d == 0 ? (t > 0) : (t + 1 < it_len[it]);
          if (has_prev) cp[it] = X::ld(sv_ptr[it] + prev_off * N * NSH_ + 4 * H);
        }
        if (CELL == CELL_RNN) hp[it] = X::ld(hs_ptr[it]);
      }
    }
    ds2_f32x4 acc[MT][RT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][0] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      if constexpr (FLAGS)
        gather_mma2f<T, MT, RT, KSW, RAGGED>(acc, w, rsrc, ((s + 3) & 3) * PAR_BYTES, ks0, cnt, lq, li, Ns, a.err, a.lerr, dead);
      else if constexpr (Q4)
        gather_mma2_q4<KSW, SP, RAGGED, true>(acc[0][0], w, w, rsrc, (par ^ 1) * PAR_BYTES, ks0, cnt, lq, li, Ns, (unsigned)s, a.err, a.lerr, dead);
      else
        gather_mma2<T, MT, RT, KSW, SP, RAGGED>(acc, w, rsrc, (par ^ 1) * PAR_BYTES, ks0, cnt, lq, li, Ns, (unsigned)s, a.err, a.lerr, dead);
    }
    float* pp = part + par * PART_FLOATS;
    if constexpr (Q4)
      store_partials_q4b(pp, acc[0][0], wave, lane);
    else
      store_partials2<MT, RT>(pp, acc, wave, lane);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      if (on[it]) {
        const bool act = t < it_len[it];
        const int jo = it_j[it] - p * 16;
        const float2 mp = Q4 ? load_partials_q4b(pp, it_row[it], jo) : load_partials2<MT, RT>(pp, it_mt[it], 0, it_row[it], jo);
        const float din0 = car[it][0] + mp.x, din1 = car[it][1] + mp.y;
        char* xo = xg + (FLAGS ? (s & 3) : par) * PAR_BYTES;
        char* xr = xg + ((s + 2) & 3) * PAR_BYTES;      // payload-only form: the slot re-armed for step s + 2
        const unsigned tag = (unsigned)(s + 1);
#define DS2Q_PUB(off_, a_, b_)                                \
  do {                                                        \
    if (FLAGS) {                                              \
      publish_pay(xo, (off_), (a_), (b_));                    \
      rearm_pay(xr, (off_));                                  \
    } else {                                                  \
      X::publish(xo, (off_), tag, (a_), (b_));                \
    }                                                         \
  } while (0)
        const int xo_ = it_xoff[it];
        T* dgi = dgi_ptr[it];
        constexpr int M = NS ? NS : 1;
        if (CELL == CELL_GRU) {
          float dr0 = 0.f, dr1 = 0.f, dz0 = 0.f, dz1 = 0.f, dn0 = 0.f, dn1 = 0.f, dq0 = 0.f, dq1 = 0.f;
          car[it][0] = din0;
          car[it][1] = din1;
          if (act) {
            const float r0 = X::lo(sp[it][0]), r1 = X::hi(sp[it][0]), z0 = X::lo(sp[it][1 % M]), z1 = X::hi(sp[it][1 % M]);
            const float n0 = X::lo(sp[it][2 % M]), n1 = X::hi(sp[it][2 % M]), q0 = X::lo(sp[it][3 % M]), q1 = X::hi(sp[it][3 % M]);
            const float dh0 = X::lo(dout[it]) + din0, dh1 = X::hi(dout[it]) + din1;
            dn0 = dh0 * (1.f - z0) * (1.f - n0 * n0);
            dn1 = dh1 * (1.f - z1) * (1.f - n1 * n1);
            dz0 = dh0 * (X::lo(hp[it]) - n0) * z0 * (1.f - z0);
            dz1 = dh1 * (X::hi(hp[it]) - n1) * z1 * (1.f - z1);
            dr0 = dn0 * q0 * r0 * (1.f - r0);
            dr1 = dn1 * q1 * r1 * (1.f - r1);
            dq0 = dn0 * r0;
            dq1 = dn1 * r1;
            car[it][0] = dh0 * z0;
            car[it][1] = dh1 * z1;
          }
          if (dead) dr0 = dr1 = QNAN;
          DS2Q_PUB(xo_, dr0, dr1);
          DS2Q_PUB(xo_ + GATE_BYTES, dz0, dz1);
          DS2Q_PUB(xo_ + 2 * GATE_BYTES, dq0, dq1);
          X::st(dgi, dr0, dr1);
          X::st(dgi + H, dz0, dz1);
          X::st(dgi + 2 * H, dn0, dn1);
          X::st(dgh_ptr[it], dq0, dq1);
          bsum[it][0][0] += X::rnd(dr0); bsum[it][0][1] += X::rnd(dr1);
          bsum[it][1 % NB][0] += X::rnd(dz0); bsum[it][1 % NB][1] += X::rnd(dz1);
          bsum[it][2 % NB][0] += X::rnd(dn0); bsum[it][2 % NB][1] += X::rnd(dn1);
          bsum[it][3 % NB][0] += X::rnd(dq0); bsum[it][3 % NB][1] += X::rnd(dq1);
        } else if (CELL == CELL_LSTM) {
          float di0 = 0.f, di1 = 0.f, df0 = 0.f, df1 = 0.f, dg0 = 0.f, dg1 = 0.f, do0 = 0.f, do1 = 0.f;
          car[it][0] = din0;
          car[it][1] = din1;
          if (act) {
            const float i0 = X::lo(sp[it][0]), i1 = X::hi(sp[it][0]), f0 = X::lo(sp[it][1 % M]), f1 = X::hi(sp[it][1 % M]);
            const float g0 = X::lo(sp[it][2 % M]), g1 = X::hi(sp[it][2 % M]), o0 = X::lo(sp[it][3 % M]), o1 = X::hi(sp[it][3 % M]);
            const float tc0 = X::tnh(X::lo(sp[it][4 % M])), tc1 = X::tnh(X::hi(sp[it][4 % M]));
            const float dh0 = X::lo(dout[it]) + din0, dh1 = X::hi(dout[it]) + din1;
            const float dcn0 = dc[it][0] + dh0 * o0 * (1.f - tc0 * tc0), dcn1 = dc[it][1] + dh1 * o1 * (1.f - tc1 * tc1);
            di0 = dcn0 * g0 * i0 * (1.f - i0);
            di1 = dcn1 * g1 * i1 * (1.f - i1);
            df0 = dcn0 * X::lo(cp[it]) * f0 * (1.f - f0);
            df1 = dcn1 * X::hi(cp[it]) * f1 * (1.f - f1);
            dg0 = dcn0 * i0 * (1.f - g0 * g0);
            dg1 = dcn1 * i1 * (1.f - g1 * g1);
            do0 = dh0 * tc0 * o0 * (1.f - o0);
            do1 = dh1 * tc1 * o1 * (1.f - o1);
            car[it][0] = car[it][1] = 0.f;
            dc[it][0] = dcn0 * f0;
            dc[it][1] = dcn1 * f1;
          }
          if (dead) di0 = di1 = QNAN;
          DS2Q_PUB(xo_, di0, di1);
          DS2Q_PUB(xo_ + GATE_BYTES, df0, df1);
          DS2Q_PUB(xo_ + 2 * GATE_BYTES, dg0, dg1);
          DS2Q_PUB(xo_ + 3 * GATE_BYTES, do0, do1);
          X::st(dgi, di0, di1);
          X::st(dgi + H, df0, df1);
          X::st(dgi + 2 * H, dg0, dg1);
          X::st(dgi + 3 * H, do0, do1);
          bsum[it][0][0] += X::rnd(di0); bsum[it][0][1] += X::rnd(di1);
          bsum[it][1 % NB][0] += X::rnd(df0); bsum[it][1 % NB][1] += X::rnd(df1);
          bsum[it][2 % NB][0] += X::rnd(dg0); bsum[it][2 % NB][1] += X::rnd(dg1);
          bsum[it][3 % NB][0] += X::rnd(do0); bsum[it][3 % NB][1] += X::rnd(do1);
        } else {
          float dg0 = 0.f, dg1 = 0.f;
          car[it][0] = din0;
          car[it][1] = din1;
          if (act) {
            const float h0v = X::lo(hp[it]), h1v = X::hi(hp[it]);
            dg0 = (X::lo(dout[it]) + din0) * (1.f - h0v * h0v);
            dg1 = (X::hi(dout[it]) + din1) * (1.f - h1v * h1v);
            car[it][0] = car[it][1] = 0.f;
          }
          if (dead) dg0 = dg1 = QNAN;
          DS2Q_PUB(xo_, dg0, dg1);
          X::st(dgi, dg0, dg1);
          bsum[it][0][0] += X::rnd(dg0); bsum[it][0][1] += X::rnd(dg1);
        }
      }
      do_ptr[it] += dstep * N * H;
      if (NS) sv_ptr[it] += dstep * N * NSH_;
      hs_ptr[it] += dstep * N * H;
      dgi_ptr[it] += dstep * N * ldgi;
      if (CELL == CELL_GRU) dgh_ptr[it] += dstep * N * H;
    }
#undef DS2Q_PUB
  }
  if (a.dBacc) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      if (on[it]) {
        float* bo = a.dBacc + ((long)d * N + it_n[it]) * NB * H + it_j[it];
#pragma unroll
        for (int g = 0; g < NB; ++g) *reinterpret_cast<float2*>(bo + (long)g * H) = make_float2(bsum[it][g][0], bsum[it][g][1]);
      }
    }
  }
}

template <int CELL, typename T, int H, int MT, int SP>
int launch2_sp(bool bwd, const QArgs& a, hipStream_t st) {
  constexpr int G = CellInfo<CELL>::G;
  const int P = H / 16;
  const size_t shm = (size_t)2 * 4 * MT * (bwd ? 1 : G) * 256 * sizeof(float);
  static bool attr[2][DS2_MAX_DEVICES];
  const void* fn = bwd ? (const void*)k_rnn_persist2_bwd<CELL, T, H, MT, SP> : (const void*)k_rnn_persist2_fwd<CELL, T, H, MT, SP>;
  if (ds2_first_use_on_device(attr[bwd ? 1 : 0])) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  if (bwd)
    hipLaunchKernelGGL((k_rnn_persist2_bwd<CELL, T, H, MT, SP>), dim3(a.NG * P), dim3(256), shm, st, a);
  else
    hipLaunchKernelGGL((k_rnn_persist2_fwd<CELL, T, H, MT, SP>), dim3(a.NG * P), dim3(256), shm, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// MT == 1: the lanes of absent sample rows share the gather loads (SP = 4 up to 4 samples per group, 2 up to 8)
template <int CELL, typename T, int H, int MT>
int launch2(bool bwd, const QArgs& a, hipStream_t st) {
  if constexpr (MT == 1) {
    const int ns = (a.N + a.gpd - 1) / a.gpd;
    if (ns <= 4) return launch2_sp<CELL, T, H, 1, 4>(bwd, a, st);
    if (ns <= 8) return launch2_sp<CELL, T, H, 1, 2>(bwd, a, st);
  }
  return launch2_sp<CELL, T, H, MT, 1>(bwd, a, st);
}

}  // namespace ds2q
