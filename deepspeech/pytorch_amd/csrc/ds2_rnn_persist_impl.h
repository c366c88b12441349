// Persistent recurrent sweeps (forward and BPTT) of BatchRNN (reference model.py:94-102) for bf16 storage:
// ONE launch per layer and sweep, all T' dependent time steps inside the kernel, W_hh resident in registers.
//
// Why: the recurrence is 2*L*T' dependent steps per training step (7 510 for the LibriSpeech configuration); a kernel
// boundary per step costs >= 10 us in practice (ds2_rnn.hip: launch gap + cold dependent-load chain + W_hh re-streamed
// from L2), the arithmetic of one step is < 0.5 us of MFMA time.  Here the chip is cut into 8 independent GROUPS of 32
// workgroups (one workgroup per CU; group = blockIdx % 8, which the dispatcher places on one XCD -- a speed
// assumption only).  A group owns one direction and a slice of the minibatch (samples are independent, so groups never
// talk to each other); inside a group, workgroup p owns hidden units [p*U, (p+1)*U) of every gate and keeps its slice
// of W_hh (G*U rows x H, 192 KiB for GRU-1024) in VGPRs as ready-made MFMA B fragments for the whole sweep.
// Per time step a workgroup
//   1. gathers the group's h_{t-1} (samples x H, bf16) from the exchange buffer -- 8-byte {tag = step, 2 x bf16}
//      granules written write-through (sc1) by their producers and polled with sc1 loads: the data is the flag, no fence,
//      no barrier (MI355X guide section 6 G16, recipe R2).  Wave w gathers exactly the K-quarter it multiplies, straight
//      into MFMA A-fragment registers (no LDS staging);
//   2. multiplies it with its resident W slice (v_mfma_f32_16x16x32_bf16, samples are the M rows), K split over the 4
//      waves, partial sums to LDS (double-buffered by step parity: one barrier per step);
//   3. 2 hidden units per thread: adds the hoisted input projection (prefetched at the top of the step), gate math in
//      fp32 with the fp32 carried state in a register, packed-sequence masking, stores (h_t, saved gates), and publishes
//      its two new h values as one granule.
// The BPTT kernel has the same skeleton with W_hh^T resident (U rows x G*H), exchanging the gate gradients.
//
// Placement independence: any assignment of workgroups to CUs/XCDs is correct as long as all 256 workgroups are
// co-resident (grid == CU count, 1 wave per SIMD by register use); every spin is bounded and a timed-out workgroup
// raises *err, poisons its outputs with NaN and stops waiting, so the launch always terminates.
// Roofline: MFMA-shaped work (2*samples*G*H*H flops per step and direction) but latency-bound by construction; the
// measured quantity is us per time step (bench.py roofline object).
#pragma once
#include <type_traits>
#include <stdlib.h>

#include "ds2_common.h"

namespace ds2p {

typedef unsigned long long u64;

enum { CELL_GRU = 0, CELL_LSTM = 1, CELL_RNN = 2 };
constexpr int NGROUPS = 8;
constexpr int MAXS = 16;                 // samples per group (MFMA M tile)
constexpr unsigned SPIN_LIMIT = 4000000; // ~ seconds; then give up loudly
constexpr unsigned TAG_INIT = 0x40000000u;

template <int CELL>
struct CellInfo;
template <>
struct CellInfo<CELL_GRU> {
  static constexpr int G = 3, NS = 4;
};
template <>
struct CellInfo<CELL_LSTM> {
  static constexpr int G = 4, NS = 5;
};
template <>
struct CellInfo<CELL_RNN> {
  static constexpr int G = 1, NS = 0;
};

struct PArgs {
  int N, Tp, D, gpd;        // gpd = groups per direction (NGROUPS / D)
  const int* lens;          // [N]
  const bf16_t* W;          // fwd: W_hh [D][G*H][H]        bwd: W_hh^T [D][H][G*H]
  const float* bhh;         // [D][G*H]
  const bf16_t* GI;         // fwd: input projection [Tp*N][D*G*H]
  bf16_t* Hseq;             // h_t of direction d at Hseq + d*hseq_dstride + (t*N+n)*H (guard slots at t=-1 and t=Tp are zero)
  long hseq_dstride;
  bf16_t* S;                // saved planes [D][Tp][N][NS*H]
  const float* h0;          // [D][N][H] or null
  const float* c0;
  float* hn;                // [D][N][H] or null
  float* cn;
  const bf16_t* dOut;       // bwd: [Tp][N][H]
  bf16_t* dGI;              // bwd: [Tp*N][D*G*H]
  bf16_t* dGH;              // bwd, GRU only: dQ [D][Tp][N][H] = dn * r (the n-gate slot of d(W_hh h + b_hh); the r and z slots
                            // equal dGI's and are not stored twice)
  float* dBacc;             // bwd: [D][N][NB*H] per-sample sums over time of the gate gradients (NB planes: GRU dr,dz,dn,dq;
                            // LSTM di,df,dg,do; RNN dg) -- the bias gradients are their sums over the samples
  u64* xbuf;                // [NGROUPS][2][MAXS][X/2] granules (or four payload-only slots), filled with 0xFF bytes before the launch
  int* err;                 // device word, set to 1 on a spin time-out (sticky: the host reads it)
  int* lerr;                // per-LAUNCH word in the scratch (reset before the launch, raised == 1): lets the peers of a timed-out workgroup stop early
  u64* xcc;                 // [NGROUPS][32] start-up exchange of the workgroups' XCC ids, reset (0xFF bytes) before the launch
  unsigned startup_ms;      // per-launch budget of the start-up handshake (wall clock; ds2_persist_opts.startup_ms, never 0 here)
#ifdef DS2_PROBE            // tools/probe_rnn_persist.py builds its own library with -DDS2_PROBE; the shipping kernels carry none of it
  unsigned long long* dbg;  // [NGROUPS][8] cycle counters of workgroup 0 of each group
  unsigned long long* tl;   // [32 workgroups of group 0][TL_N steps][TL_K stamps]: s_memtime of wave 0 (see DS2_TL)
  int dbgmask;              // 1 skip GI/dOut/S prefetch loads, 2 skip output stores, 8 skip the gather (no exchange), 16 no L2 warm-up
#endif
};
#ifdef DS2_PROBE
#define DS2_PROBE_ONLY(...) __VA_ARGS__
#define DS2_DBG(a, bit) ((a).dbgmask & (bit))
// timeline of a few steps: stamps 0 step top, 1 gather issued, 2 gather + products done, 3 barrier passed, 4 publish issued, 5 step end
constexpr int TL_S0 = 300, TL_N = 8, TL_K = 6;
#define DS2_TL(k) do { if (grp == 0 && tid == 0 && s >= TL_S0 && s < TL_S0 + TL_N) tl_[s - TL_S0][k] = __builtin_readcyclecounter(); } while (0)
#else
#define DS2_PROBE_ONLY(...)
#define DS2_DBG(a, bit) 0
#define DS2_TL(k) do { } while (0)
#endif

__device__ __forceinline__ u64 g_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void g_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Publish one granule.  `local` (all workgroups of the group verified to sit on ONE XCD, see group_is_xcd_local): a plain
// 8-byte store, which stays in that XCD's L2 where the peers' sc1 (L1-bypassing) loads find it at L2 latency.  Otherwise the
// write-through (sc1) form that is visible chip-wide but drops the line from L2, so every reader pays a fabric round trip.
__device__ __forceinline__ void publish(u64* p, u64 v, bool local) {
  if (local)
    __builtin_nontemporal_store(v, p);   // lowers to a plain-policy `nt` store: line stays in this XCD's L2, no wait inserted
  else
    g_store(p, v);
}

// Start-up handshake (placement-independent sc1 protocol): every workgroup publishes its XCC id, waits for the 31 peers of
// its group and returns true iff all 32 ids are equal.  Every member evaluates the same 32 words, so the group agrees.
// Every 1024 polls of a mid-sweep wait: has a peer already given up (lerr[0] == 1), or has this wait outlived the launch's own spin
// budget (lerr[1]: all-ones from the scratch reset = none beyond SPIN_LIMIT; ds2_persist_opts.spin_limit lowers it per launch for
// fault-injection tests)?
__device__ __forceinline__ bool spin_check(int* lerr, unsigned spins) {
  return __hip_atomic_load(lerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1 ||
         spins > (unsigned)__hip_atomic_load(lerr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void raise_err(int* err, int* lerr) {
  __hip_atomic_fetch_max(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // max: never hides a start-up failure (code 2) -- the healthy
  __hip_atomic_store(lerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // groups of the same launch stop with code 1 when they see lerr
}
// The start-up handshake is where a sweep finds out that its workgroups are NOT all resident: every workgroup needs a whole CU (512
// registers per lane), so a kernel of another process on this GPU -- or a long kernel on another stream, e.g. an RCCL collective
// that waits for a late peer rank -- that holds CUs keeps some of the 256 from starting while the others wait here.  That wait has
// its own budget, a property of the LAUNCH (PArgs::startup_ms, measured on the 100 MHz wall clock, so it does not depend on the
// shader clock or on how long a poll takes): the host passes ~0.3 s for a single-process run (nothing legitimate holds CUs that
// long: fail fast and name the cause) and the process group's time-out under data parallelism (a sweep behind a collective must
// WAIT, as any stock kernel would queue) -- ops.persist_startup_ms().  Its own code in the sticky error word lets the host NAME the
// cause (ops.check_persistent_kernels) instead of reporting a generic time-out.  max: a later mid-sweep time-out of the same
// launch must not hide it.
__device__ __forceinline__ void raise_err_startup(int* err, int* lerr) {
  __hip_atomic_fetch_max(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(lerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every 1024 polls of a start-up wait: has a peer given up, or is the launch's start-up budget spent?
__device__ __forceinline__ bool startup_expired(int* lerr, u64 t0, unsigned startup_ms) {
  return __hip_atomic_load(lerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1 || wall_clock64() - t0 > (u64)startup_ms * 100000ull;
}
__device__ __forceinline__ bool group_is_xcd_local(u64* slots /* this group's [32] */, int p, int tid, int* err, int* lerr, unsigned startup_ms,
                                                   bool& dead) {
  __shared__ int s_local;
  if (tid < 64) {
    const unsigned my = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;   // HW_REG_XCC_ID[3:0]
    if (tid == 0) g_store(slots + p, (0x5ca1ab1eull << 32) | my);
    bool same = true;
    unsigned spins = 0;
    const u64 t0 = wall_clock64();
    for (;;) {
      u64 v = 0;
      if (tid < 32) v = g_load(slots + tid);
      const bool bad = tid < 32 && (unsigned)(v >> 32) != 0x5ca1ab1eu;
      if (!__any(bad)) {
        same = !(tid < 32) || ((unsigned)v == my);
        break;
      }
      if (((++spins) & 1023u) == 0 && startup_expired(lerr, t0, startup_ms)) {
        dead = true;
        raise_err_startup(err, lerr);
        same = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    const bool all_same = __all(same);
    if (tid == 0) s_local = all_same ? 1 : 0;
  }
  __syncthreads();
  return s_local != 0;
}
// Kernels without an XCC-id handshake (groups that span XCDs, the round-2 general kernels): ONE arrival word per launch in the
// scratch head (lerr[2], all-ones after the reset): every workgroup adds 1 and waits until all gridDim.x have, under the same
// start-up budget and with the same error code.  After it every wait of the sweep is between RESIDENT workgroups.
__device__ __forceinline__ void wait_all_resident(int* lerr, int tid, int* err, unsigned startup_ms, bool& dead) {
  __shared__ int s_dead;
  if (tid == 0) {
    int gone = 0;
    unsigned* arrive = reinterpret_cast<unsigned*>(lerr + 2);
    __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = gridDim.x - 1u;          // 0xFFFFFFFF + gridDim.x
    unsigned spins = 0;
    const u64 t0 = wall_clock64();
    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
      if (((++spins) & 1023u) == 0 && startup_expired(lerr, t0, startup_ms)) {
        raise_err_startup(err, lerr);
        gone = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    s_dead = gone;
  }
  __syncthreads();
  if (s_dead) dead = true;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) { return cvt_pk_bf16(lo, hi); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// fast gate math for the bf16 path: v_exp_f32 / v_rcp_f32 (1 ulp) instead of the full-precision expf + IEEE division of
// the fp32 parity kernels -- the gate phase runs on one wave per SIMD, where every VALU instruction is exposed latency.
__device__ __forceinline__ float fsigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

__device__ __forceinline__ uint32_t ror8(uint32_t v) {   // lane i <- lane i^8 within each row of 16 lanes
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, true);
}

// ---- L2 warm-up through the scalar cache --------------------------------------------------------------------------------------
// The gate phase's operands (hoisted input projection; dOut / saved planes / h_prev in BPTT) are 2-byte reads that walk through
// tensors of hundreds of MB: every one is an HBM miss, and a vector load that is still out holds up everything behind it in the
// wave (vmcnt retires in order: the gather's waits, and every re-poll).  A scalar load has its own counter and its own path to
// the same L2, so a wave touches the 64-byte runs its workgroup will read a few steps AHEAD with `s_load_dword` and never looks
// at the result: by the time the vector loads are issued the lines are L2 hits.
#ifndef DS2_L2_AHEAD
#define DS2_L2_AHEAD 3        // steps; a build flag for A/B runs (tools/ab_variants.py)
#endif
constexpr int L2_AHEAD = DS2_L2_AHEAD;
// A/B knobs of the 8-clip sweeps (tools/ab_sweeps.py, profiles/r05h_ab_sweep_timing.txt).  Forward: the input-projection loads of
// step s + 1 are issued at the END of step s (behind the publish, raw 16-bit values converted after the next gather) and a short pause
// stands in front of the gather where their wait used to pace it: GRU 1.345 -> 1.315-1.32 us per time step with a pause of 2-3 (no
// pause: 1.37-1.39, re-polls; 6: 1.38), LSTM unchanged.  BPTT: a pause in front of the gather costs its own length (1 / 2 / 4:
// +0.00 / +0.03 / +0.07 us per step).
#ifndef DS2_FWD_GI_AHEAD
#define DS2_FWD_GI_AHEAD 1
#endif
#ifndef DS2_FWD_SLEEP
#define DS2_FWD_SLEEP 2
#endif
// (rejected, profiles/r05h_ab_sweep_timing.txt: all 12 partial-sum reads of the forward gate phase issued before the first add -- 1.72
// instead of 1.33 us per step; the compiler's own order overlaps the first gate's exp / rcp chain with the remaining reads)
#ifndef DS2_BWD_SLEEP
#define DS2_BWD_SLEEP 0
#endif

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {   // uniform values may still live in vector registers (64-bit multiplies do)
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
         (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);          // the builtin returns int: no sign extension
}
__device__ __forceinline__ uint64_t uniform64(const void* p) { return uniform64(reinterpret_cast<uint64_t>(p)); }
// Every touch lands in ONE fixed scalar register, L2_SINK, that nothing else in these kernels uses: the write arrives
// asynchronously (tens to hundreds of cycles after the instruction), so the destination must not be a register the allocator may
// hand to something else in the meantime -- an ordinary asm output can be copied and released right after the statement.  The
// clobber keeps values from living in it ACROSS a touch; build.py additionally checks the generated assembly of these files: the
// only instructions that may name the register are the touches themselves.  `addr` must be wave-uniform (uniform64) and 4-byte
// aligned.  l2_touch_retire() (end of the step: the loads have long returned, no stall) bounds the number in flight.
#define DS2_L2_SINK "s101"
__device__ __forceinline__ void l2_touch(uint64_t addr) {
  asm volatile("s_load_dword " DS2_L2_SINK ", %0, 0x0" : : "s"(addr) : "memory", DS2_L2_SINK);
}
__device__ __forceinline__ void l2_touch_retire() { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory", DS2_L2_SINK); }
struct NoTouch {
  __device__ __forceinline__ void operator()() const {}
};

constexpr int chunk_ksteps(int KS, int SPLIT) {
#ifdef DS2_CHUNK
  return (KS % DS2_CHUNK == 0 && DS2_CHUNK % SPLIT == 0) ? DS2_CHUNK : KS;
#else
  // measured on cfg3 (tools/probe_rnn_persist.py): forward (KS = 8) 2.73 us/step with chunks of 4 vs 2.80 with 8; BPTT (KS = 24)
  // 3.65 with chunks of 8 vs 3.68 (6, 12), 4.04 (4), 4.29 (24): several small polls overlap better with the products of
  // the previous chunk and keep the register footprint low enough for other kernels to share the CU
  return KS <= 8 ? (KS / 2 >= SPLIT && KS % 2 == 0 ? KS / 2 : KS) : 8;
#endif
}

// Exchange buffer layout (per group and parity): the MFMA A-fragment order, so that every gather instruction of a wave
// reads contiguous memory:  [k-step (K/32)][q (2)][lq (4)][row (NROWS)] x 16 bytes, a 16-byte unit = the two granules
// {k, k+1}, {k+2, k+3} of one sample row with k = 32*kstep + 8*lq + 4*q.  NROWS = 8 (SPLIT == 2) or 16.
template <int NROWS>
__device__ __forceinline__ int xunit_bytes(int kstep, int q, int lq, int row) { return (((kstep * 2 + q) * 4 + lq) * NROWS + row) * 16; }
// byte offset of the granule that carries elements (k, k+1), k even, of sample row `row`
template <int NROWS>
__device__ __forceinline__ int xgranule_bytes(int k, int row) {
  const int kk = k & 31;
  return xunit_bytes<NROWS>(k >> 5, (kk & 7) >> 2, kk >> 3, row) + ((kk & 3) >> 1) * 8;
}

// Gathers this wave's K-quarter of the exchanged vector (granules tagged `epoch`) and multiplies it with the resident
// fragments: acc[tile] += A(samples x K-quarter) * w[tile](16 rows x K-quarter)^T.
// Loads are 16-byte sc1 buffer loads (2 granules each: the sweep is priced per load INSTRUCTION, not per byte).  With
// SPLIT == 2 (at most 8 samples in the group) the otherwise idle upper 8 lanes of every 16-lane row fetch the second
// half of each chunk's k-steps for sample (lane & 7) and hand it over with a DPP row rotate: half the loads per lane.
// A rows >= the sample count may hold anything: they only feed D rows that nobody reads.
// The MFMAs run SPECULATIVELY on whatever the loads returned, k-step by k-step as the data lands; the tags are checked
// afterwards and a (rare) miss restores the accumulators and repeats the chunk.
// kstep0 = first k-step of this wave's K-quarter.
template <int TILES, int KS, int SPLIT>
__device__ __forceinline__ void gather_mma(ds2_f32x4 (&acc)[TILES], const uint4 (&w)[TILES][KS], __amdgpu_buffer_rsrc_t rsrc,
                                           int par_off, int kstep0, int lq, int srow, int half, bool need, unsigned epoch,
                                           int* err, int* lerr, bool& dead, unsigned& rounds) {
  constexpr int CH = chunk_ksteps(KS, SPLIT);
  constexpr int PER = CH / SPLIT;   // k-steps a lane loads per chunk
  constexpr int NROWS = SPLIT == 2 ? 8 : 16;
  static_assert(KS % CH == 0 && CH % SPLIT == 0, "k-steps per wave must tile into poll chunks");
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#pragma unroll
  for (int c = 0; c < KS / CH; ++c) {
    u32x4 v[PER][2];
    // products of this chunk: straight into `acc` when the K-quarter is a single chunk (the caller passes zeros)
    ds2_f32x4 part_[KS == CH ? 1 : TILES];
    ds2_f32x4 (&part)[TILES] = *reinterpret_cast<ds2_f32x4 (*)[TILES]>(KS == CH ? &acc[0] : &part_[0]);
    // straight-line helpers (the retry path below repeats them; keeping the MFMAs out of the poll loop keeps the
    // register allocator away from spilling the resident fragments)
#define DS2_GATHER_LOAD()                                                                                              \
    if (need) {                                                                                                        \
      _Pragma("unroll") for (int i = 0; i < PER; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q)                    \
          v[i][q] = __builtin_amdgcn_raw_buffer_load_b128(                                                              \
              rsrc, par_off + xunit_bytes<NROWS>(kstep0 + c * CH + half * PER + i, q, lq, srow), 0, 16 /* sc1 */);      \
    }
#define DS2_GATHER_MMA()                                                                                               \
    _Pragma("unroll") for (int t = 0; t < TILES; ++t) part[t] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};                        \
    _Pragma("unroll") for (int k = 0; k < CH; ++k) {                                                                   \
      const int i = k % PER;                                                                                           \
      uint4 a = make_uint4(0, 0, 0, 0);                                                                                \
      if (need) a = make_uint4(v[i][0][0], v[i][0][2], v[i][1][0], v[i][1][2]);                                        \
      if (SPLIT == 2 && k >= PER) a = make_uint4(ror8(a.x), ror8(a.y), ror8(a.z), ror8(a.w));                          \
      _Pragma("unroll") for (int t = 0; t < TILES; ++t) Mma<bf16_t>::mma16(part[t], a, w[t][c * CH + k]);              \
    }
#define DS2_GATHER_CHECK(bad)                                                                                          \
    bool bad = false;                                                                                                  \
    if (need) {                                                                                                        \
      _Pragma("unroll") for (int i = 0; i < PER; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q)                    \
          bad |= (v[i][q][1] != epoch) | (v[i][q][3] != epoch);                                                        \
    }
    DS2_GATHER_LOAD()
    DS2_GATHER_MMA()          // speculative: runs k-step by k-step as the loads land
    DS2_GATHER_CHECK(bad0)
    if (__any(bad0) && !dead) {   // some granule was not there yet: one more speculative pass (products overlap the reload)
      __builtin_amdgcn_s_sleep(1);
      ++rounds;
      DS2_GATHER_LOAD()
      DS2_GATHER_MMA()
      DS2_GATHER_CHECK(bad1)
      if (__any(bad1)) {          // still not: poll without products until everything is there, then multiply
        unsigned spins = 0;
        for (;;) {
          __builtin_amdgcn_s_sleep(1);
          ++rounds;
          DS2_GATHER_LOAD()
          DS2_GATHER_CHECK(bad2)
          if (!__any(bad2)) break;
          // give up after the spin limit, or early when a peer of THIS launch already did (the per-launch word lives in the
          // scratch that is reset before every launch: a time-out of an earlier launch never shortens this one's patience)
          if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && spin_check(lerr, spins))) {
            dead = true;
            raise_err(err, lerr);
            break;
          }
        }
        DS2_GATHER_MMA()
      }
    }
#undef DS2_GATHER_LOAD
#undef DS2_GATHER_MMA
#undef DS2_GATHER_CHECK
    if (KS != CH) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) acc[t] += part[t];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Tag-free exchange (groups of <= 8 samples: k_rnn_persist_fwd4 / bwd4).  The tagged granules spend half of every gathered byte
// -- and half of the load instructions, which is what a step is priced in -- on tags.  Here the payload alone travels:
//   * the buffer is FOUR slots (step e publishes into slot e & 3), each in MFMA A-fragment order
//     [k-step][lq (4)][sample row (8)] x 16 bytes = the 8 bf16 k-values 32*kstep + 8*lq .. +7 of one row: ONE 16-byte load is
//     one A fragment;
//   * "not there yet" is the dword 0xFFFFFFFF (two bf16 NaNs with all mantissa bits set, which no published pair is: a pair
//     that would be is published as two canonical NaNs).  The host fills the buffer with 0xFF bytes; a dword is written by
//     exactly one thread with one 4-byte store, so it is either the sentinel or complete data;
//   * every publisher re-arms its own dwords: at step s, after its gather, it writes the sentinel into slot (s + 2) & 3.
//     That slot last held step s-2's data, which every peer finished reading before it published step s-1's (and those
//     publishes are what this workgroup's gather of step s just consumed).  The re-arm is acknowledged before this wave
//     publishes step s+1 (its gather of step s+1 waits on vmcnt, which retires in order), and a peer polls the slot for step
//     s+2's data only after it has seen this workgroup's step s+1 data: it can never read the stale step s-2 values.
// ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t XSENT = 0xffffffffu;
constexpr int chunk_ksteps_tf(int KS) {
#ifdef DS2_CHUNK
  return (KS % DS2_CHUNK == 0 && DS2_CHUNK % 2 == 0) ? DS2_CHUNK : KS;
#else
  // measured on cfg3 inside the training step (tools/bench_with_lib.py, us per time step forward / BPTT): chunks of 2: 2.55 / 5.38,
  // 4: 2.18 / 4.04, 6: - / 3.66, 8: 1.90 / 3.51, 12: - / 3.41, 24: - / 2.85 -- but the 24-k-step form holds 409 registers and
  // locks the weight-gradient GEMMs out of the CU (116 registers: they then run after the sweep, +2 ms per step), 12 holds 340
  // round 3 (nothing co-resident any more, L2-warm operands): 24 in one chunk again -- 2.19 us per BPTT step against 1.81-1.95 with 12
  return KS <= 8 ? KS : (KS % 12 == 0 ? 12 : 8);
#endif
}
__device__ __forceinline__ int xtf_unit_bytes(int kstep, int lq, int row) { return ((kstep * 4 + lq) * 8 + row) * 16; }
// byte offset of the dword that carries elements (k, k+1), k even, of sample row `row`
__device__ __forceinline__ int xtf_pair_bytes(int k, int row) {
  const int kk = k & 31;
  return xtf_unit_bytes(k >> 5, kk >> 3, row) + ((kk & 7) >> 1) * 4;
}
__device__ __forceinline__ uint32_t xtf_word(uint32_t pk) { return pk == XSENT ? 0x7fc07fc0u : pk; }
__device__ __forceinline__ void publish32(void* p, uint32_t v, bool local) {
  if (local)
    __builtin_nontemporal_store(v, (uint32_t*)p);
  else
    __hip_atomic_store((uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- 8 clips on all 16 MFMA rows: the structured-sparse form (round 5) ---------------------------------------------------------
// A group of <= 8 clips fills half of the 16 rows of a dense 16x16x32 tile: half of every product of the sweeps was padding.
// v_smfmac_f32_16x16x64_bf16 multiplies a 2:4-sparse A (two non-zeros in every four consecutive k, their positions in an index word)
// with a dense B over K = 64 at the cost of a dense K = 32 instruction.  Rows s and s + 8 of the tile both belong to clip s: row s
// carries its k = 0, 1 (mod 4) elements, row s + 8 its k = 2, 3 (mod 4) ones -- each row is 2:4 sparse by construction (index words
// 0x4444 / 0xEEEE), and D[s] + D[s + 8] is the full dot product: 24 instead of 48 matrix instructions per wave and step.
// (Layout of the operands measured by tools/probe_smfmac.py, profiles/r05a_smfmac.txt: lane 16*lq + i holds A row i, compressed
// elements 8*lq .. 8*lq + 7 = logical k 16*lq + 4*(e/2) + index(e); B column i as TWO dense 16x16x32 fragments, elements e = 0..7:
// k = 8*lq + e, elements 8..15: k = 32 + 8*lq + (e - 8) -- the weights stay in the registers of the dense kernels, fragment pairs
// (2*kb, 2*kb + 1) feed one instruction.)
// Exchange slot: [k-block (K/64)][lq (4)][tile row (16)] x 16 bytes; the unit of (kb, lq, row) = that lane's compressed A fragment =
// the four pairs k = 64*kb + 16*lq + 4*g + 2*(row >> 3) + {0, 1}, g = 0..3, of clip row & 7: ONE 16-byte load per lane and k-block.
typedef __attribute__((ext_vector_type(16))) __bf16 ds2_bf16x16;
__device__ __forceinline__ int xsp_unit_bytes(int kb, int lq, int row16) { return ((kb * 4 + lq) * 16 + row16) * 16; }
// byte offset of the dword that carries elements (k, k+1), k even, of clip `row`
__device__ __forceinline__ int xsp_pair_bytes(int k, int row) {
  const int kk = k & 63, r = kk & 15;
  return xsp_unit_bytes(k >> 6, kk >> 4, row + 8 * ((r & 3) >> 1)) + (r >> 2) * 4;
}
// The sparse instruction has no C operand: the sums of a chunk start from zeros IN the destination.  Left to itself the compiler keeps
// one zero tuple in accumulation registers, multiplies the first k-block there and copies the result out (s_nop + 8 reads in front of
// the second k-block); zeros it must produce in vector registers make the whole chain run in place.
__device__ __forceinline__ ds2_f32x4 zero_in_vgprs() {
  ds2_f32x4 z;
  asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z[0]), "=v"(z[1]), "=v"(z[2]), "=v"(z[3]));
  return z;
}
// A/B VARIANT (-DDS2_PREADD, round 6; measured and not kept, profiles/r06g_ab_preadd.txt): the two tile rows of a clip (lanes l and
// l + 32 of an accumulator register) are added BEFORE the partial sums go to LDS -- one v_permlane32_swap + one add per register, 24
// of each per wave and forward step -- so that a gate thread reads 4 instead of 8 partial sums per gate.
#ifndef DS2_PREADD
#define DS2_PREADD 0
#endif
template <int TILES>
__device__ __forceinline__ void preadd_rows(ds2_f32x4 (&acc)[TILES]) {
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[t][r]), __float_as_uint(acc[t][r]), false, false);
      acc[t][r] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
}
__device__ __forceinline__ void smma16(ds2_f32x4& acc, const uint4& a, const uint4& b0, const uint4& b1, int idx) {
  struct {
    uint4 lo, hi;
  } bb{b0, b1};
  acc = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(ds2_bf16x8, a), __builtin_bit_cast(ds2_bf16x16, bb), acc, idx, 0, 0);
}

// ---- 16-byte merged stores of the 8-samples-per-group kernels (-DDS2_QUAD_STORES: an A/B variant) ---------------------------------
// The even lanes 8m, 8m+2, 8m+4, 8m+6 of a wave hold four CONSECUTIVE dwords of one sample row -- in the exchange slot and in every
// stored plane (thread bits: e, pair within the 16-byte unit (2), sample row (3), lq (2)).  row4() collects them on lane 8m, which then
// issues ONE 16-byte store where four lanes issued four 4-byte ones (8 instead of 32 active lanes per store instruction).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4q;
template <int N>
__device__ __forceinline__ uint32_t row_plus(uint32_t v) {      // lane i of every 16-lane row <- lane (i + N) mod 16
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x120 + (16 - N), 0xf, 0xf, true);
}
__device__ __forceinline__ u32x4q row4(uint32_t v) {
  u32x4q r;
  r[0] = v;
  r[1] = row_plus<2>(v);
  r[2] = row_plus<4>(v);
  r[3] = row_plus<6>(v);
  return r;
}
__device__ __forceinline__ void publish128(__amdgpu_buffer_rsrc_t rsrc, int off, u32x4q v, bool local) {
  if (local)
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 2 /* nt: plain policy, the line stays in this XCD's L2 */);
  else
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 16 /* sc1: write-through */);
}

// SP: the structured-sparse form (see smma16): srow = the lane's tile row (0..15), `half` is unused, w[t][2*kb], w[t][2*kb + 1] (the
// dense fragments of k-steps 2*kb, 2*kb + 1) are the B operand of k-block kb, `spidx` the lane's index word.
template <int TILES, int KS, bool SP = false, typename F = NoTouch>
__device__ __forceinline__ void gather_mma_tf(ds2_f32x4 (&acc)[TILES], const uint4 (&w)[TILES][KS], __amdgpu_buffer_rsrc_t rsrc,
                                              int slot_off, int kstep0, int lq, int srow, int half, bool need, int* err, int* lerr,
                                              bool& dead, unsigned& rounds, F after_issue = F(), int spidx = 0) {
  constexpr int CH = chunk_ksteps_tf(KS);
  constexpr int PER = CH / 2;       // loads of a lane per chunk: half of the chunk's k-steps (the other half arrives by the DPP row
                                    // rotate), or (SP) its fragment of each of the chunk's CH / 2 k-blocks
  static_assert(KS % CH == 0 && CH % 2 == 0, "k-steps per wave must tile into poll chunks");
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#pragma unroll
  for (int c = 0; c < KS / CH; ++c) {
    u32x4 v[PER];
    ds2_f32x4 part_[KS == CH ? 1 : TILES];
    ds2_f32x4 (&part)[TILES] = *reinterpret_cast<ds2_f32x4 (*)[TILES]>(KS == CH ? &acc[0] : &part_[0]);
    // rows without a sample load from beyond the resource's range: the buffer check returns zeros, and there is no branch (a
    // conditional block would end in the copies of its first result, i.e. in a wait, ahead of after_issue)
    // (round 5, measured and rejected: the slot and the k-step as the instruction's SCALAR offset, as in the general kernels -- 36
    // fewer VALU instructions per BPTT step, and 1.34 / 1.65 instead of 1.29 / 1.56 us per time step, profiles/r05h_ab_sweep_timing.txt)
#define DS2_TF_LOAD(FIRST)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < PER; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(                      \
        rsrc, need ? slot_off + (SP ? xsp_unit_bytes((kstep0 + c * CH) / 2 + i, lq, srow)                              \
                                    : xtf_unit_bytes(kstep0 + c * CH + half * PER + i, lq, srow)) : 0x7ffffff0, 0, 16 /* sc1 */); \
    if (FIRST && c == 0 && !std::is_same<F, NoTouch>::value) {                                                         \
      after_issue();                     /* scalar work that hides under the first round trip (l2_touch) */            \
      __builtin_amdgcn_sched_barrier(0); /* nothing that waits for the loads may be scheduled above it   */            \
    }
#define DS2_TF_MMA()                                                                                                   \
    _Pragma("unroll") for (int t = 0; t < TILES; ++t) part[t] = (SP && KS != CH) ? zero_in_vgprs() : ds2_f32x4{0.f, 0.f, 0.f, 0.f}; \
    if (SP) {                                                                                                          \
      _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                                                \
        const uint4 a = make_uint4(v[i][0], v[i][1], v[i][2], v[i][3]);                                                \
        _Pragma("unroll") for (int t = 0; t < TILES; ++t)                                                              \
            smma16(part[t], a, w[t][c * CH + 2 * i], w[t][c * CH + 2 * i + 1], spidx);                                 \
      }                                                                                                                \
    } else {                                                                                                           \
      _Pragma("unroll") for (int k = 0; k < CH; ++k) {                                                                 \
        const int i = k % PER;                                                                                         \
        uint4 a = make_uint4(v[i][0], v[i][1], v[i][2], v[i][3]);                                                      \
        if (k >= PER) a = make_uint4(ror8(a.x), ror8(a.y), ror8(a.z), ror8(a.w));                                      \
        _Pragma("unroll") for (int t = 0; t < TILES; ++t) Mma<bf16_t>::mma16(part[t], a, w[t][c * CH + k]);            \
      }                                                                                                                \
    }
#define DS2_TF_CHECK(bad)                                                                                              \
    bool bad = false;                                                                                                  \
    if (need) {                                                                                                        \
      uint32_t m = 0;                                                                                                  \
      _Pragma("unroll") for (int i = 0; i < PER; ++i) m = max(max(m, max(v[i][0], v[i][1])), max(v[i][2], v[i][3]));   \
      bad = m == XSENT;                                                                                                \
    }
    DS2_TF_LOAD(true)
    DS2_TF_MMA()              // speculative: runs k-step by k-step as the loads land
    DS2_TF_CHECK(bad0)
    if (__any(bad0) && !dead) {
      __builtin_amdgcn_s_sleep(1);
      ++rounds;
      DS2_TF_LOAD(false)
      DS2_TF_MMA()
      DS2_TF_CHECK(bad1)
      if (__any(bad1)) {
        unsigned spins = 0;
        for (;;) {
          __builtin_amdgcn_s_sleep(1);
          ++rounds;
          DS2_TF_LOAD(false)
          DS2_TF_CHECK(bad2)
          if (!__any(bad2)) break;
          if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && spin_check(lerr, spins))) {
            dead = true;
            raise_err(err, lerr);
            break;
          }
        }
        DS2_TF_MMA()
      }
    }
#undef DS2_TF_LOAD
#undef DS2_TF_MMA
#undef DS2_TF_CHECK
    if (KS != CH) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) acc[t] += part[t];
    }
  }
}

template <int TILES>
__device__ __forceinline__ void store_partials(float* part, const ds2_f32x4 (&acc)[TILES], int wave, int lane) {
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[((wave * TILES + t) * 16 + mma16_row(r, lane)) * 16 + (lane & 15)] = acc[t][r];
}
// The 8-sample kernels' form: a tile is stored [col][row] with a column stride of PT_COL floats, so that the four rows a lane holds
// of an accumulator are ONE 16-byte store (6 instead of 24 LDS instructions per forward step); 20 = 16 + 4: the 16 lanes of a row
// group then cover the 64 banks exactly once (16-float columns would put lanes li, li+4, li+8, li+12 on the same banks).
constexpr int PT_COL = 20, PT_TILE = 16 * PT_COL;
template <int TILES>
__device__ __forceinline__ void store_partials_t(float* part, const ds2_f32x4 (&acc)[TILES], int wave, int lane) {
#pragma unroll
  for (int t = 0; t < TILES; ++t)
    *reinterpret_cast<ds2_f32x4*>(part + (wave * TILES + t) * PT_TILE + (lane & 15) * PT_COL + 4 * (lane >> 4)) = acc[t];
}
__device__ __forceinline__ int partial_t_index(int tile, int row, int col) { return tile * PT_TILE + col * PT_COL + row; }
// sum over the 4 waves of the two adjacent columns (col, col+1) of row `row` of tile `t`
template <int TILES>
__device__ __forceinline__ float2 load_partials(const float* part, int t, int row, int col) {
  float2 s = make_float2(0.f, 0.f);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((w * TILES + t) * 16 + row) * 16 + col);
    s.x += v.x;
    s.y += v.y;
  }
  return s;
}

// ------------------------------------------------------------------------------------------------------------------
// forward sweep
// ------------------------------------------------------------------------------------------------------------------
template <int CELL, int H, int P, int SPLIT>
__global__ void __launch_bounds__(256, 1) k_rnn_persist_fwd(PArgs a) {
  constexpr int G = CellInfo<CELL>::G, NS = CellInfo<CELL>::NS;
  constexpr int U = H / P;                 // hidden units owned by a workgroup
  constexpr int TILES = G * U / 16;        // 16-row MFMA tiles of the resident W slice
  constexpr int KS = H / 128;              // k-steps (of 32) per wave: K quarter = H/4
  constexpr int X2 = H / 2;                // granules per sample
  static_assert(U % 16 == 0 && H % 128 == 0, "unsupported hidden size for the persistent kernel");
  __shared__ __attribute__((aligned(16))) float part[2][4 * TILES * 256];
  __builtin_amdgcn_s_setprio(3);   // latency-critical: outrank any throughput kernel's waves that share the SIMD
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x % NGROUPS, p = blockIdx.x / NGROUPS;
  const int d = grp / a.gpd, slice = grp % a.gpd;
  const int N = a.N, Tp = a.Tp;
  const int Ns = (N - slice + a.gpd - 1) / a.gpd;          // samples n = slice + gpd*i, i < Ns
  const int li = lane & 15, lq = lane >> 4;
  constexpr long GH = (long)G * H;
  const long ldgi = (long)a.D * GH;

  // ---- resident W fragments: tile t row li = local row r = 16 t + li -> gate r / U, unit p*U + r % U
  uint4 w[TILES][KS];
  {
    const bf16_t* Wd = a.W + (long)d * GH * H;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const int r = 16 * t + li;
      const bf16_t* row = Wd + ((long)(r / U) * H + p * U + (r % U)) * H + wave * (H / 4) + lq * 8;
#pragma unroll
      for (int k = 0; k < KS; ++k) w[t][k] = *reinterpret_cast<const uint4*>(row + 32 * k);
    }
  }
  u64* xg = a.xbuf + (long)grp * 2 * MAXS * X2;
  const int srow = SPLIT == 2 ? (li & 7) : li, half = SPLIT == 2 ? (li >> 3) : 0;
  const bool need = srow < Ns;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, 2 * MAXS * X2 * 8, 0x00020000);
  constexpr int NROWS = SPLIT == 2 ? 8 : 16;
  constexpr int PAR_BYTES = MAXS * X2 * 8;     // one parity of the group's buffer

  // ---- gate-phase identity: thread -> (sample i, unit pair)
  constexpr int UP = U / 2;
  // Gate threads: the LAST NROWS*16 threads (with <= 8 samples they fill waves 2-3, so that waves 0-1 never issue stores:
  // vmcnt retires in order and a gather would otherwise wait for the wave's own previous publish to be acknowledged).
  // Thread bits = (pos, sample row, lq, q) in the order of the exchange layout, so a wave's publish is one contiguous
  // 512-byte (1 KiB) run; the unit pair of the thread is up = lq*4 + q*2 + pos.
  static_assert(UP == 16, "the gate-thread <-> exchange-layout map assumes 32 hidden units per workgroup");
  constexpr int NROWS_ = SPLIT == 2 ? 8 : 16, LR_ = SPLIT == 2 ? 3 : 4;
  const int gtid = tid - (256 - NROWS_ * UP);
  const int gi_i = gtid >= 0 ? ((gtid >> 1) & (NROWS_ - 1)) : MAXS;
  const int up = gtid >= 0 ? (((gtid >> (1 + LR_)) & 3) * 4 + ((gtid >> (3 + LR_)) & 1) * 2 + (gtid & 1)) : 0;
  const bool gate_thread = gtid >= 0 && gi_i < Ns;
  const int n = slice + a.gpd * gi_i;
  const int j = p * U + 2 * up;              // first of the two hidden units of this thread
  int len = 0;
  float hprev0 = 0.f, hprev1 = 0.f, cprev0 = 0.f, cprev1 = 0.f;
  float bh[G][2];
#pragma unroll
  for (int g = 0; g < G; ++g) bh[g][0] = bh[g][1] = 0.f;
  if (gate_thread) {
    len = a.lens[n];
    const long so = ((long)d * N + n) * H + j;
    if (a.h0) {
      hprev0 = a.h0[so];
      hprev1 = a.h0[so + 1];
    }
    if (CELL == CELL_LSTM && a.c0) {
      cprev0 = a.c0[so];
      cprev1 = a.c0[so + 1];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bh[g][0] = a.bhh[(long)d * GH + (long)g * H + j];
      bh[g][1] = a.bhh[(long)d * GH + (long)g * H + j + 1];
    }
  }
  // per-thread running pointers (advanced by a constant stride per step: 64-bit address arithmetic with multiplies inside
  // the gate phase costs hundreds of cycles on a single wave per SIMD)
  const long dstep = d == 0 ? 1 : -1;
  const int t_first = d == 0 ? 0 : Tp - 1;
  const int nn_ = gate_thread ? n : 0;
  const bf16_t* gi_ptr = a.GI + ((long)t_first * N + nn_) * ldgi + (long)d * GH + j;
  const long gi_stride = dstep * N * ldgi;
  constexpr long NSH_ = (long)(NS ? NS : 1) * H;
  bf16_t* sv_ptr = NS ? a.S + (((long)d * Tp + t_first) * N + nn_) * NSH_ + j : nullptr;
  const long sv_stride = dstep * N * NSH_;
  bf16_t* hs_ptr = a.Hseq + (long)d * a.hseq_dstride + ((long)t_first * N + nn_) * H + j;
  const long hs_stride = dstep * N * H;
  bool dead = false;
  const bool local = group_is_xcd_local(a.xcc + grp * 32, p, tid, a.err, a.lerr, a.startup_ms, dead);
  if (gate_thread) {         // zero guard slots of the state sequence at t = -1 and t = T' ("previous h" reads are unconditional)
    bf16_t* hb = a.Hseq + (long)d * a.hseq_dstride + (long)n * H + j;
    *reinterpret_cast<uint32_t*>(hb - (long)N * H) = 0u;
    *reinterpret_cast<uint32_t*>(hb + (long)Tp * N * H) = 0u;
  }
  if (gate_thread && a.h0)   // initial state as "step -1": parity 1, tag TAG_INIT
    publish((u64*)((char*)xg + PAR_BYTES + xgranule_bytes<NROWS>(j, gi_i)), ((u64)TAG_INIT << 32) | pack_bf16x2(hprev0, hprev1), local);
  unsigned rounds = 0;
  DS2_PROBE_ONLY(unsigned long long c_gather = 0, c_bar = 0, c_gate = 0;)

  for (int s = 0; s < Tp; ++s) {
    DS2_PROBE_ONLY(const unsigned long long t0 = __builtin_readcyclecounter();)
    const int t = d == 0 ? s : Tp - 1 - s;
    const int par = s & 1;
    // prefetch the hoisted input projection of this step
    uint32_t gi[G];
#pragma unroll
    for (int g = 0; g < G; ++g) gi[g] = 0;
    if (gate_thread && !DS2_DBG(a, 1)) {
      const bf16_t* gp = gi_ptr;
#pragma unroll
      for (int g = 0; g < G; ++g) gi[g] = *reinterpret_cast<const uint32_t*>(gp + (long)g * H);
    }
    ds2_f32x4 acc[TILES];
#pragma unroll
    for (int tt = 0; tt < TILES; ++tt) acc[tt] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
    if ((s > 0 || a.h0) && !DS2_DBG(a, 8))
      gather_mma<TILES, KS, SPLIT>(acc, w, rsrc, (par ^ 1) * PAR_BYTES, wave * KS, lq, srow, half, need, s > 0 ? (unsigned)s : TAG_INIT,
                                   a.err, a.lerr, dead, rounds);
    DS2_PROBE_ONLY(const unsigned long long t1 = __builtin_readcyclecounter();)
    store_partials<TILES>(part[par], acc, wave, lane);
    __syncthreads();
    DS2_PROBE_ONLY(const unsigned long long t2 = __builtin_readcyclecounter();)
    if (gate_thread) {
      const bool act = t < len;
      float hn0 = 0.f, hn1 = 0.f;     // emitted h_t (0 when inactive)
      const bool st_on = !DS2_DBG(a, 2);
      bf16_t* sv = sv_ptr;
      uint32_t pl[NS ? NS : 1];   // packed saved planes; stored AFTER the publish (the publish is what the peers wait for)
      float2 gh[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int r = g * U + 2 * up;
        gh[g] = load_partials<TILES>(part[par], r / 16, gi_i, r % 16);
      }
      if (CELL == CELL_GRU) {
        float r0 = 0.f, r1 = 0.f, z0 = 0.f, z1 = 0.f, n0 = 0.f, n1 = 0.f, q0 = 0.f, q1 = 0.f;
        if (act) {
          q0 = gh[2].x + bh[2][0];
          q1 = gh[2].y + bh[2][1];
          r0 = fsigmoid(bf_lo(gi[0]) + gh[0].x + bh[0][0]);
          r1 = fsigmoid(bf_hi(gi[0]) + gh[0].y + bh[0][1]);
          z0 = fsigmoid(bf_lo(gi[1]) + gh[1].x + bh[1][0]);
          z1 = fsigmoid(bf_hi(gi[1]) + gh[1].y + bh[1][1]);
          n0 = ftanh(bf_lo(gi[2]) + r0 * q0);
          n1 = ftanh(bf_hi(gi[2]) + r1 * q1);
          hn0 = (1.f - z0) * n0 + z0 * hprev0;
          hn1 = (1.f - z1) * n1 + z1 * hprev1;
          hprev0 = hn0;
          hprev1 = hn1;
        }
        pl[0] = pack_bf16x2(r0, r1);
        pl[1 % (NS ? NS : 1)] = pack_bf16x2(z0, z1);
        pl[2 % (NS ? NS : 1)] = pack_bf16x2(n0, n1);
        pl[3 % (NS ? NS : 1)] = pack_bf16x2(q0, q1);
      } else if (CELL == CELL_LSTM) {
        float i0 = 0.f, i1 = 0.f, f0 = 0.f, f1 = 0.f, g0 = 0.f, g1 = 0.f, o0 = 0.f, o1 = 0.f, c0 = 0.f, c1 = 0.f;
        if (act) {
          i0 = fsigmoid(bf_lo(gi[0]) + gh[0].x + bh[0][0]);
          i1 = fsigmoid(bf_hi(gi[0]) + gh[0].y + bh[0][1]);
          f0 = fsigmoid(bf_lo(gi[1]) + gh[1].x + bh[1][0]);
          f1 = fsigmoid(bf_hi(gi[1]) + gh[1].y + bh[1][1]);
          g0 = ftanh(bf_lo(gi[2]) + gh[2].x + bh[2][0]);
          g1 = ftanh(bf_hi(gi[2]) + gh[2].y + bh[2][1]);
          o0 = fsigmoid(bf_lo(gi[3 % G]) + gh[3 % G].x + bh[3 % G][0]);
          o1 = fsigmoid(bf_hi(gi[3 % G]) + gh[3 % G].y + bh[3 % G][1]);
          c0 = f0 * cprev0 + i0 * g0;
          c1 = f1 * cprev1 + i1 * g1;
          hn0 = o0 * ftanh(c0);
          hn1 = o1 * ftanh(c1);
          cprev0 = c0;
          cprev1 = c1;
          hprev0 = hn0;
          hprev1 = hn1;
        }
        pl[0] = pack_bf16x2(i0, i1);
        pl[1 % (NS ? NS : 1)] = pack_bf16x2(f0, f1);
        pl[2 % (NS ? NS : 1)] = pack_bf16x2(g0, g1);
        pl[3 % (NS ? NS : 1)] = pack_bf16x2(o0, o1);
        pl[4 % (NS ? NS : 1)] = pack_bf16x2(c0, c1);
      } else {
        if (act) {
          hn0 = ftanh(bf_lo(gi[0]) + gh[0].x + bh[0][0]);
          hn1 = ftanh(bf_hi(gi[0]) + gh[0].y + bh[0][1]);
          hprev0 = hn0;
          hprev1 = hn1;
        }
      }
      if (dead) hn0 = hn1 = hprev0 = hprev1 = __uint_as_float(0x7fc00000u);   // fail loudly downstream
      // publish the carried state first (inactive samples republish their unchanged state), then the bookkeeping stores
      publish((u64*)((char*)xg + par * PAR_BYTES + xgranule_bytes<NROWS>(j, gi_i)), ((u64)(unsigned)(s + 1) << 32) | pack_bf16x2(hprev0, hprev1),
              local);
      if (st_on) {
        *reinterpret_cast<uint32_t*>(hs_ptr) = pack_bf16x2(hn0, hn1);
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<uint32_t*>(sv + (long)q * H) = pl[q];
      }
    }
    gi_ptr += gi_stride;
    if (NS) sv_ptr += sv_stride;
    hs_ptr += hs_stride;
    DS2_PROBE_ONLY(const unsigned long long t3 = __builtin_readcyclecounter(); c_gather += t1 - t0; c_bar += t2 - t1; c_gate += t3 - t2;)
  }
#ifdef DS2_PROBE
  if (a.dbg && p == 0 && (tid == 0 || tid == 255)) {
    const int o = grp * 8 + (tid == 0 ? 0 : 4);
    a.dbg[o + 0] = c_gather;
    a.dbg[o + 1] = c_bar;
    a.dbg[o + 2] = c_gate;
    a.dbg[o + 3] = rounds;
  }
#endif
  if (gate_thread) {
    const long so = ((long)d * N + n) * H + j;
    if (a.hn) {
      a.hn[so] = hprev0;
      a.hn[so + 1] = hprev1;
    }
    if (CELL == CELL_LSTM && a.cn) {
      a.cn[so] = cprev0;
      a.cn[so + 1] = cprev1;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward sweep, <= 8 samples per group, gate phase on ALL four waves: one hidden unit per thread.
// The probe build shows where a step goes when the group is small (cfg3, cycles per step): gather + MFMA ~3700 (of which ~300 are
// MFMAs: the rest is the exchange), partial sums + barrier ~400, gate phase ~1200 -- and the gate phase is a single wave per SIMD
// executing ~130 dependent-latency-exposed instructions for its two hidden units.  Giving every thread ONE unit (256 items =
// 8 samples x 32 units) halves that chain; the two lanes of a unit pair meet through a DPP quad permute and the even lane
// publishes the granule and stores the packed planes, so the memory instructions stay 4 and 8 bytes wide.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_xor1(float v) {      // value of the neighbouring lane (lane ^ 1)
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
}

template <int CELL, int H, int P, bool SP>
__global__ void __launch_bounds__(256, 1) k_rnn_persist_fwd4(PArgs a) {
  constexpr int G = CellInfo<CELL>::G, NS = CellInfo<CELL>::NS;
  constexpr int U = H / P;
  constexpr int TILES = G * U / 16;
  constexpr int KS = H / 128;
  constexpr int X2 = H / 2;
  static_assert(U == 32 && H % 128 == 0, "the one-unit-per-thread gate map assumes 32 hidden units per workgroup");
  static_assert(!SP || H % 256 == 0, "a wave's K-quarter is whole k-blocks of 64");
  __shared__ __attribute__((aligned(16))) float part[2][4 * TILES * PT_TILE];
  DS2_PROBE_ONLY(__shared__ unsigned long long tl_[TL_N][TL_K];)
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x % NGROUPS, p = blockIdx.x / NGROUPS;
  const int d = grp / a.gpd, slice = grp % a.gpd;
  const int N = a.N, Tp = a.Tp;
  const int Ns = (N - slice + a.gpd - 1) / a.gpd;
  const int li = lane & 15, lq = lane >> 4;
  constexpr long GH = (long)G * H;
  const long ldgi = (long)a.D * GH;

  uint4 w[TILES][KS];
  {
    const bf16_t* Wd = a.W + (long)d * GH * H;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const int r = 16 * t + li;
      const bf16_t* row = Wd + ((long)(r / U) * H + p * U + (r % U)) * H + wave * (H / 4) + lq * 8;
#pragma unroll
      for (int k = 0; k < KS; ++k) w[t][k] = *reinterpret_cast<const uint4*>(row + 32 * k);   // SP: the same fragments (see smma16)
    }
  }
  // tag-free exchange: 4 slots of [H/32 k-steps][4][8 rows] x 16 bytes inside this group's share of the scratch
  constexpr int SLOT_BYTES = (H / 32) * 512;
  static_assert(4 * SLOT_BYTES <= 2 * MAXS * X2 * 8, "four payload slots fit where the two tagged parities lived");
  char* xg = (char*)(a.xbuf + (long)grp * 2 * MAXS * X2);
  const int srow = SP ? li : (li & 7), half = li >> 3;
  const bool need = (li & 7) < Ns;
  const int spidx = (li & 8) ? 0xEEEE : 0x4444;      // SP: rows 8..15 carry the k = 2, 3 (mod 4) elements
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, 4 * SLOT_BYTES, 0x00020000);

  // ---- gate identity: thread bits = (e: unit of the pair, pair within the 16-byte unit (2), sample row (3), lq (2)) -- the pair
  // index (tid >> 1) has the bit order of the exchange layout, so the even lanes of a wave publish one contiguous 128-byte run
  // (SP: (e, pair within the unit (2), tile-row half (1), sample row (3), 16-k block of the workgroup's 32 units (1)): two 64-byte runs)
  const int e = tid & 1, pid = tid >> 1;
  const int gi_i = SP ? (pid >> 3) & 7 : (pid >> 2) & 7;
  const int up = SP ? (pid >> 6) * 8 + (pid & 3) * 2 + ((pid >> 2) & 1) : ((pid >> 5) & 3) * 4 + (pid & 3);
  const bool gate_thread = gi_i < Ns;
  const int n = slice + a.gpd * gi_i;
  const int j = p * U + 2 * up;              // first unit of the pair
  const int ju = j + e;                      // this thread's unit
  int len = 0;
  float hprev = 0.f, cprev = 0.f;
  float bh[G];
#pragma unroll
  for (int g = 0; g < G; ++g) bh[g] = 0.f;
  if (gate_thread) {
    len = a.lens[n];
    const long so = ((long)d * N + n) * H + ju;
    if (a.h0) hprev = a.h0[so];
    if (CELL == CELL_LSTM && a.c0) cprev = a.c0[so];
#pragma unroll
    for (int g = 0; g < G; ++g) bh[g] = a.bhh[(long)d * GH + (long)g * H + ju];
  }
  const long dstep = d == 0 ? 1 : -1;
  const int t_first = d == 0 ? 0 : Tp - 1;
  const int nn_ = gate_thread ? n : 0;
  const uint16_t* gi_ptr = reinterpret_cast<const uint16_t*>(a.GI + ((long)t_first * N + nn_) * ldgi + (long)d * GH + ju);
  const long gi_stride = dstep * N * ldgi;
  constexpr long NSH_ = (long)(NS ? NS : 1) * H;
  bf16_t* sv_ptr = NS ? a.S + (((long)d * Tp + t_first) * N + nn_) * NSH_ + j : nullptr;     // pair base (even lane stores)
  const long sv_stride = dstep * N * NSH_;
  bf16_t* hs_ptr = a.Hseq + (long)d * a.hseq_dstride + ((long)t_first * N + nn_) * H + j;
  const long hs_stride = dstep * N * H;
  bool dead = false;
  const bool local = group_is_xcd_local(a.xcc + grp * 32, p, tid, a.err, a.lerr, a.startup_ms, dead);
  if (gate_thread && e == 0) {   // zero guard slots of the state sequence at t = -1 and t = T'
    bf16_t* hb = a.Hseq + (long)d * a.hseq_dstride + (long)n * H + j;
    *reinterpret_cast<uint32_t*>(hb - (long)N * H) = 0u;
    *reinterpret_cast<uint32_t*>(hb + (long)Tp * N * H) = 0u;
  }
  const int xoff = SP ? xsp_pair_bytes(j, gi_i) : xtf_pair_bytes(j, gi_i);     // this pair's dword inside a slot
  if (a.h0) {                    // initial state as "step -1": slot 3
    const float other = dpp_xor1(hprev);
    if (gate_thread && e == 0) publish32(xg + 3 * SLOT_BYTES + xoff, xtf_word(pack_bf16x2(hprev, other)), local);
  }
  unsigned rounds = 0;
  // partial-sum address of this thread's unit in tile coordinates: gate g -> local row g*U + 2*up + e
  int pidx[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int r = g * U + 2 * up + e;
    pidx[g] = partial_t_index(r / 16, gi_i, r % 16);
  }

  DS2_PROBE_ONLY(unsigned long long c_gather = 0, c_bar = 0, c_gate = 0;)
  // The hoisted input projection of a step is loaded at the top of the step, IN FRONT of the gather.  vmcnt retires in order,
  // so these loads (2-byte reads walking through a 295 MB tensor: HBM latency) hold the gather's waits up -- the probe build
  // runs at 1.66 instead of 1.97 us per step without them -- but moving them behind the gather (one or two steps ahead, three
  // rotating register sets) was slower in the training step, 2.05 vs 1.90: the gather then returns before every peer has
  // published (0.65 instead of 0.02 re-polls per step, each a full L2 round trip), i.e. the slow loads also pace the group.
  // Round 3: the lines are made L2 hits instead (l2_touch, L2_AHEAD steps ahead, issued while the gather's first round trip is
  // out): wave w touches samples 2w and 2w + 1; running scalar pointers, one add per step.
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  uint64_t pf_p[2];
  const int pf_s0 = min(L2_AHEAD, Tp - 1);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long nr = min(slice + a.gpd * max(0, min(2 * wv + i, Ns - 1)), N - 1);
    pf_p[i] = uniform64(a.GI + ((long)(d == 0 ? pf_s0 : Tp - 1 - pf_s0) * N + nr) * ldgi + (long)d * GH + p * U);
  }
  const uint64_t pf_step = uniform64((uint64_t)(dstep * N * ldgi * 2));
#if DS2_FWD_GI_AHEAD
  uint32_t gi_raw[G];
#pragma unroll
  for (int g = 0; g < G; ++g) gi_raw[g] = 0u;
  if (gate_thread) {
#pragma unroll
    for (int g = 0; g < G; ++g) gi_raw[g] = gi_ptr[(long)g * H];
  }
#endif
  for (int s = 0; s < Tp; ++s) {
    DS2_PROBE_ONLY(const unsigned long long t0 = __builtin_readcyclecounter();)
    DS2_TL(0);
    const int t = d == 0 ? s : Tp - 1 - s;
    const int par = s & 1;
    auto touch = [&]() {
      DS2_TL(1);
      if (DS2_DBG(a, 16)) return;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) l2_touch(pf_p[i] + (uint64_t)(g * H * 2));
      if (s + L2_AHEAD < Tp - 1) {
        pf_p[0] += pf_step;
        pf_p[1] += pf_step;
      }
    };
    float gi[G];
#pragma unroll
    for (int g = 0; g < G; ++g) gi[g] = 0.f;
#if !DS2_FWD_GI_AHEAD
    if (gate_thread && !DS2_DBG(a, 1)) {
#pragma unroll
      for (int g = 0; g < G; ++g) gi[g] = __uint_as_float((uint32_t)gi_ptr[(long)g * H] << 16);
    }
#endif
#if DS2_FWD_SLEEP
    if (s > 0) __builtin_amdgcn_s_sleep(DS2_FWD_SLEEP);
#endif
    ds2_f32x4 acc[TILES];
#pragma unroll
    for (int tt = 0; tt < TILES; ++tt) acc[tt] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0 || a.h0)
      gather_mma_tf<TILES, KS, SP>(acc, w, rsrc, ((s + 3) & 3) * SLOT_BYTES, wave * KS, lq, srow, half, need, a.err, a.lerr, dead, rounds, touch, spidx);
    else
      touch();
#if DS2_FWD_GI_AHEAD
    {   // loaded at the end of the step before (the gather's waits have drained them: vmcnt retires in order)
      if (s == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int g = 0; g < G; ++g) {
        gi[g] = __uint_as_float(gi_raw[g] << 16);
        asm volatile("" : "+v"(gi[g]));
      }
    }
#endif
    DS2_PROBE_ONLY(const unsigned long long t1 = __builtin_readcyclecounter();)
    DS2_TL(2);
    if (SP && DS2_PREADD) preadd_rows<TILES>(acc);
    store_partials_t<TILES>(part[par], acc, wave, lane);
    __syncthreads();
    DS2_PROBE_ONLY(const unsigned long long t2 = __builtin_readcyclecounter();)
    DS2_TL(3);
    float hn = 0.f;                 // emitted h_t (0 when inactive)
    float pl[NS ? NS : 1];
#pragma unroll
    for (int q = 0; q < (NS ? NS : 1); ++q) pl[q] = 0.f;
    if (gate_thread) {
      const bool act = t < len;
      float gh[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float* pp = part[par] + pidx[g];
        gh[g] = (pp[0] + pp[TILES * PT_TILE]) + (pp[2 * TILES * PT_TILE] + pp[3 * TILES * PT_TILE]);
        if (SP && !DS2_PREADD)      // the clip's second tile row (k = 2, 3 mod 4), 8 floats on in the [col][row] tile
          gh[g] += (pp[8] + pp[TILES * PT_TILE + 8]) + (pp[2 * TILES * PT_TILE + 8] + pp[3 * TILES * PT_TILE + 8]);
      }
      if (act) {
        if (CELL == CELL_GRU) {
          const float q = gh[2 % G] + bh[2 % G];
          const float r = fsigmoid(gi[0] + gh[0] + bh[0]);
          const float z = fsigmoid(gi[1 % G] + gh[1 % G] + bh[1 % G]);
          const float nn = ftanh(gi[2 % G] + r * q);
          hn = (1.f - z) * nn + z * hprev;
          hprev = hn;
          pl[0] = r; pl[1 % (NS ? NS : 1)] = z; pl[2 % (NS ? NS : 1)] = nn; pl[3 % (NS ? NS : 1)] = q;
        } else if (CELL == CELL_LSTM) {
          const float ig = fsigmoid(gi[0] + gh[0] + bh[0]);
          const float fg = fsigmoid(gi[1 % G] + gh[1 % G] + bh[1 % G]);
          const float gg = ftanh(gi[2 % G] + gh[2 % G] + bh[2 % G]);
          const float og = fsigmoid(gi[3 % G] + gh[3 % G] + bh[3 % G]);
          const float c = fg * cprev + ig * gg;
          hn = og * ftanh(c);
          cprev = c;
          hprev = hn;
          pl[0] = ig; pl[1 % (NS ? NS : 1)] = fg; pl[2 % (NS ? NS : 1)] = gg; pl[3 % (NS ? NS : 1)] = og; pl[4 % (NS ? NS : 1)] = c;
        } else {
          hn = ftanh(gi[0] + gh[0] + bh[0]);
          hprev = hn;
        }
      }
      if (dead) hn = hprev = __uint_as_float(0x7fc00000u);
    }
    // the pair meets: the even lane publishes the carried state of both units, then stores the packed planes
    {
      const float hp_o = dpp_xor1(hprev), hn_o = dpp_xor1(hn);
      float pl_o[NS ? NS : 1];
#pragma unroll
      for (int q = 0; q < (NS ? NS : 1); ++q) pl_o[q] = dpp_xor1(pl[q]);
#ifdef DS2_QUAD_STORES
      const u32x4q pubv = row4(xtf_word(pack_bf16x2(hprev, hp_o))), hsv = row4(pack_bf16x2(hn, hn_o));
      u32x4q plv[NS ? NS : 1];
#pragma unroll
      for (int q = 0; q < (NS ? NS : 1); ++q) plv[q] = row4(pack_bf16x2(pl[q], pl_o[q]));
      if (gate_thread && (tid & 7) == 0) {
        publish128(rsrc, (s & 3) * SLOT_BYTES + xoff, pubv, local);
        publish128(rsrc, ((s + 2) & 3) * SLOT_BYTES + xoff, u32x4q{XSENT, XSENT, XSENT, XSENT}, local);   // re-arm the slot of step s + 2
        *reinterpret_cast<u32x4q*>(hs_ptr) = hsv;
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<u32x4q*>(sv_ptr + (long)q * H) = plv[q];
      }
#else
      if (gate_thread && e == 0) {
        publish32(xg + (s & 3) * SLOT_BYTES + xoff, xtf_word(pack_bf16x2(hprev, hp_o)), local);
        publish32(xg + ((s + 2) & 3) * SLOT_BYTES + xoff, XSENT, local);        // re-arm the slot of step s + 2
        DS2_TL(4);
        *reinterpret_cast<uint32_t*>(hs_ptr) = pack_bf16x2(hn, hn_o);
#pragma unroll
        for (int q = 0; q < NS; ++q) *reinterpret_cast<uint32_t*>(sv_ptr + (long)q * H) = pack_bf16x2(pl[q], pl_o[q]);
      }
#endif
    }
    gi_ptr += gi_stride;
#if DS2_FWD_GI_AHEAD
    if (gate_thread && s + 1 < Tp) {
#pragma unroll
      for (int g = 0; g < G; ++g) gi_raw[g] = gi_ptr[(long)g * H];
    }
#endif
    if (NS) sv_ptr += sv_stride;
    hs_ptr += hs_stride;
    l2_touch_retire();
    DS2_PROBE_ONLY(const unsigned long long t3 = __builtin_readcyclecounter(); c_gather += t1 - t0; c_bar += t2 - t1; c_gate += t3 - t2;)
    DS2_TL(5);
  }
#ifdef DS2_PROBE
  if (a.tl && grp == 0 && tid == 0)
    for (int i = 0; i < TL_N * TL_K; ++i) a.tl[(long)p * TL_N * TL_K + i] = tl_[i / TL_K][i % TL_K];
  if (a.dbg && p == 0 && (tid == 0 || tid == 255)) {
    const int o = grp * 8 + (tid == 0 ? 0 : 4);
    a.dbg[o + 0] = c_gather;
    a.dbg[o + 1] = c_bar;
    a.dbg[o + 2] = c_gate;
    a.dbg[o + 3] = rounds;
  }
#endif
  (void)rounds;
  if (gate_thread) {
    const long so = ((long)d * N + n) * H + ju;
    if (a.hn) a.hn[so] = hprev;
    if (CELL == CELL_LSTM && a.cn) a.cn[so] = cprev;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BPTT sweep.  dh_t = dOut[t] + carry (elementwise part of dh from the step processed before) + dgh_{t'} * W_hh.
// ------------------------------------------------------------------------------------------------------------------
template <int CELL, int H, int P, int SPLIT>
__global__ void __launch_bounds__(256, 1) k_rnn_persist_bwd(PArgs a) {
  constexpr int G = CellInfo<CELL>::G, NS = CellInfo<CELL>::NS;
  constexpr int U = H / P;
  constexpr int TILES = U / 16;            // rows of W_hh^T owned: the U output units
  constexpr int KS = G * H / 128;          // K = G*H split over 4 waves, 32 per k-step
  constexpr int X2 = G * H / 2;
  static_assert(U % 16 == 0 && (G * H) % 128 == 0, "unsupported hidden size for the persistent kernel");
  __shared__ __attribute__((aligned(16))) float part[2][4 * TILES * 256];
  __builtin_amdgcn_s_setprio(3);   // latency-critical: outrank any throughput kernel's waves that share the SIMD
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x % NGROUPS, p = blockIdx.x / NGROUPS;
  const int d = grp / a.gpd, slice = grp % a.gpd;
  const int N = a.N, Tp = a.Tp;
  const int Ns = (N - slice + a.gpd - 1) / a.gpd;
  const int li = lane & 15, lq = lane >> 4;
  constexpr long GH = (long)G * H;
  const long ldgi = (long)a.D * GH;

  uint4 w[TILES][KS];
  {
    const bf16_t* WT = a.W + (long)d * H * GH;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const bf16_t* row = WT + (long)(p * U + 16 * t + li) * GH + wave * (GH / 4) + lq * 8;
#pragma unroll
      for (int k = 0; k < KS; ++k) w[t][k] = *reinterpret_cast<const uint4*>(row + 32 * k);
    }
  }
  u64* xg = a.xbuf + (long)grp * 2 * MAXS * X2;
  const int srow = SPLIT == 2 ? (li & 7) : li, half = SPLIT == 2 ? (li >> 3) : 0;
  const bool need = srow < Ns;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, 2 * MAXS * X2 * 8, 0x00020000);
  constexpr int NROWS = SPLIT == 2 ? 8 : 16;
  constexpr int PAR_BYTES = MAXS * X2 * 8;

  constexpr int UP = U / 2;
  // Gate threads: the LAST NROWS*16 threads (with <= 8 samples they fill waves 2-3, so that waves 0-1 never issue stores:
  // vmcnt retires in order and a gather would otherwise wait for the wave's own previous publish to be acknowledged).
  // Thread bits = (pos, sample row, lq, q) in the order of the exchange layout, so a wave's publish is one contiguous
  // 512-byte (1 KiB) run; the unit pair of the thread is up = lq*4 + q*2 + pos.
  static_assert(UP == 16, "the gate-thread <-> exchange-layout map assumes 32 hidden units per workgroup");
  constexpr int NROWS_ = SPLIT == 2 ? 8 : 16, LR_ = SPLIT == 2 ? 3 : 4;
  const int gtid = tid - (256 - NROWS_ * UP);
  const int gi_i = gtid >= 0 ? ((gtid >> 1) & (NROWS_ - 1)) : MAXS;
  const int up = gtid >= 0 ? (((gtid >> (1 + LR_)) & 3) * 4 + ((gtid >> (3 + LR_)) & 1) * 2 + (gtid & 1)) : 0;
  const bool gate_thread = gtid >= 0 && gi_i < Ns;
  const int n = slice + a.gpd * gi_i;
  const int j = p * U + 2 * up;
  int len = 0;
  if (gate_thread) len = a.lens[n];
  float car0 = 0.f, car1 = 0.f, dc0 = 0.f, dc1 = 0.f;   // carried dh (elementwise part) and dc
  const long dstep = d == 0 ? -1 : 1;                       // BPTT walks the direction's time axis backwards
  const int t_first = d == 0 ? Tp - 1 : 0;
  const int nn_ = gate_thread ? n : 0;
  constexpr long NSH_ = (long)(NS ? NS : 1) * H;
  const bf16_t* do_ptr = a.dOut + ((long)t_first * N + nn_) * H + j;
  const bf16_t* sv_ptr = NS ? a.S + (((long)d * Tp + t_first) * N + nn_) * NSH_ + j : nullptr;
  const bf16_t* hs_ptr = a.Hseq + (long)d * a.hseq_dstride + ((long)t_first * N + nn_) * H + j;   // h_t
  bf16_t* dgi_ptr = a.dGI + ((long)t_first * N + nn_) * ldgi + (long)d * GH + j;
  bf16_t* dgh_ptr = a.dGH ? a.dGH + (((long)d * Tp + t_first) * N + nn_) * H + j : nullptr;   // dQ
  constexpr int NB = CELL == CELL_GRU ? 4 : G;
  float bsum[NB][2];        // bias-gradient accumulators of this thread's (sample, unit pair): sums over the time steps
#pragma unroll
  for (int g = 0; g < NB; ++g) bsum[g][0] = bsum[g][1] = 0.f;
  const long prev_off = d == 0 ? -1 : 1;                    // previous step in FORWARD order of this direction
  bool dead = false;
  const bool local = group_is_xcd_local(a.xcc + grp * 32, p, tid, a.err, a.lerr, a.startup_ms, dead);
  unsigned rounds = 0;
  DS2_PROBE_ONLY(unsigned long long c_gather = 0, c_bar = 0, c_gate = 0;)

  for (int s = 0; s < Tp; ++s) {
    DS2_PROBE_ONLY(const unsigned long long t0 = __builtin_readcyclecounter();)
    const int t = d == 0 ? Tp - 1 - s : s;
    const int par = s & 1;
    // ---- prefetch everything the gate phase needs
    uint32_t dout = 0, sp[NS ? NS : 1], hp = 0, cp = 0;
#pragma unroll
    for (int q = 0; q < (NS ? NS : 1); ++q) sp[q] = 0;
    const int tprev = d == 0 ? t - 1 : t + 1;            // previous step in FORWARD order of this direction
    if (gate_thread && !DS2_DBG(a, 1)) {
      dout = *reinterpret_cast<const uint32_t*>(do_ptr);
      if (NS) {
        const bf16_t* sv = sv_ptr;
#pragma unroll
        for (int q = 0; q < NS; ++q) sp[q] = *reinterpret_cast<const uint32_t*>(sv + (long)q * H);
      }
      // h_{prev}: guard slots / inactive frames hold zeros, so the read is unconditional (tprev in [-1, Tp])
      hp = *reinterpret_cast<const uint32_t*>(hs_ptr + prev_off * N * H);
      if (CELL == CELL_LSTM) {
        const bool has_prev = d == 0 ? (t > 0) : (t + 1 < len);
        if (has_prev) cp = *reinterpret_cast<const uint32_t*>(sv_ptr + prev_off * N * NSH_ + 4 * H);
      }
      if (CELL == CELL_RNN) hp = *reinterpret_cast<const uint32_t*>(hs_ptr);
    }
    ds2_f32x4 acc[TILES];
#pragma unroll
    for (int tt = 0; tt < TILES; ++tt) acc[tt] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
    if (s > 0 && !DS2_DBG(a, 8))
      gather_mma<TILES, KS, SPLIT>(acc, w, rsrc, (par ^ 1) * PAR_BYTES, wave * KS, lq, srow, half, need, (unsigned)s, a.err, a.lerr, dead,
                                   rounds);
    DS2_PROBE_ONLY(const unsigned long long t1 = __builtin_readcyclecounter();)
    store_partials<TILES>(part[par], acc, wave, lane);
    __syncthreads();
    DS2_PROBE_ONLY(const unsigned long long t2 = __builtin_readcyclecounter();)
    if (gate_thread) {
      const bool act = t < len;
      const float2 mp = load_partials<TILES>(part[par], (2 * up) / 16, gi_i, (2 * up) % 16);
      const float din0 = car0 + mp.x, din1 = car1 + mp.y;
      bf16_t* dgi = dgi_ptr;
      char* xo = (char*)xg + par * PAR_BYTES;   // granule of gate g, units (j, j+1): element k = g*H + j
      const u64 tag = (u64)(unsigned)(s + 1) << 32;
      const bool st_on = !DS2_DBG(a, 2);
      if (CELL == CELL_GRU) {
        float dr0 = 0.f, dr1 = 0.f, dz0 = 0.f, dz1 = 0.f, dn0 = 0.f, dn1 = 0.f, dq0 = 0.f, dq1 = 0.f;
        car0 = din0;
        car1 = din1;
        if (act) {
          const float r0 = bf_lo(sp[0]), r1 = bf_hi(sp[0]), z0 = bf_lo(sp[1 % (NS ? NS : 1)]), z1 = bf_hi(sp[1 % (NS ? NS : 1)]);
          const float n0 = bf_lo(sp[2 % (NS ? NS : 1)]), n1 = bf_hi(sp[2 % (NS ? NS : 1)]);
          const float q0 = bf_lo(sp[3 % (NS ? NS : 1)]), q1 = bf_hi(sp[3 % (NS ? NS : 1)]);
          const float dh0 = bf_lo(dout) + din0, dh1 = bf_hi(dout) + din1;
          dn0 = dh0 * (1.f - z0) * (1.f - n0 * n0);
          dn1 = dh1 * (1.f - z1) * (1.f - n1 * n1);
          dz0 = dh0 * (bf_lo(hp) - n0) * z0 * (1.f - z0);
          dz1 = dh1 * (bf_hi(hp) - n1) * z1 * (1.f - z1);
          dr0 = dn0 * q0 * r0 * (1.f - r0);
          dr1 = dn1 * q1 * r1 * (1.f - r1);
          dq0 = dn0 * r0;
          dq1 = dn1 * r1;
          car0 = dh0 * z0;
          car1 = dh1 * z1;
        }
        if (dead) dr0 = dr1 = __uint_as_float(0x7fc00000u);
        const uint32_t pr = pack_bf16x2(dr0, dr1), pz = pack_bf16x2(dz0, dz1), pn = pack_bf16x2(dn0, dn1), pq = pack_bf16x2(dq0, dq1);
        publish((u64*)(xo + xgranule_bytes<NROWS>(j, gi_i)), tag | pr, local);       // peers wait for these: first
        publish((u64*)(xo + xgranule_bytes<NROWS>(H + j, gi_i)), tag | pz, local);
        publish((u64*)(xo + xgranule_bytes<NROWS>(2 * H + j, gi_i)), tag | pq, local);
        if (st_on) {
          *reinterpret_cast<uint32_t*>(dgi) = pr;
          *reinterpret_cast<uint32_t*>(dgi + H) = pz;
          *reinterpret_cast<uint32_t*>(dgi + 2 * H) = pn;
          *reinterpret_cast<uint32_t*>(dgh_ptr) = pq;
        }
        // bias gradients: what the stored (bf16-rounded) planes sum to, so that they equal the column sums of dGI / dQ
        bsum[0][0] += bf_lo(pr); bsum[0][1] += bf_hi(pr);
        bsum[1 % NB][0] += bf_lo(pz); bsum[1 % NB][1] += bf_hi(pz);
        bsum[2 % NB][0] += bf_lo(pn); bsum[2 % NB][1] += bf_hi(pn);
        bsum[3 % NB][0] += bf_lo(pq); bsum[3 % NB][1] += bf_hi(pq);
      } else if (CELL == CELL_LSTM) {
        float di0 = 0.f, di1 = 0.f, df0 = 0.f, df1 = 0.f, dg0 = 0.f, dg1 = 0.f, do0 = 0.f, do1 = 0.f;
        car0 = din0;
        car1 = din1;
        if (act) {
          constexpr int M = NS ? NS : 1;
          const float i0 = bf_lo(sp[0]), i1 = bf_hi(sp[0]), f0 = bf_lo(sp[1 % M]), f1 = bf_hi(sp[1 % M]);
          const float g0 = bf_lo(sp[2 % M]), g1 = bf_hi(sp[2 % M]), o0 = bf_lo(sp[3 % M]), o1 = bf_hi(sp[3 % M]);
          const float tc0 = ftanh(bf_lo(sp[4 % M])), tc1 = ftanh(bf_hi(sp[4 % M]));
          const float dh0 = bf_lo(dout) + din0, dh1 = bf_hi(dout) + din1;
          const float dcn0 = dc0 + dh0 * o0 * (1.f - tc0 * tc0), dcn1 = dc1 + dh1 * o1 * (1.f - tc1 * tc1);
          di0 = dcn0 * g0 * i0 * (1.f - i0);
          di1 = dcn1 * g1 * i1 * (1.f - i1);
          df0 = dcn0 * bf_lo(cp) * f0 * (1.f - f0);
          df1 = dcn1 * bf_hi(cp) * f1 * (1.f - f1);
          dg0 = dcn0 * i0 * (1.f - g0 * g0);
          dg1 = dcn1 * i1 * (1.f - g1 * g1);
          do0 = dh0 * tc0 * o0 * (1.f - o0);
          do1 = dh1 * tc1 * o1 * (1.f - o1);
          car0 = car1 = 0.f;
          dc0 = dcn0 * f0;
          dc1 = dcn1 * f1;
        }
        if (dead) di0 = di1 = __uint_as_float(0x7fc00000u);
        const uint32_t pi = pack_bf16x2(di0, di1), pf = pack_bf16x2(df0, df1), pg = pack_bf16x2(dg0, dg1), po = pack_bf16x2(do0, do1);
        if (st_on) *reinterpret_cast<uint32_t*>(dgi) = pi;
        if (st_on) *reinterpret_cast<uint32_t*>(dgi + H) = pf;
        if (st_on) *reinterpret_cast<uint32_t*>(dgi + 2 * H) = pg;
        if (st_on) *reinterpret_cast<uint32_t*>(dgi + 3 * H) = po;
        publish((u64*)(xo + xgranule_bytes<NROWS>(j, gi_i)), tag | pi, local);
        publish((u64*)(xo + xgranule_bytes<NROWS>(H + j, gi_i)), tag | pf, local);
        publish((u64*)(xo + xgranule_bytes<NROWS>(2 * H + j, gi_i)), tag | pg, local);
        publish((u64*)(xo + xgranule_bytes<NROWS>(3 * H + j, gi_i)), tag | po, local);
        bsum[0][0] += bf_lo(pi); bsum[0][1] += bf_hi(pi);
        bsum[1 % NB][0] += bf_lo(pf); bsum[1 % NB][1] += bf_hi(pf);
        bsum[2 % NB][0] += bf_lo(pg); bsum[2 % NB][1] += bf_hi(pg);
        bsum[3 % NB][0] += bf_lo(po); bsum[3 % NB][1] += bf_hi(po);
      } else {
        float dg0 = 0.f, dg1 = 0.f;
        car0 = din0;
        car1 = din1;
        if (act) {
          const float h0v = bf_lo(hp), h1v = bf_hi(hp);
          dg0 = (bf_lo(dout) + din0) * (1.f - h0v * h0v);
          dg1 = (bf_hi(dout) + din1) * (1.f - h1v * h1v);
          car0 = car1 = 0.f;
        }
        if (dead) dg0 = dg1 = __uint_as_float(0x7fc00000u);
        const uint32_t pg = pack_bf16x2(dg0, dg1);
        if (st_on) *reinterpret_cast<uint32_t*>(dgi) = pg;
        publish((u64*)(xo + xgranule_bytes<NROWS>(j, gi_i)), tag | pg, local);
        bsum[0][0] += bf_lo(pg); bsum[0][1] += bf_hi(pg);
      }
    }
    do_ptr += dstep * N * H;
    if (NS) sv_ptr += dstep * N * NSH_;
    hs_ptr += dstep * N * H;
    dgi_ptr += dstep * N * ldgi;
    if (CELL == CELL_GRU) dgh_ptr += dstep * N * H;
    DS2_PROBE_ONLY(const unsigned long long t3 = __builtin_readcyclecounter(); c_gather += t1 - t0; c_bar += t2 - t1; c_gate += t3 - t2;)
  }
#ifdef DS2_PROBE
  if (a.dbg && p == 0 && (tid == 0 || tid == 255)) {
    const int o = grp * 8 + (tid == 0 ? 0 : 4);
    a.dbg[o + 0] = c_gather;
    a.dbg[o + 1] = c_bar;
    a.dbg[o + 2] = c_gate;
    a.dbg[o + 3] = rounds;
  }
#endif
  (void)rounds;
  if (gate_thread && a.dBacc) {
    float* bo = a.dBacc + ((long)d * N + n) * NB * H + j;
#pragma unroll
    for (int g = 0; g < NB; ++g) *reinterpret_cast<float2*>(bo + (long)g * H) = make_float2(bsum[g][0], bsum[g][1]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BPTT sweep, <= 8 samples per group, gate phase on all four waves (one hidden unit per thread): see k_rnn_persist_fwd4.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ld_bf16(const bf16_t* p) { return __uint_as_float((uint32_t)p->v << 16); }

template <int CELL, int H, int P, bool SP>
__global__ void __launch_bounds__(256, 1) k_rnn_persist_bwd4(PArgs a) {
  constexpr int G = CellInfo<CELL>::G, NS = CellInfo<CELL>::NS;
  constexpr int U = H / P;
  constexpr int TILES = U / 16;
  constexpr int KS = G * H / 128;
  constexpr int X2 = G * H / 2;
  static_assert(U == 32 && (G * H) % 128 == 0, "the one-unit-per-thread gate map assumes 32 hidden units per workgroup");
  static_assert(!SP || ((G * H) % 256 == 0 && H % 64 == 0), "a wave's K-quarter is whole k-blocks of 64; a gate starts on a k-block");
  __shared__ __attribute__((aligned(16))) float part[2][4 * TILES * PT_TILE];
  DS2_PROBE_ONLY(__shared__ unsigned long long tl_[TL_N][TL_K];)
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x % NGROUPS, p = blockIdx.x / NGROUPS;
  const int d = grp / a.gpd, slice = grp % a.gpd;
  const int N = a.N, Tp = a.Tp;
  const int Ns = (N - slice + a.gpd - 1) / a.gpd;
  const int li = lane & 15, lq = lane >> 4;
  constexpr long GH = (long)G * H;
  const long ldgi = (long)a.D * GH;

  uint4 w[TILES][KS];
  {
    const bf16_t* WT = a.W + (long)d * H * GH;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const bf16_t* row = WT + (long)(p * U + 16 * t + li) * GH + wave * (GH / 4) + lq * 8;
#pragma unroll
      for (int k = 0; k < KS; ++k) w[t][k] = *reinterpret_cast<const uint4*>(row + 32 * k);   // SP: the same fragments (see smma16)
    }
  }
  // tag-free exchange (see gather_mma_tf): 4 slots of [G*H/32 k-steps][4][8 rows] x 16 bytes
  constexpr int SLOT_BYTES = (G * H / 32) * 512;
  static_assert(4 * SLOT_BYTES <= 2 * MAXS * X2 * 8, "four payload slots fit where the two tagged parities lived");
  char* xg = (char*)(a.xbuf + (long)grp * 2 * MAXS * X2);
  const int srow = SP ? li : (li & 7), half = li >> 3;
  const bool need = (li & 7) < Ns;
  const int spidx = (li & 8) ? 0xEEEE : 0x4444;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, 4 * SLOT_BYTES, 0x00020000);

  const int e = tid & 1, pid = tid >> 1;      // thread bits as in k_rnn_persist_fwd4
  const int gi_i = SP ? (pid >> 3) & 7 : (pid >> 2) & 7;
  const int up = SP ? (pid >> 6) * 8 + (pid & 3) * 2 + ((pid >> 2) & 1) : ((pid >> 5) & 3) * 4 + (pid & 3);
  const bool gate_thread = gi_i < Ns;
  const int n = slice + a.gpd * gi_i;
  const int j = p * U + 2 * up, ju = j + e;
  int len = 0;
  if (gate_thread) len = a.lens[n];
  float car = 0.f, dc = 0.f;                                // carried dh (elementwise part) and dc of this thread's unit
  const long dstep = d == 0 ? -1 : 1;
  const int t_first = d == 0 ? Tp - 1 : 0;
  const int nn_ = gate_thread ? n : 0;
  constexpr long NSH_ = (long)(NS ? NS : 1) * H;
  const bf16_t* do_ptr = a.dOut + ((long)t_first * N + nn_) * H + ju;
  const bf16_t* sv_ptr = NS ? a.S + (((long)d * Tp + t_first) * N + nn_) * NSH_ + ju : nullptr;
  const bf16_t* hs_ptr = a.Hseq + (long)d * a.hseq_dstride + ((long)t_first * N + nn_) * H + ju;   // h_t
  bf16_t* dgi_ptr = a.dGI + ((long)t_first * N + nn_) * ldgi + (long)d * GH + j;                     // pair base (even lane stores)
  bf16_t* dgh_ptr = a.dGH ? a.dGH + (((long)d * Tp + t_first) * N + nn_) * H + j : nullptr;         // dQ, pair base
  constexpr int NB = CELL == CELL_GRU ? 4 : G;
  float bsum[NB];
#pragma unroll
  for (int g = 0; g < NB; ++g) bsum[g] = 0.f;
  const long prev_off = d == 0 ? -1 : 1;
  bool dead = false;
  const bool local = group_is_xcd_local(a.xcc + grp * 32, p, tid, a.err, a.lerr, a.startup_ms, dead);
  unsigned rounds = 0;
  const int pidx = partial_t_index((2 * up + e) / 16, gi_i, (2 * up + e) % 16);
  const int xoff = SP ? xsp_pair_bytes(j, gi_i) : xtf_pair_bytes(j, gi_i);     // this pair's dword of gate 0 inside a slot

  DS2_PROBE_ONLY(unsigned long long c_gather = 0, c_bar = 0, c_gate = 0;)
  // Everything the gate phase of a step needs (2-byte loads of this thread's unit: dOut, the saved planes, h_prev, c_prev) is
  // loaded TWO STEPS AHEAD, right after a gather has returned: loads that are still on their way from HBM when the gather's loads
  // come back hold the gather up (vmcnt retires in order; probe build: 2.78 -> 2.09 us per step without them).  Unlike in the
  // forward sweep (see k_rnn_persist_fwd4) this pays in the training step, a little: 3.40 -> 3.33 us.  Raw 16-bit values (a
  // conversion at load time would wait right there) in three rotating register sets.
  constexpr int M = NS ? NS : 1;
  struct Pre {
    uint32_t dout, sp[M], hp, cp;
  };
  auto prefetch = [&](Pre& r, int s, long off) {     // the loads of step s, `off` steps ahead of the running pointers
    r.dout = 0u; r.hp = 0u; r.cp = 0u;
#pragma unroll
    for (int q = 0; q < M; ++q) r.sp[q] = 0u;
    if (gate_thread && !DS2_DBG(a, 1)) {
      const int t = d == 0 ? Tp - 1 - s : s;
      const bf16_t* dop = do_ptr + off * dstep * N * H;
      const bf16_t* svp = NS ? sv_ptr + off * dstep * N * NSH_ : nullptr;
      const bf16_t* hsp = hs_ptr + off * dstep * N * H;
      r.dout = dop->v;
#pragma unroll
      for (int q = 0; q < NS; ++q) r.sp[q] = (svp + (long)q * H)->v;
      r.hp = (hsp + prev_off * N * H)->v;              // guard slots / inactive frames hold zeros: unconditional
      if (CELL == CELL_LSTM) {
        const bool has_prev = d == 0 ? (t > 0) : (t + 1 < len);
        if (has_prev) r.cp = (svp + prev_off * N * NSH_ + 4 * H)->v;
      }
      if (CELL == CELL_RNN) r.hp = hsp->v;
    }
  };
  Pre pra, prb, prc;
  prefetch(pra, 0, 0);
  prefetch(prb, Tp > 1 ? 1 : 0, Tp > 1 ? 1 : 0);
  prefetch(prc, 0, 0);            // defined contents for the set that is filled from step 2 on
  // L2 warm-up (l2_touch) of what `prefetch` will load L2_AHEAD steps from now: wave w touches the 64-byte runs of samples 2w, 2w+1
  // in dOut, the NS saved planes and the state sequence (h_prev of a step: guard frames exist at t = -1 and t = T')
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  uint64_t pf_do[2], pf_sv[2], pf_hs[2];
  {
    const int sa = min(L2_AHEAD, Tp - 1);
    const long ta = d == 0 ? Tp - 1 - sa : sa;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long nr = min(slice + a.gpd * max(0, min(2 * wv + i, Ns - 1)), N - 1);
      pf_do[i] = uniform64(a.dOut + (ta * N + nr) * H + p * U);
      pf_sv[i] = NS ? uniform64(a.S + (((long)d * Tp + ta) * N + nr) * NSH_ + p * U) : 0;
      pf_hs[i] = uniform64(a.Hseq + (long)d * a.hseq_dstride + ((ta + prev_off) * N + nr) * H + p * U);
    }
  }
  const uint64_t pf_step_h = uniform64((uint64_t)(dstep * N * H * 2)), pf_step_s = uniform64((uint64_t)(dstep * N * NSH_ * 2));
  auto body = [&](int s, Pre& pu, Pre& pn) {
    DS2_PROBE_ONLY(const unsigned long long t0 = __builtin_readcyclecounter();)
    DS2_TL(0);
    const int t = d == 0 ? Tp - 1 - s : s;
    const int par = s & 1;
    auto touch = [&]() {
      DS2_TL(1);
      if (DS2_DBG(a, 16)) return;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        l2_touch(pf_do[i]);
        l2_touch(pf_hs[i]);
#pragma unroll
        for (int q = 0; q < NS; ++q) l2_touch(pf_sv[i] + (uint64_t)(q * H * 2));
      }
      if (s + L2_AHEAD < Tp - 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          pf_do[i] += pf_step_h;
          pf_hs[i] += pf_step_h;
          pf_sv[i] += pf_step_s;
        }
      }
    };
    ds2_f32x4 acc[TILES];
#pragma unroll
    for (int tt = 0; tt < TILES; ++tt) acc[tt] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
#if DS2_BWD_SLEEP
    if (s > 0) __builtin_amdgcn_s_sleep(DS2_BWD_SLEEP);
#endif
    if (s > 0)
      gather_mma_tf<TILES, KS, SP>(acc, w, rsrc, ((s + 3) & 3) * SLOT_BYTES, wave * KS, lq, srow, half, need, a.err, a.lerr, dead, rounds, touch, spidx);
    else
      touch();
    // this step's values first (loaded two steps ago; the gather has just drained vmcnt), THEN the new loads: converted after
    // them, the compiler's wait-count bookkeeping across the rotated loop would wait for the new loads as well
    const float dout_ = __uint_as_float(pu.dout << 16), hp_ = __uint_as_float(pu.hp << 16), cp_ = __uint_as_float(pu.cp << 16);
    float sp[M];
#pragma unroll
    for (int q = 0; q < M; ++q) sp[q] = __uint_as_float(pu.sp[q] << 16);
    // ... and pinned HERE: left alone, the conversions sink into the gate phase's `act` block, i.e. behind the new loads, and the
    // first use of a raw value then waits vmcnt(0) -- for the loads just issued (a full L2 round trip after every barrier)
    float dout = dout_, hp = hp_, cp = cp_;
    asm volatile("" : "+v"(dout), "+v"(hp), "+v"(cp));
#pragma unroll
    for (int q = 0; q < M; ++q) asm volatile("" : "+v"(sp[q]));
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < Tp) prefetch(pn, s + 2, 2);      // (behind the barrier instead: 1.71 vs 1.63 us per step, profiles/r05h_ab_sweep_timing.txt)
    __builtin_amdgcn_sched_barrier(0);
    DS2_PROBE_ONLY(const unsigned long long t1 = __builtin_readcyclecounter();)
    DS2_TL(2);
    if (SP && DS2_PREADD) preadd_rows<TILES>(acc);
    store_partials_t<TILES>(part[par], acc, wave, lane);
    __syncthreads();
    DS2_PROBE_ONLY(const unsigned long long t2 = __builtin_readcyclecounter();)
    DS2_TL(3);
    // gate gradients of this thread's unit (zeros when inactive); gq[] = what is exchanged, gs[] = what is stored in dGI
    float gx[G], gn = 0.f, gq = 0.f;        // gx: the G exchanged planes; GRU: gx = {dr, dz, dq}, gn = dn (stored), gq = dq
#pragma unroll
    for (int g = 0; g < G; ++g) gx[g] = 0.f;
    if (gate_thread) {
      const bool act = t < len;
      const float* pp = part[par] + pidx;
      float psum = (pp[0] + pp[TILES * PT_TILE]) + (pp[2 * TILES * PT_TILE] + pp[3 * TILES * PT_TILE]);
      if (SP && !DS2_PREADD) psum += (pp[8] + pp[TILES * PT_TILE + 8]) + (pp[2 * TILES * PT_TILE + 8] + pp[3 * TILES * PT_TILE + 8]);   // tile row s + 8
      const float din = car + psum;
      car = din;
      if (CELL == CELL_GRU) {
        if (act) {
          const float r = sp[0], z = sp[1 % M], nn = sp[2 % M], q = sp[3 % M];
          const float dh = dout + din;
          gn = dh * (1.f - z) * (1.f - nn * nn);
          gx[1 % G] = dh * (hp - nn) * z * (1.f - z);
          gx[0] = gn * q * r * (1.f - r);
          gx[2 % G] = gn * r;
          car = dh * z;
        }
        if (dead) gx[0] = __uint_as_float(0x7fc00000u);
      } else if (CELL == CELL_LSTM) {
        if (act) {
          const float ig = sp[0], fg = sp[1 % M], gg = sp[2 % M], og = sp[3 % M];
          const float tc = ftanh(sp[4 % M]);
          const float dh = dout + din;
          const float dcn = dc + dh * og * (1.f - tc * tc);
          gx[0] = dcn * gg * ig * (1.f - ig);
          gx[1 % G] = dcn * cp * fg * (1.f - fg);
          gx[2 % G] = dcn * ig * (1.f - gg * gg);
          gx[3 % G] = dh * tc * og * (1.f - og);
          car = 0.f;
          dc = dcn * fg;
        }
        if (dead) gx[0] = __uint_as_float(0x7fc00000u);
      } else {
        if (act) {
          gx[0] = (dout + din) * (1.f - hp * hp);
          car = 0.f;
        }
        if (dead) gx[0] = __uint_as_float(0x7fc00000u);
      }
    }
    // the pair meets: the even lane publishes and stores the packed values; both lanes accumulate their own bias sums from
    // the ROUNDED values (= the column sums of the stored planes)
    {
      float go[G];
#pragma unroll
      for (int g = 0; g < G; ++g) go[g] = dpp_xor1(gx[g]);
      const float gn_o = dpp_xor1(gn);
      // (selects in front of ONE pack per plane: the pack is an asm statement, which the compiler cannot if-convert -- written as
      // `e == 0 ? pack(a, b) : pack(b, a)` every plane became a divergent branch diamond in front of the publish)
      // BPTT 1.63 -> 1.55 us per time step (GRU), 2.12 -> 2.01 (LSTM): profiles/r05h_ab_sweep_timing.txt
      uint32_t pk[G];
#pragma unroll
      for (int g = 0; g < G; ++g) pk[g] = pack_bf16x2(e == 0 ? gx[g] : go[g], e == 0 ? go[g] : gx[g]);
      const uint32_t pkn = pack_bf16x2(e == 0 ? gn : gn_o, e == 0 ? gn_o : gn);
#ifdef DS2_QUAD_STORES
      u32x4q pubv[G], stv[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        stv[g] = row4(pk[g]);
        pubv[g] = row4(xtf_word(pk[g]));
      }
      const u32x4q nv = row4(pkn);
      if (gate_thread && (tid & 7) == 0) {
        const int xo = (s & 3) * SLOT_BYTES + xoff, xr = ((s + 2) & 3) * SLOT_BYTES + xoff;
#pragma unroll
        for (int g = 0; g < G; ++g) publish128(rsrc, xo + g * (H / 32) * 512, pubv[g], local);
#pragma unroll
        for (int g = 0; g < G; ++g) publish128(rsrc, xr + g * (H / 32) * 512, u32x4q{XSENT, XSENT, XSENT, XSENT}, local);
        if (CELL == CELL_GRU) {
          *reinterpret_cast<u32x4q*>(dgi_ptr) = stv[0];
          *reinterpret_cast<u32x4q*>(dgi_ptr + H) = stv[1 % G];
          *reinterpret_cast<u32x4q*>(dgi_ptr + 2 * H) = nv;
          *reinterpret_cast<u32x4q*>(dgh_ptr) = stv[2 % G];
        } else {
#pragma unroll
          for (int g = 0; g < G; ++g) *reinterpret_cast<u32x4q*>(dgi_ptr + (long)g * H) = stv[g];
        }
      }
#endif
      if (gate_thread) {
#ifndef DS2_QUAD_STORES
        if (e == 0) {
          char* xo = xg + (s & 3) * SLOT_BYTES + xoff;         // gate g, units (j, j+1): element k = g*H + j -> k-step 32*g + p
          char* xr = xg + ((s + 2) & 3) * SLOT_BYTES + xoff;   // re-armed for step s + 2
#pragma unroll
          for (int g = 0; g < G; ++g) publish32(xo + g * (H / 32) * 512, xtf_word(pk[g]), local);
#pragma unroll
          for (int g = 0; g < G; ++g) publish32(xr + g * (H / 32) * 512, XSENT, local);
          DS2_TL(4);
          if (CELL == CELL_GRU) {
            *reinterpret_cast<uint32_t*>(dgi_ptr) = pk[0];
            *reinterpret_cast<uint32_t*>(dgi_ptr + H) = pk[1 % G];
            *reinterpret_cast<uint32_t*>(dgi_ptr + 2 * H) = pkn;
            *reinterpret_cast<uint32_t*>(dgh_ptr) = pk[2 % G];
          } else {
#pragma unroll
            for (int g = 0; g < G; ++g) *reinterpret_cast<uint32_t*>(dgi_ptr + (long)g * H) = pk[g];
          }
        }
#endif
        // own unit's rounded values: low half of the pair word on the even lane, high half on the odd lane
        if (CELL == CELL_GRU) {
          bsum[0] += e == 0 ? bf_lo(pk[0]) : bf_hi(pk[0]);
          bsum[1 % NB] += e == 0 ? bf_lo(pk[1 % G]) : bf_hi(pk[1 % G]);
          bsum[2 % NB] += e == 0 ? bf_lo(pkn) : bf_hi(pkn);
          bsum[3 % NB] += e == 0 ? bf_lo(pk[2 % G]) : bf_hi(pk[2 % G]);
        } else {
#pragma unroll
          for (int g = 0; g < G; ++g) bsum[g % NB] += e == 0 ? bf_lo(pk[g]) : bf_hi(pk[g]);
        }
      }
    }
    do_ptr += dstep * N * H;
    if (NS) sv_ptr += dstep * N * NSH_;
    hs_ptr += dstep * N * H;
    dgi_ptr += dstep * N * ldgi;
    if (CELL == CELL_GRU) dgh_ptr += dstep * N * H;
    l2_touch_retire();
    DS2_PROBE_ONLY(const unsigned long long t3 = __builtin_readcyclecounter(); c_gather += t1 - t0; c_bar += t2 - t1; c_gate += t3 - t2;)
    DS2_TL(5);
  };
  for (int s = 0; s < Tp; s += 3) {
    body(s, pra, prc);
    if (s + 1 < Tp) body(s + 1, prb, pra);
    if (s + 2 < Tp) body(s + 2, prc, prb);
  }
#ifdef DS2_PROBE
  if (a.tl && grp == 0 && tid == 0)
    for (int i = 0; i < TL_N * TL_K; ++i) a.tl[(long)p * TL_N * TL_K + i] = tl_[i / TL_K][i % TL_K];
  if (a.dbg && p == 0 && (tid == 0 || tid == 255)) {
    const int o = grp * 8 + (tid == 0 ? 0 : 4);
    a.dbg[o + 0] = c_gather;
    a.dbg[o + 1] = c_bar;
    a.dbg[o + 2] = c_gate;
    a.dbg[o + 3] = rounds;
  }
#endif
  (void)rounds;
  if (gate_thread && a.dBacc) {
    float* bo = a.dBacc + ((long)d * N + n) * NB * H + ju;
#pragma unroll
    for (int g = 0; g < NB; ++g) bo[(long)g * H] = bsum[g];
  }
}

// dense = the round-2..4 form of the 8-clip kernels (half of every 16-row tile is padding); default: the structured-sparse form
template <int CELL, int H, int P>
int launch(bool bwd, const PArgs& a, hipStream_t st, bool dense) {
  const bool split = (a.N + a.gpd - 1) / a.gpd <= 8;   // <= 8 samples per group: lane pairs share the gather
  if (bwd) {
    if (split && dense)
      hipLaunchKernelGGL((k_rnn_persist_bwd4<CELL, H, P, false>), dim3(NGROUPS * P), dim3(256), 0, st, a);
    else if (split)
      hipLaunchKernelGGL((k_rnn_persist_bwd4<CELL, H, P, true>), dim3(NGROUPS * P), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((k_rnn_persist_bwd<CELL, H, P, 1>), dim3(NGROUPS * P), dim3(256), 0, st, a);
  } else {
    if (split && dense)
      hipLaunchKernelGGL((k_rnn_persist_fwd4<CELL, H, P, false>), dim3(NGROUPS * P), dim3(256), 0, st, a);
    else if (split)
      hipLaunchKernelGGL((k_rnn_persist_fwd4<CELL, H, P, true>), dim3(NGROUPS * P), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((k_rnn_persist_fwd<CELL, H, P, 1>), dim3(NGROUPS * P), dim3(256), 0, st, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}


}  // namespace ds2p
