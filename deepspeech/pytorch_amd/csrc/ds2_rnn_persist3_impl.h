// General persistent recurrent sweeps, round 4 (bf16 storage): 32 hidden units per workgroup for ANY hidden size that is a
// multiple of 32 -- BASELINE.json config 5 (7 x LSTM-1280, 64 clips) and every width the tuned H = 1024 kernels do not cover.
//
// What round 3's general kernels (ds2_rnn_persist2_impl.h: 16 units per workgroup) paid for, and what changes here:
//   * a workgroup gathers the WHOLE exchanged vector of its group's samples every step through one CU's vector-memory path
//     (~64 GB/s when the group spans XCDs): config 5a had 2 groups of 64 samples on 160 of the 256 CUs, 655 KB per workgroup and
//     BPTT step.  With 32 units per workgroup a group is P = H/32 workgroups: H = 1280 -> 6 groups of 40 (240 CUs), 22 samples per
//     group (a third of the bytes); H <= 1024 -> P <= 32 and the groups are laid out INSIDE one XCD (8 x floor(32/P) groups,
//     block b -> XCD b % 8): publishes are plain stores that stay in that XCD's L2 (905 instead of 2 236 cycles per bare exchange
//     step, profiles/r03k_exchange.txt) -- checked at start-up by the XCC-id handshake, correct under any placement;
//   * the W_hh slice of 32 units x G gates (LSTM-1280: 327 KB) no longer fits the register file beside the accumulators: part of
//     every wave's B fragments lives in LDS, ready-made (one conflict-free ds_read_b128 per fragment, read a k-step or three ahead
//     of its MFMA), the rest in registers (plan3: <= 224-256 weight registers per lane, LDS <= 156 KB incl. the partial sums);
//   * a group's samples are cut into NSET SETS of <= 16 consecutive samples (one MFMA m-tile each).  Samples are independent
//     recurrences, so a time step is NSET half-steps, each with its own four exchange slots: while set A's hand-off is in flight
//     the workgroup multiplies set B -- the exchange latency (the floor of a persistent sweep) hides behind the other set's work,
//     and only ONE m-tile of accumulators, gather buffers and gate state is live at a time.  A set's half-steps are executed only
//     while one of its clips is inside its sequence (sched3): with the batch sorted by length the second set holds the group's
//     shortest clips and drops out of the schedule early (joins it late in a descending sweep);
//   * all 256 threads run the gate phase (one sample row x one pair of hidden units each), thread bits in the order of the
//     exchange layout so that a wave publishes one contiguous 1 KiB run.
// Exchange: payload-only bf16 in MFMA A-fragment order, four slots per set, the all-ones dword = "not published yet", publishers
// re-arm slot (s + 2) & 3 at step s -- the protocol of gather_mma_tf (ds2_rnn_persist_impl.h), model-checked in
// tests/test_exchange_protocol.py.  Safety as everywhere: grid <= CU count (one workgroup per CU by LDS / register use), bounded
// spins, *err / per-launch word, NaN poisoning.  Reference: BatchRNN.forward, model.py:94-102 (nn.GRU / nn.LSTM and their autograd).
#pragma once
#include "ds2_rnn_persist2_impl.h"

namespace ds2r {
using namespace ds2q;

struct RArgs {
  QArgs q;
  u64* xcc;      // [NG][32] start-up exchange of the workgroups' XCC ids (xmap only)
  int P;         // workgroups per group = H / 32
  int xmap;      // 1: group g's workgroups are the blocks with blockIdx % 8 == g % 8 (one XCD, if the dispatcher keeps its habit)
  int gx;        // xmap: group slots per XCD (floor(32 / P)); the grid is 8 * gx * P blocks, groups >= NG stay empty
  int nset;      // sample sets per group (the kernel's NSET)
  int skip;      // 1: a set executes only the time steps at which one of its clips is inside its sequence (the caller zeroes the
                 // padding rows of the outputs: they are no longer written for the steps left out)
  int sparse;    // 1: sets of <= 8 clips on the structured-sparse products (Gather3S); 0: dense 16-row tiles
#ifdef DS2_PROBE   // tools/probe_persist3.py builds its own library with -DDS2_PROBE; the shipping kernels carry none of it
  unsigned long long* dbg;  // [NG][8] cycle counters of workgroup 0 of each group (thread 0 and thread 255)
  int dbgmask;              // 1 skip GI / dOut / S prefetch loads, 2 skip output stores, 8 skip the gather + products (no exchange),
                            // 32 skip the products with the LDS-resident fragments, 64 publish with plain stores whatever the placement
                            // (timing only: results are wrong across XCDs), 128 keep the gather but skip the products
#endif
};

// Which B fragments of a wave's K-quarter (RT tiles x KSW k-steps) live in LDS instead of registers.  Forward kernels (RT = 2G
// tiles per k-step): the tiles t >= TR of EVERY k-step, so that each k-step's few LDS reads hide under the register tiles' MFMAs.
// BPTT kernels (RT = 2): whole k-steps, spread evenly: k-step k is LDS-resident iff bit (k % 10) of kmask is set.  PB = partial-sum
// buffers (2: one barrier per half-step; 1: a second barrier in front of the stores, when LDS is short).
// Compile-time ablations for tools/ab_variants.py (-DDS2R_VAR=<bits>, timing only -- results are wrong): the DS2_PERSIST_DBG bits of
// the probe build (1 no prefetch loads of the gate operands, 2 no output stores, 8 no gather and no products, 32 no products with the
// LDS-resident fragments, 128 gather but no products) plus 256 no gate math (every frame treated as inactive) and 512 no gather loads
// (products on zeros) -- without the probe build's counters and run-time branches, i.e. in the shipping kernels' own schedule.
#ifndef DS2R_VAR
#define DS2R_VAR 0
#endif

struct Plan3 {
  int TR;
  unsigned kmask;
  int PB, nlds;      // nlds = LDS-resident fragments per wave
  bool ok;
};
// (scalar arguments: device code must not odr-use a constexpr Plan3 object)
// PS = pair shift: 0 for the dense kernels (k-steps of 32); 1 for the structured-sparse ones (round 6), whose matrix instruction
// takes the fragments of k-steps (2u, 2u + 1) as ONE operand: whole k-BLOCKS of 64 are LDS-resident or not (bit (u % 10) of kmask).
constexpr bool in_lds3(int TR, unsigned kmask, int t, int k, int PS = 0) { return t >= TR || ((kmask >> ((k >> PS) % 10)) & 1u); }
constexpr int count_lds3(int TR, unsigned kmask, int RT, int KSW, int PS = 0) {
  int n = 0;
  for (int k = 0; k < KSW; ++k)
    for (int t = 0; t < RT; ++t) n += in_lds3(TR, kmask, t, k, PS) ? 1 : 0;
  return n;
}
// index of fragment (t, k) among the LDS-resident ones, k-major
constexpr int lds_index3(int TR, unsigned kmask, int RT, int t, int k, int PS = 0) {
  int n = 0;
  for (int kk = 0; kk <= k; ++kk)
    for (int tt = 0; tt < RT; ++tt)
      if ((kk < k || tt < t) && in_lds3(TR, kmask, tt, kk, PS)) ++n;
  return n;
}
// position of tile t among the LDS-resident tiles of k-step k (the staging ring's second index)
constexpr int lds_tile3(int TR, unsigned kmask, int t, int k, int PS = 0) { return ((kmask >> ((k >> PS) % 10)) & 1u) ? t : t - TR; }
constexpr int PT3_COL = 20, PT3_TILE = 16 * PT3_COL;     // partial-sum tiles stored [col][row], column stride 20 floats (conflict-free)
constexpr int part_bytes3(int RT, int PB) { return PB * 4 * RT * PT3_TILE * 4; }
constexpr int FR_MAX3 = 56, FR_CAP3 = 64, LDS_MAX3 = 156 * 1024;
constexpr Plan3 plan3(int RT, int KSW, int PS = 0) {
  const int F = RT * KSW;
  if (F <= FR_MAX3) {
    Plan3 pl{RT, 0u, 2, 0, part_bytes3(RT, 2) <= LDS_MAX3};
    return pl;
  }
  if (RT >= 4) {
    const int tr0 = FR_MAX3 / KSW > 0 ? FR_MAX3 / KSW : 1;
    for (int tr = tr0; tr <= RT && tr * KSW <= FR_CAP3; ++tr)
      for (int pb = 2; pb >= 1; --pb) {
        Plan3 pl{tr, 0u, pb, (RT - tr) * KSW, true};
        if (pl.nlds * 4096 + part_bytes3(RT, pb) <= LDS_MAX3) return pl;
      }
    return Plan3{0, 0u, 0, 0, false};
  }
  for (int n10 = 1; n10 <= 9; ++n10) {          // n10 of every 10 k-steps (k-blocks) in LDS, evenly spread
    unsigned m = 0;
    for (int i = 0; i < n10; ++i) m |= 1u << ((i * 10 + 5) / n10);
    Plan3 pl{RT, m, 2, 0, true};
    pl.nlds = count_lds3(RT, m, RT, KSW, PS);
    if (F - pl.nlds > FR_CAP3 - 4) continue;
    for (int pb = 2; pb >= 1; --pb) {
      pl.PB = pb;
      if (pl.nlds * 4096 + part_bytes3(RT, pb) <= LDS_MAX3) return pl;
    }
  }
  return Plan3{0, 0u, 0, 0, false};
}
constexpr int lds_bytes3(int RT, int KSW, int PS = 0) {
  const Plan3 pl = plan3(RT, KSW, PS);
  return pl.nlds * 4096 + part_bytes3(RT, pl.PB);
}

// XCC-id handshake of ds2_rnn_persist_impl.h for groups of P <= 32 workgroups
__device__ __forceinline__ bool group_is_xcd_local3(u64* slots, int p, int P, int tid, int* err, int* lerr, unsigned startup_ms, bool& dead) {
  __shared__ int s_local3;
  if (tid < 64) {
    const unsigned my = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;   // HW_REG_XCC_ID[3:0]
    if (tid == 0) g_store(slots + p, (0x5ca1ab1eull << 32) | my);
    bool same = true;
    unsigned spins = 0;
    const u64 t0 = wall_clock64();
    for (;;) {
      u64 v = 0;
      if (tid < P) v = g_load(slots + tid);
      const bool bad = tid < P && (unsigned)(v >> 32) != 0x5ca1ab1eu;
      if (!__any(bad)) {
        same = !(tid < P) || ((unsigned)v == my);
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
    if (tid == 0) s_local3 = all_same ? 1 : 0;
  }
  __syncthreads();
  return s_local3 != 0;
}

__device__ __forceinline__ void pub32(char* p, uint32_t v, bool local) {
  if (local)
    __builtin_nontemporal_store(v, (uint32_t*)p);     // plain-policy store: stays in this XCD's L2
  else
    __hip_atomic_store((uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A/B VARIANT (-DDS2R_QUAD_STORES), measured and not kept: the four lanes of a quad (tid & 3 = dw) hold four CONSECUTIVE dwords of a
// sample row -- in the exchange slot and in every stored plane.  quad_gather() collects them on every lane (meaningful on the quad's
// lane 0), so that lane 0 issues ONE 16-byte store where four lanes issued four 4-byte ones.  Fewer active lanes per store, but three
// DPP moves per stored dword in front of the publish: config 5a 108.1 vs 106.2 ms with plain dword stores, 5b 58.0 vs 57.7
// (profiles/r04g_ab_dword_stores_*.txt); the same idea loses 3 % in the tuned H = 1024 kernels (r04f_ab_quad_stores_cfg3.txt).
// Without the flag quad_gather() is the identity on element 0 and every lane stores its own dword.
__device__ __forceinline__ u32x4_t quad_gather(uint32_t v) {
  u32x4_t r;
  r[0] = v;
#ifndef DS2R_QUAD_STORES
  r[1] = r[2] = r[3] = 0u;
  return r;
#endif
  r[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55 /* quad_perm [1,1,1,1] */, 0xf, 0xf, true);
  r[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA /* quad_perm [2,2,2,2] */, 0xf, 0xf, true);
  r[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF /* quad_perm [3,3,3,3] */, 0xf, 0xf, true);
  return r;
}
// 16-byte publish into the group's exchange buffer (byte offset `off` inside the resource): plain policy inside one XCD, write-through
// (sc1) across XCDs.  Dwords are what the protocol looks at (sentinel or data), so a 16-byte store need not be atomic as a whole.
__device__ __forceinline__ void pub128(__amdgpu_buffer_rsrc_t rsrc, int off, u32x4_t v, bool local) {
  if (local)
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 0);
  else
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ void st128(bf16_t* p, u32x4_t v) { *reinterpret_cast<u32x4_t*>(p) = v; }

__device__ __forceinline__ uint32_t pay_word(float a, float b) {
  const uint32_t pk = cvt_pk_bf16(a, b);
  return pk == XSENT2 ? 0x7fc07fc0u : pk;
}

// EXPERIMENT, OFF BY DEFAULT (-DDS2R_ASM_MFMA; measured and rejected in round 4: cfg5a 111 -> 104 ms, cfg5b 58.6 -> 57.0, but WRONG
// RESULTS on every shape whose MFMAs are not separated by LDS reads -- 15 of 32 kernel tests -- although the instruction itself is
// fine: tools/_build/mfma_asm_test.hip reproduces the builtin bit for bit with B in AGPRs.  The compiler cannot see that an asm
// statement is an MFMA, so it re-uses source registers inside the few cycles in which the matrix pipe still reads them; keeping the
// fragments live across a pad was not enough).  What it is:
// MFMA with the B fragment read STRAIGHT from an accumulation register (AGPR).  These kernels hold 400-490 registers per lane, so the
// compiler parks the resident W fragments in AGPRs -- and, through the builtin, copies every one of them into VGPRs in front of its
// MFMA (4 x v_accvgpr_read_b32: 240 extra VALU-slot instructions per forward half-step at LSTM-1280, as many issue cycles as the
// MFMAs' own pipe time).  gfx950's MFMA takes srcB from the AGPR file directly; the builtin never asks for it, inline asm does.
// The compiler does not see an MFMA in the asm, so it pads no hazards: accumulators are reused at a distance of >= 4 MFMAs (> the 18
// wait states any XDL result needs), and mfma_results_ready() -- an asm statement every accumulator passes through -- holds the 18
// wait states between the last MFMA and the first reader of its result.
__device__ __forceinline__ void mfma_breg(ds2_f32x4& acc, const uint4& a, const uint4& b) {       // b: register-resident fragment
#ifndef DS2R_ASM_MFMA
  Mma<bf16_t>::mma16(acc, a, b);
  return;
#endif
  const u32x4_t av = __builtin_bit_cast(u32x4_t, a), bv = __builtin_bit_cast(u32x4_t, b);
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "a"(bv));
}
__device__ __forceinline__ void mfma_bvgpr(ds2_f32x4& acc, const uint4& a, const uint4& b) {      // b: staged from LDS
#ifndef DS2R_ASM_MFMA
  Mma<bf16_t>::mma16(acc, a, b);
  return;
#endif
  const u32x4_t av = __builtin_bit_cast(u32x4_t, a), bv = __builtin_bit_cast(u32x4_t, b);
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
}
#ifndef DS2R_READY_PAD
#define DS2R_READY_PAD "s_nop 15\n\ts_nop 3"
#endif
template <int NA>
__device__ __forceinline__ void mfma_results_ready(ds2_f32x4 (&acc)[NA]) {
  static_assert(NA == 2 || NA == 4 || NA == 6 || NA == 8, "accumulator count");
#ifndef DS2R_ASM_MFMA
  return;
#endif
  if constexpr (NA == 2)
    asm volatile(DS2R_READY_PAD : "+v"(acc[0]), "+v"(acc[1]));
  else if constexpr (NA == 4)
    asm volatile(DS2R_READY_PAD : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
  else if constexpr (NA == 6)
    asm volatile(DS2R_READY_PAD : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]));
  else
    asm volatile(DS2R_READY_PAD : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                 "+v"(acc[7]));
}

// Resident weight fragments beyond what the vector registers hold: these kernels keep 56-60 B fragments (224-240 registers) per lane
// beside gather buffers and gate state, so the register allocator SPILLS about a third of them into accumulation registers and copies
// each one back in front of its MFMA (4 x v_accvgpr_read_b32 per fragment: 120-310 VALU instructions per half-step on one wave per
// SIMD, round 4's PMC pass: a third of a half-step's issue cycles are VALU).  gfx950's MFMA reads srcB from the accumulation file
// directly, and the compiler does so by itself for a value whose register class IS an accumulation register: an empty asm with an
// "a" constraint pins the fragment there once, at load time; the builtin MFMA (which the compiler knows as one: hazards handled)
// then takes it in place.  (Round 4 tried the same through an asm MFMA, which the compiler could not schedule safely.)
// Which fragments: the register-resident ones from ordinal VKEEP on, in (k, t) order.
#ifndef DS2R_VKEEP_FWD
#define DS2R_VKEEP_FWD 20
#endif
#ifndef DS2R_VKEEP_BWD
#define DS2R_VKEEP_BWD 0
#endif
__device__ __forceinline__ void pin_agpr(uint4& f) {
  u32x4_t v = __builtin_bit_cast(u32x4_t, f);
  asm volatile("" : "+a"(v));
  f = __builtin_bit_cast(uint4, v);
}
// ordinal of fragment (t, k) among the register-resident ones, k-major
constexpr int reg_index3(int TR, unsigned kmask, int RT, int t, int k, int PS = 0) {
  int n = 0;
  for (int kk = 0; kk <= k; ++kk)
    for (int tt = 0; tt < RT; ++tt)
      if ((kk < k || tt < t) && !in_lds3(TR, kmask, tt, kk, PS)) ++n;
  return n;
}
typedef __attribute__((ext_vector_type(8))) unsigned int u32x8_t;
__device__ __forceinline__ void pin_agpr8(u32x8_t& v) { asm volatile("" : "+a"(v)); }

// k-steps per gather chunk: two chunks per K-quarter up to 16 k-steps (the products of the first overlap the arrival of the
// second), chunks of 8 beyond (BPTT: K = G*H)
#ifndef DS2R_CHUNK_MAX
#define DS2R_CHUNK_MAX 16
#endif
constexpr int chunk3(int KSW, int SP) { return KSW <= 2 * DS2R_CHUNK_MAX ? ((KSW + 1) / 2 + SP - 1) / SP * SP : 8; }

// The gather of one wave: a stream of chunks (CH k-steps = CH / SP 16-byte loads per lane) through TWO register buffers.  The kernel
// keeps two chunks in flight: after the products of a chunk its buffer is refilled with the chunk two ahead -- of the same
// half-step or, with two sample sets, of the half-step executed NEXT (normally the other set's, published a half-step ago), so
// the fabric round trip of the exchange overlaps this half-step's products, barrier and gate phase instead of preceding them.
//   issue(b, c, base...)  loads chunk c (k-steps c*CH ..) of the slot at byte offset `base` (= set + slot + lq*256 + row*16; k-step k
//                         adds k*1024) into buffer b.  Lanes without a sample row / beyond the wave's ragged share load from
//                         beyond the resource: zeros, no branch.  SP == 2 (<= 8 rows): lane (part, row) = (li >> 3, li & 7) loads
//                         the k-steps part, part + 2, ...; the fragment of an odd k-step is rotated into place before its MFMA;
//   bad(b)                some lane of the chunk still holds the all-ones "not published yet" dword;
//   lds_prefetch(c)       starts the LDS reads of the chunk's first LA k-steps (independent of the gather: issued before the check);
//   mma(b, c, ...)        acc[t] += A(chunk) * W[t](chunk)^T, register-resident fragments from w, LDS-resident ones through a ring
//                         of LA + 1 k-steps of staging registers (read LA k-steps ahead of their MFMAs).
template <int RT, int KSW, int SP, bool RAGGED, bool BWD>
struct Gather3 {
  static constexpr int TR = plan3(RT, KSW).TR, NLDS = plan3(RT, KSW).nlds;
  static constexpr unsigned KM = plan3(RT, KSW).kmask;
  static constexpr int PS = 0;
  static constexpr int CH = chunk3(KSW, SP), PER = CH / SP, NCH = (KSW + CH - 1) / CH;
  typedef uint4 WArr[RT][KSW];       // the register-resident B fragments (only those are ever touched)
  int spidx;                         // (unused by the dense form)
  static __device__ __forceinline__ void set_frag(WArr& w, int t, int k, const uint4& f) { w[t][k] = f; }
  template <int VKEEP>
  static __device__ __forceinline__ void pin(WArr& w) {      // see pin_agpr
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int k = 0; k < KSW; ++k)
        if (!in_lds3(TR, KM, t, k) && reg_index3(TR, KM, RT, t, k) >= VKEEP) pin_agpr(w[t][k]);
  }
  static constexpr int LA = BWD ? 3 : 1;
  static constexpr int TL = NLDS == 0 ? 1 : (BWD ? RT : (RT - TR > 0 ? RT - TR : 1));   // LDS fragments of one k-step, at most
  static_assert(CH % SP == 0 && NCH >= 2, "two chunks at least, tiling into lane parts");
  u32x4_t v[2][PER];
  uint4 st[LA + 1][TL];

  __device__ __forceinline__ void issue(int b, int c, __amdgpu_buffer_rsrc_t rsrc, int base, bool need, int ks0, int cnt, int part) {
    // per-lane offset once per chunk (rows without a sample: beyond the resource for every k-step of the chunk), the k-step as the
    // instruction's SCALAR offset: no vector arithmetic per load (it was an add and a select per load: 160 of a BPTT half-step's 670
    // VALU instructions)
    const int lane_off = need ? base + part * 1024 : XOOB;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int k0 = c * CH + SP * i;                                  // k-step of lane part 0 (compile-time after unrolling)
      const bool k_ok = k0 < KSW && (!RAGGED || k0 < cnt);             // wave-uniform
      const int soff = __builtin_amdgcn_readfirstlane(k_ok ? (ks0 + k0) * 1024 : 0);
      int voff = k_ok ? lane_off : XOOB;
      if (SP == 2 && k0 + 1 >= KSW) voff = part ? XOOB : voff;         // odd K-quarter: the second lane part has no k-step here
      if (SP == 2 && RAGGED) voff = (part && k0 + 1 >= cnt) ? XOOB : voff;
      if (DS2R_VAR & 512)
        v[b][i] = u32x4_t{0u, 0u, 0u, 0u};
      else
#ifdef DS2R_NO_SOFF
        v[b][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff == XOOB ? XOOB : voff + soff, 0, 16 /* sc1 */);
#else
        v[b][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 16 /* sc1 */);
#endif
    }
  }
  __device__ __forceinline__ bool bad(int b) const {
    uint32_t mx = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) mx = max(max(mx, max(v[b][i][0], v[b][i][1])), max(v[b][i][2], v[b][i][3]));
    return mx == XSENT2;
  }
  __device__ __forceinline__ void lds_kstep(int slot, int k_, const uint4* wl_lane) {
    if (NLDS > 0 && k_ < KSW) {
#pragma unroll
      for (int t = 0; t < RT; ++t)
        if (in_lds3(TR, KM, t, k_)) st[slot][lds_tile3(TR, KM, t, k_)] = wl_lane[lds_index3(TR, KM, RT, t, k_) * 256];
    }
  }
  __device__ __forceinline__ void lds_prefetch(int c, const uint4* wl_lane) {
#pragma unroll
    for (int kk = 0; kk < LA; ++kk)
      if (kk < CH) lds_kstep(kk % (LA + 1), c * CH + kk, wl_lane);
  }
  static constexpr int NACC = BWD ? 2 * RT : RT;     // BPTT (two tiles): even and odd k-steps accumulate apart, so that an accumulator is
                                                     // reused every 4th MFMA at the earliest (see mfma_breg)
  __device__ __forceinline__ void mma(int b, int c, ds2_f32x4 (&acc)[NACC], const WArr& w, const uint4* wl_lane, int dbgmask) {
    u32x4_t rot[SP == 2 ? PER : 1];     // SP == 2: the rotated fragments of the odd k-steps
#pragma unroll
    for (int kk = 0; kk < CH; ++kk) {
      const int k_ = c * CH + kk;                    // compile-time after unrolling
      if (kk + LA < CH) lds_kstep((kk + LA) % (LA + 1), k_ + LA, wl_lane);
      if (k_ < KSW && !(dbgmask & 128)) {
        const int i = kk / SP;
        uint4 a_ = make_uint4(v[b][i][0], v[b][i][1], v[b][i][2], v[b][i][3]);
        if (SP == 2 && (kk & 1)) {
          a_ = row_from_plus4(a_, 8);
          rot[i] = u32x4_t{a_.x, a_.y, a_.z, a_.w};
          a_ = make_uint4(rot[i][0], rot[i][1], rot[i][2], rot[i][3]);
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          ds2_f32x4& ac = acc[BWD ? t + RT * (k_ & 1) : t];
          if (in_lds3(TR, KM, t, k_)) {
            const uint4 wv = st[kk % (LA + 1)][lds_tile3(TR, KM, t, k_)];
            if (!(dbgmask & 32)) mfma_bvgpr(ac, a_, wv);
          } else {
            mfma_breg(ac, a_, w[t][k_]);
          }
        }
      }
    }
#ifdef DS2R_ASM_MFMA
    // The compiler does not know that the asm statements above are MFMAs whose source registers are still being read for a few
    // cycles after issue: left alone it re-uses an A-fragment register for the next VALU result right behind the last MFMA (the
    // sentinel check's v_max landed in the fragment of the last k-step: wrong sums on every shape without LDS reads in between).
    // So the chunk's fragments stay live across a pad behind the block.
    asm volatile("s_nop 15\n\ts_nop 3");
#pragma unroll
    for (int i = 0; i < PER; ++i) asm volatile("" : : "v"(v[b][i]));
    if (SP == 2) {
#pragma unroll
      for (int i = 0; i < PER; ++i)
        if ((c * CH + SP * i + 1) < KSW && !(dbgmask & 128)) asm volatile("" : : "v"(rot[i]));
    }
#endif
  }
};

// ---- Structured-sparse form (round 6): sets of <= 8 clips on all 16 tile rows ---------------------------------------------------
// What ds2_rnn_persist_impl.h does for config 3 (see smma16 there), in the general kernels: tile rows s and s + 8 both belong to clip
// s -- row s carries its k = 0, 1 (mod 4) elements, row s + 8 its k = 2, 3 (mod 4) ones, so every row is 2:4-sparse by construction --
// and v_smfmac_f32_16x16x64_bf16 multiplies a k-BLOCK of 64 at the cost of a dense k-step of 32: half the matrix instructions for a
// set, and D[s] + D[s + 8] is the full dot product.  Exchange slot: [k-block][lq][tile row (16)] x 16 bytes (ds2p::xsp_unit_bytes /
// xsp_pair_bytes); a lane loads ONE 16-byte unit per k-block = its compressed A fragment; the B operand of k-block u is the pair of
// dense fragments of k-steps (2u, 2u + 1), kept as one 256-bit value (registers: pinned pairs; LDS: two reads into one tuple).
// Needs whole k-blocks per wave: K-quarters of an even number of k-steps (H % 256 == 0).
__device__ __forceinline__ void smma16v(ds2_f32x4& acc, const u32x4_t& a, const u32x8_t& b, int idx) {
  acc = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(ds2_bf16x8, a), __builtin_bit_cast(ds2_bf16x16, b), acc, idx, 0, 0);
}
template <int RT, int KSW, bool BWD>
struct Gather3S {
  static_assert(KSW % 2 == 0, "whole k-blocks of 64 per wave");
  static constexpr int PS = 1, KB = KSW / 2;
  static constexpr int TR = plan3(RT, KSW, 1).TR, NLDS = plan3(RT, KSW, 1).nlds;
  static constexpr unsigned KM = plan3(RT, KSW, 1).kmask;
  static constexpr int CH = chunk3(KB, 1), PER = CH, NCH = (KB + CH - 1) / CH;
  static constexpr int LA = BWD ? 3 : 1;             // k-blocks the LDS reads run ahead of their products
  static constexpr int TL = NLDS == 0 ? 1 : (BWD ? RT : (RT - TR > 0 ? RT - TR : 1));
  static_assert(NCH >= 2, "two chunks at least");
  typedef u32x8_t WArr[RT][KB];
  u32x4_t v[2][PER];
  u32x8_t st[LA + 1][TL];
  int spidx;                                         // this lane's index word: 0x4444 (tile rows 0-7) / 0xEEEE (rows 8-15)
  static __device__ __forceinline__ void set_frag(WArr& w, int t, int k, const uint4& f) {
    const int o = (k & 1) * 4;
    w[t][k >> 1][o] = f.x; w[t][k >> 1][o + 1] = f.y; w[t][k >> 1][o + 2] = f.z; w[t][k >> 1][o + 3] = f.w;
  }
  template <int VKEEP>
  static __device__ __forceinline__ void pin(WArr& w) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int u = 0; u < KB; ++u)
        if (!in_lds3(TR, KM, t, 2 * u, 1) && reg_index3(TR, KM, RT, t, 2 * u, 1) >= VKEEP) pin_agpr8(w[t][u]);
  }
  __device__ __forceinline__ void issue(int b, int c, __amdgpu_buffer_rsrc_t rsrc, int base, bool need, int ks0, int /*cnt*/, int /*part*/) {
    const int lane_off = need ? base : XOOB;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = c * CH + i;                                        // k-block of the wave's K-quarter (compile-time after unrolling)
      const bool ok = u < KB;
      const int soff = __builtin_amdgcn_readfirstlane(ok ? (ks0 / 2 + u) * 1024 : 0);
      const int voff = ok ? lane_off : XOOB;
      if (DS2R_VAR & 512)
        v[b][i] = u32x4_t{0u, 0u, 0u, 0u};
      else
        v[b][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 16 /* sc1 */);
    }
  }
  __device__ __forceinline__ bool bad(int b) const {
    uint32_t mx = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) mx = max(max(mx, max(v[b][i][0], v[b][i][1])), max(v[b][i][2], v[b][i][3]));
    return mx == XSENT2;
  }
  __device__ __forceinline__ void lds_unit(int slot, int u, const uint4* wl_lane) {
    if (NLDS > 0 && u < KB) {
#pragma unroll
      for (int t = 0; t < RT; ++t)
        if (in_lds3(TR, KM, t, 2 * u, 1)) {
          const uint4 f0 = wl_lane[lds_index3(TR, KM, RT, t, 2 * u, 1) * 256], f1 = wl_lane[lds_index3(TR, KM, RT, t, 2 * u + 1, 1) * 256];
          st[slot][lds_tile3(TR, KM, t, 2 * u, 1)] = u32x8_t{f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        }
    }
  }
  __device__ __forceinline__ void lds_prefetch(int c, const uint4* wl_lane) {
#pragma unroll
    for (int kk = 0; kk < LA; ++kk)
      if (kk < CH) lds_unit(kk % (LA + 1), c * CH + kk, wl_lane);
  }
  static constexpr int NACC = BWD ? 2 * RT : RT;
  __device__ __forceinline__ void mma(int b, int c, ds2_f32x4 (&acc)[NACC], const WArr& w, const uint4* wl_lane, int dbgmask) {
#pragma unroll
    for (int kk = 0; kk < CH; ++kk) {
      const int u = c * CH + kk;                     // compile-time after unrolling
      if (kk + LA < CH) lds_unit((kk + LA) % (LA + 1), u + LA, wl_lane);
      if (u < KB && !(dbgmask & 128)) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          ds2_f32x4& ac = acc[BWD ? t + RT * (u & 1) : t];
          if (in_lds3(TR, KM, t, 2 * u, 1)) {
            if (!(dbgmask & 32)) smma16v(ac, v[b][kk], st[kk % (LA + 1)][lds_tile3(TR, KM, t, 2 * u, 1)], spidx);
          } else {
            smma16v(ac, v[b][kk], w[t][u], spidx);
          }
        }
      }
    }
  }
};

// partial sums of the 4 K-quarters: tile t of wave w at part + (w*RT + t)*PT3_TILE, stored [col][row] with a 20-float column
// stride: the four rows a lane holds of an accumulator are ONE conflict-free 16-byte store
template <int RT>
__device__ __forceinline__ void store_partials3(float* part, const ds2_f32x4 (&acc)[RT], int wave, int lane) {
#pragma unroll
  for (int t = 0; t < RT; ++t)
    *reinterpret_cast<ds2_f32x4*>(part + (wave * RT + t) * PT3_TILE + (lane & 15) * PT3_COL + 4 * (lane >> 4)) = acc[t];
}
// sum over the 4 waves of the columns (col, col + 1) of row `row` of tile t; SS (structured-sparse sets): of rows `row` and `row + 8`,
// the two halves of clip `row`'s dot product
template <int RT, bool SS = false>
__device__ __forceinline__ float2 load_partials3(const float* part, int t, int row, int col) {
  const float* p0 = part + t * PT3_TILE + col * PT3_COL + row;
  float2 s;
  s.x = (p0[0] + p0[RT * PT3_TILE]) + (p0[2 * RT * PT3_TILE] + p0[3 * RT * PT3_TILE]);
  s.y = (p0[PT3_COL] + p0[RT * PT3_TILE + PT3_COL]) + (p0[2 * RT * PT3_TILE + PT3_COL] + p0[3 * RT * PT3_TILE + PT3_COL]);
  if (SS) {
    const float* p8 = p0 + 8;
    s.x += (p8[0] + p8[RT * PT3_TILE]) + (p8[2 * RT * PT3_TILE] + p8[3 * RT * PT3_TILE]);
    s.y += (p8[PT3_COL] + p8[RT * PT3_TILE + PT3_COL]) + (p8[2 * RT * PT3_TILE + PT3_COL] + p8[3 * RT * PT3_TILE + PT3_COL]);
  }
  return s;
}

// The gather phase of half-step (s, q) -- shared by the forward and the BPTT kernel (local names: gx, acc, w, wl_lane, rsrc, gbase,
// glen, asc, ks0, cnt, gpart, dead, rounds, dbgmask).  HAS0: the exchanged vector exists at step 0 (forward with an initial state).
// NSET == 2: the first two chunks of a half-step are normally issued during the previous one (`pre`): after its products every buffer
// is refilled with the chunk two ahead, which for the last two chunks belongs to the half-step executed NEXT -- normally the other set's,
// published a half-step ago.  Sets whose clips are all outside their sequences at a time step do not execute it (lo / hi).
#define DS2R_BASE(S_, Q_) ((Q_) * SETB + (((S_) + 3) & 3) * SLOT + gbase)
// A lane gathers the exchanged vector of ONE clip (row `srow` of the set) at EVERY executed step, also while that clip is outside its
// sequence (glen[q] = T' for a row that exists, 0 otherwise: the loads of a lane without a row go out of range -- zeros, no fabric
// request).  Gathering only the clips inside their sequences (glen = the clip's length) was tried: it saves a fifth to a third of
// the requests on configs 5a / 5b, gains nothing (the half-step is a latency chain) -- and it is WRONG: a workgroup none of whose
// rows is needed stops waiting for its peers, runs ahead of them and re-arms slots they have not read yet (the four-slot protocol
// relies on the lock-step that "every workgroup reads every peer's publish of the step before" enforces).  A group of two short
// clips hit exactly that: peers timed out on a slot its owner had passed (round 4, tests/test_gpu_loop.py lstm_bi_1280).
#define DS2R_T(S_) (asc ? (S_) : Tp - 1 - (S_))
#define DS2R_NEED(S_, Q_) (DS2R_T(S_) < glen[Q_])
#define DS2R_GATHER_PHASE(HAS0)                                                                        \
  {                                                                                                     \
    constexpr int NCH_ = GX::NCH;                                                                       \
    constexpr bool PIPE_ = NSET == 2;                                                                   \
    const bool have_ = (s > lo[q] || (HAS0)) && !(dbgmask & 8);                                         \
    /* the half-step executed next (sets leave / join the schedule: see `lo`, `hi`), and whether its first two chunks may be  \
       started from here: it has data, and its buffer parity continues this one's ((q2 - q - 1) * NCH even) */               \
    int s2 = s, q2 = q;                                                                                 \
    bool pre_next = false;                                                                              \
    if (PIPE_ && have_) {                                                                               \
      bool found_ = false;                                                                              \
      _Pragma("unroll") for (int k_ = 1; k_ < 2 * NSET; ++k_) {                                         \
        const int qq_ = (q + k_) % NSET, ss_ = s + (q + k_) / NSET;                                     \
        if (!found_ && ss_ >= lo[qq_] && ss_ < hi[qq_]) {                                               \
          found_ = true;                                                                                \
          s2 = ss_;                                                                                     \
          q2 = qq_;                                                                                     \
        }                                                                                               \
      }                                                                                                 \
      pre_next = found_ && (s2 > lo[q2] || (HAS0)) && (NCH_ % 2 == 0 || ((q2 - q - 1) & 1) == 0);       \
    }                                                                                                   \
    if (have_) {                                                                                        \
      if (!pre) {                                                                                       \
        gx.issue(PIPE_ ? ((q * NCH_) & 1) : 0, 0, rsrc, DS2R_BASE(s, q), DS2R_NEED(s, q), ks0, cnt, gpart);    \
        gx.issue(PIPE_ ? ((q * NCH_ + 1) & 1) : 1, 1, rsrc, DS2R_BASE(s, q), DS2R_NEED(s, q), ks0, cnt, gpart); \
      }                                                                                                 \
      _Pragma("unroll") for (int c = 0; c < NCH_; ++c) {                                                \
        const int b = PIPE_ ? ((q * NCH_ + c) & 1) : (c & 1);                                           \
        gx.lds_prefetch(c, wl_lane);                                                                    \
        if (__any(gx.bad(b)) && !dead) { /* a sentinel in the chunk: poll it (bounded) before its products */ \
          unsigned spins = 0;                                                                           \
          for (;;) {                                                                                    \
            __builtin_amdgcn_s_sleep(1);                                                                \
            ++rounds;                                                                                   \
            gx.issue(b, c, rsrc, DS2R_BASE(s, q), DS2R_NEED(s, q), ks0, cnt, gpart);                           \
            if (!__any(gx.bad(b))) break;                                                               \
            if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && spin_check(a.lerr, spins))) {          \
              dead = true;                                                                              \
              raise_err(a.err, a.lerr);                                                                 \
              break;                                                                                    \
            }                                                                                           \
          }                                                                                             \
        }                                                                                               \
        gx.mma(b, c, acc, w, wl_lane, dbgmask);                                                         \
        const int c2 = c + 2;                                                                           \
        if (c2 < NCH_) {                                                                                \
          gx.issue(b, c2, rsrc, DS2R_BASE(s, q), DS2R_NEED(s, q), ks0, cnt, gpart);                            \
        } else if (PIPE_ && pre_next) {                                                                 \
          gx.issue(b, c2 - NCH_, rsrc, DS2R_BASE(s2, q2), DS2R_T(s2) < (q2 == 0 ? glen[0] : glen[NSET - 1]), ks0, cnt, gpart); \
        }                                                                                               \
      }                                                                                                 \
    }                                                                                                   \
    pre = pre_next;                                                                                     \
  }

// Samples of a group: i < Ns (clip n = slice + gpd * i); set q owns the CONSECUTIVE samples i = q * RPS + row.  RPS = 16 for two
// sets: set 0 is a full m-tile, set 1 the remainder -- with the batch sorted by length (the reference's loader does that,
// data_loader.py:249) set 1 holds the group's SHORTEST clips and is scheduled for as few steps as possible (a half-step costs the
// same for 6 rows as for 16: it is a latency chain).  sched3: the step range [lo, hi) of the sweep's step counter s in which set q has a clip inside its
// sequence, for a sweep that visits t = s (ascending = true) or t = T' - 1 - s (false): outside it the set's half-steps are not
// executed at all -- no gather, no products, no publish (the carried state of every clip is its initial one there).
template <int NSET, int CAP = 16>
__device__ __forceinline__ void sched3(const int* lens, int slice, int gpd, int Ns, int Tp, bool ascending, bool skip, int (&lo)[NSET],
                                       int (&hi)[NSET], int& RPS) {
  RPS = NSET == 1 ? Ns : min(CAP, Ns);
#pragma unroll
  for (int q = 0; q < NSET; ++q) {
    int mx = 0;
    const int rows = min(RPS, Ns - q * RPS);
    for (int r = 0; r < rows; ++r) mx = max(mx, min(lens[slice + gpd * (q * RPS + r)], Tp));
    if (!skip && rows > 0) mx = Tp;
    mx = __builtin_amdgcn_readfirstlane(mx);
    lo[q] = ascending ? 0 : Tp - mx;
    hi[q] = ascending ? mx : Tp;
  }
}

__device__ __forceinline__ void block_map3(const RArgs& ra, int& grp, int& p) {
  if (ra.xmap) {
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    grp = (i / ra.P) * 8 + x;
    p = i % ra.P;
  } else {
    grp = blockIdx.x % ra.q.NG;
    p = blockIdx.x / ra.q.NG;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward sweep
// ------------------------------------------------------------------------------------------------------------------
template <int CELL, int H, int NSET, int SP, bool SS = false>
__global__ void __launch_bounds__(256, 1) k_rnn_persist3_fwd(RArgs ra) {
  typedef XT<bf16_t> X;
  const QArgs& a = ra.q;
  constexpr int G = CellInfo<CELL>::G, NS = CellInfo<CELL>::NS, M = NS ? NS : 1;
  constexpr int RT = 2 * G;                              // tile 2g + ut = gate g, units 16 ut .. 16 ut + 15 of this workgroup's 32
  constexpr int KT = H / 32, KSW = (KT + 3) / 4;
  constexpr bool RAGGED = KT % 4 != 0;
  constexpr int PS = SS ? 1 : 0;
  constexpr Plan3 PL = plan3(RT, KSW, PS);
  constexpr int PB = PL.PB, P_TR = PL.TR, P_NLDS = PL.nlds;
  constexpr unsigned P_KM = PL.kmask;
  constexpr int SLOT = KT * 1024, SETB = 4 * SLOT;       // bytes of one slot / of one set's four slots (SS uses the first half of a slot)
  static_assert(H % 32 == 0 && PL.ok, "unsupported hidden size");
  static_assert(SP == 1 || NSET == 1, "lane sharing is instantiated for single-set groups only");
  static_assert(!SS || (SP == 1 && KT % 8 == 0), "structured-sparse sets: whole k-blocks per wave, no lane sharing");
  typedef typename std::conditional<SS, Gather3S<RT, KSW, false>, Gather3<RT, KSW, SP, RAGGED, false>>::type GX;
  extern __shared__ __attribute__((aligned(16))) uint4 smem3[];
  uint4* wl = smem3;                                                   // [LDS-resident fragment][4 waves][64 lanes]
  float* part = reinterpret_cast<float*>(smem3 + P_NLDS * 256);       // [PB][4 waves][RT tiles][PT3_TILE]
  constexpr int PART_FLOATS = 4 * RT * PT3_TILE;
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int grp, p;
  block_map3(ra, grp, p);
  if (grp >= a.NG) return;                                 // an empty group slot (fewer samples than groups): the whole workgroup leaves
  const int d = grp / a.gpd, slice = grp % a.gpd;
  const int N = a.N, Tp = a.Tp;
  const int Ns = (N - slice + a.gpd - 1) / a.gpd;          // samples n = slice + gpd*i, i < Ns; set = i % NSET, row = i / NSET
  const int li = lane & 15, lq = lane >> 4;
  constexpr long GH = (long)G * H;
  const long ldgi = (long)a.D * GH;
  const int ks0 = wave * KSW;
  const int cnt = RAGGED ? max(0, min(KSW, KT - ks0)) : KSW;

  typename GX::WArr w;         // only the register-resident fragments are ever touched (the others never become registers)
  {
    const bf16_t* Wd = (const bf16_t*)a.W + (long)d * GH * H;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const bf16_t* row = Wd + ((long)(t >> 1) * H + p * 32 + (t & 1) * 16 + li) * H + lq * 8;
#pragma unroll
      for (int k = 0; k < KSW; ++k) {
        const uint4 f = (!RAGGED || k < cnt) ? *reinterpret_cast<const uint4*>(row + (long)(ks0 + k) * 32) : make_uint4(0, 0, 0, 0);
        if (in_lds3(P_TR, P_KM, t, k, PS))
          wl[lds_index3(P_TR, P_KM, RT, t, k, PS) * 256 + tid] = f;
        else
          GX::set_frag(w, t, k, f);
      }
    }
    GX::template pin<DS2R_VKEEP_FWD>(w);
  }
  const uint4* wl_lane = wl + tid;
  char* xg = a.xbuf + (long)grp * a.xgroup_bytes;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, NSET * SETB, 0x00020000);
  // gather identity of this lane (SS: tile row li carries one half of clip li & 7)
  const int srow = SS ? li : (SP == 2 ? (li & 7) : li), gpart = SP == 2 ? (li >> 3) : 0;
  const int crow = SS ? (li & 7) : srow;                // the clip (row of the set) this lane gathers
  const int gbase = lq * 256 + srow * 16;
  int lo[NSET], hi[NSET], RPS;
  const bool asc = d == 0;                              // the sweep's step s visits t = s (else t = T' - 1 - s)
  sched3<NSET, SS ? 8 : 16>(a.lens, slice, a.gpd, Ns, Tp, asc, (ra.skip & 1) != 0, lo, hi, RPS);
  int glen[NSET];
#pragma unroll
  for (int q = 0; q < NSET; ++q)
    glen[q] = (crow < RPS && q * RPS + crow < Ns) ? Tp : 0;

  // ---- gate identity: thread bits (dw: unit pair of the 16-byte unit (2), sample row (4), lq (2)) = the exchange layout's order
  const int lqw = tid >> 6, grow = (tid >> 2) & 15, dw = tid & 3;
  const int jl = lqw * 8 + dw * 2, j = p * 32 + jl;                   // units (j, j + 1)
  const int xoff = SS ? xsp_pair_bytes(j, grow & 7)                   // this pair's dword inside a slot: sparse layout,
                      : ((p * 4 + lqw) * 16 + grow) * 16 + dw * 4;    // dense: k-step p
  bool on[NSET];
  int len[NSET];
  float hprev[NSET][2], cprev[NSET][2], bh[G][2];
  const bf16_t* gi_ptr[NSET];      // where the NEXT prefetch of the set's input projection reads
  bf16_t* sv_ptr[NSET];
  bf16_t* hs_ptr[NSET];
  const long dstep = d == 0 ? 1 : -1;
  constexpr long NSH_ = (long)M * H;
  int nsmp[NSET];
#pragma unroll
  for (int q = 0; q < NSET; ++q) {
    const int t_first = d == 0 ? lo[q] : Tp - 1 - lo[q];      // the set's first executed step
    const int i = q * RPS + grow;
    on[q] = grow < RPS && i < Ns && grow < (SS ? 8 : 16 / SP);
    const int n = on[q] ? slice + a.gpd * i : 0;
    nsmp[q] = n;
    len[q] = on[q] ? a.lens[n] : 0;
    hprev[q][0] = hprev[q][1] = cprev[q][0] = cprev[q][1] = 0.f;
    if (on[q]) {
      const long so = ((long)d * N + n) * H + j;
      if (a.h0) {
        hprev[q][0] = a.h0[so];
        hprev[q][1] = a.h0[so + 1];
      }
      if (CELL == CELL_LSTM && a.c0) {
        cprev[q][0] = a.c0[so];
        cprev[q][1] = a.c0[so + 1];
      }
    }
    gi_ptr[q] = (const bf16_t*)a.GI + ((long)t_first * N + n) * ldgi + (long)d * GH + j;
    sv_ptr[q] = NS ? (bf16_t*)a.S + (((long)d * Tp + t_first) * N + n) * NSH_ + j : nullptr;
    hs_ptr[q] = (bf16_t*)a.Hseq + (long)d * a.hseq_dstride + ((long)t_first * N + n) * H + j;
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    bh[g][0] = a.bhh[(long)d * GH + (long)g * H + j];
    bh[g][1] = a.bhh[(long)d * GH + (long)g * H + j + 1];
  }
  const long gi_stride = dstep * N * ldgi, sv_stride = dstep * N * NSH_, hs_stride = dstep * N * H;
  bool dead = false;
  bool local = false;
  if (ra.xmap)
    local = group_is_xcd_local3(ra.xcc + grp * 32, p, ra.P, tid, a.err, a.lerr, a.startup_ms, dead);
  else
    wait_all_resident(a.lerr, tid, a.err, a.startup_ms, dead);    // groups that span XCDs: no XCC-id handshake, one arrival word
#pragma unroll
  for (int q = 0; q < NSET; ++q)
    if (on[q]) {   // zero guard slots of the state sequence at t = -1 and t = T' ("previous h" reads are unconditional)
      bf16_t* hb = (bf16_t*)a.Hseq + (long)d * a.hseq_dstride + (long)nsmp[q] * H + j;
      X::st(hb - (long)N * H, 0.f, 0.f);
      X::st(hb + (long)Tp * N * H, 0.f, 0.f);
      // the step "before the set's first": slot (lo + 3) & 3
      if (a.h0) pub32(xg + q * SETB + ((lo[q] + 3) & 3) * SLOT + xoff, pay_word(hprev[q][0], hprev[q][1]), local);
    }
  __syncthreads();      // the LDS-resident fragments are in place

  // The input projection of a set's step is loaded DEP steps of that set ahead, right BEHIND the gather phase of a half-step (vmcnt
  // retires in order: these are HBM misses, and a gather issued behind one would wait for it; profiles/r04d_ablation_cfg5a.txt: with
  // the loads one half-step ahead and in front of the gather they cost 0.7 us per forward and 1.7 us per BPTT time step).
  constexpr int DEP = NSET == 1 ? 2 : 1;
  uint32_t gir[NSET][DEP][G];
#pragma unroll
  for (int q = 0; q < NSET; ++q)
#pragma unroll
    for (int i = 0; i < DEP; ++i) {
#pragma unroll
      for (int g = 0; g < G; ++g) gir[q][i][g] = (on[q] && lo[q] + i < hi[q]) ? X::ld(gi_ptr[q] + (long)g * H) : 0u;
      gi_ptr[q] += gi_stride;
    }
  int hstep = 0;
  unsigned rounds = 0;
  DS2_PROBE_ONLY(unsigned long long c_gather = 0, c_bar = 0, c_gate = 0; const int dbgmask = ra.dbgmask;)
#ifndef DS2_PROBE
  constexpr int dbgmask = DS2R_VAR;
#endif
  const bool plain = local || (dbgmask & 64);
  GX gx;
  gx.spidx = (li & 8) ? 0xEEEE : 0x4444;
  bool pre = false;             // the first two chunks of the half-step about to run are in flight (NSET == 2)
  int s_lo = lo[0], s_hi = hi[0];
#pragma unroll
  for (int q = 1; q < NSET; ++q) {
    s_lo = min(s_lo, lo[q]);
    s_hi = max(s_hi, hi[q]);
  }
  for (int s = s_lo; s < s_hi; ++s) {
    const int t = d == 0 ? s : Tp - 1 - s;
#pragma unroll
    for (int q = 0; q < NSET; ++q) {
      if (s < lo[q] || s >= hi[q]) continue;     // no clip of this set is inside its sequence at t
      DS2_PROBE_ONLY(const unsigned long long t0 = __builtin_readcyclecounter();)
      uint32_t gi[G];
#pragma unroll
      for (int g = 0; g < G; ++g) gi[g] = gir[q][0][g];
      ds2_f32x4 acc[RT];
#pragma unroll
      for (int tt = 0; tt < RT; ++tt) acc[tt] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
      DS2R_GATHER_PHASE(a.h0 != nullptr)
      mfma_results_ready<RT>(acc);
      {   // this set's input projection DEP steps ahead
#pragma unroll
        for (int i = 0; i + 1 < DEP; ++i)
#pragma unroll
          for (int g = 0; g < G; ++g) gir[q][i][g] = gir[q][i + 1][g];
#pragma unroll
        for (int g = 0; g < G; ++g)
          gir[q][DEP - 1][g] = (on[q] && s + DEP < hi[q] && !(dbgmask & 1)) ? X::ld(gi_ptr[q] + (long)g * H) : 0u;
        gi_ptr[q] += gi_stride;
      }
      DS2_PROBE_ONLY(const unsigned long long t1 = __builtin_readcyclecounter();)
      float* pp = part + (PB == 2 ? (hstep & 1) * PART_FLOATS : 0);
      if (PB == 1) __syncthreads();       // every wave is through with the previous half-step's partial sums
      store_partials3<RT>(pp, acc, wave, lane);
      __syncthreads();
      DS2_PROBE_ONLY(const unsigned long long t2 = __builtin_readcyclecounter();)
      float hn0 = 0.f, hn1 = 0.f;         // emitted h_t (0 when inactive)
      float pl[M][2];
#pragma unroll
      for (int m = 0; m < M; ++m) pl[m][0] = pl[m][1] = 0.f;
      if (on[q]) {
        const bool act = t < len[q] && !(dbgmask & 256);
        float2 gh[G];
#pragma unroll
        for (int g = 0; g < G; ++g) gh[g] = load_partials3<RT, SS>(pp, 2 * g + (jl >> 4), grow, jl & 15);
        if (CELL == CELL_GRU) {
          if (act) {
            const float q0 = gh[2 % G].x + bh[2 % G][0], q1 = gh[2 % G].y + bh[2 % G][1];
            const float r0 = X::sig(X::lo(gi[0]) + gh[0].x + bh[0][0]), r1 = X::sig(X::hi(gi[0]) + gh[0].y + bh[0][1]);
            const float z0 = X::sig(X::lo(gi[1 % G]) + gh[1 % G].x + bh[1 % G][0]);
            const float z1 = X::sig(X::hi(gi[1 % G]) + gh[1 % G].y + bh[1 % G][1]);
            const float n0 = X::tnh(X::lo(gi[2 % G]) + r0 * q0), n1 = X::tnh(X::hi(gi[2 % G]) + r1 * q1);
            hn0 = (1.f - z0) * n0 + z0 * hprev[q][0];
            hn1 = (1.f - z1) * n1 + z1 * hprev[q][1];
            hprev[q][0] = hn0;
            hprev[q][1] = hn1;
            pl[0][0] = r0; pl[0][1] = r1;
            pl[1 % M][0] = z0; pl[1 % M][1] = z1;
            pl[2 % M][0] = n0; pl[2 % M][1] = n1;
            pl[3 % M][0] = q0; pl[3 % M][1] = q1;
          }
        } else if (CELL == CELL_LSTM) {
          if (act) {
            const float i0 = X::sig(X::lo(gi[0]) + gh[0].x + bh[0][0]), i1 = X::sig(X::hi(gi[0]) + gh[0].y + bh[0][1]);
            const float f0 = X::sig(X::lo(gi[1 % G]) + gh[1 % G].x + bh[1 % G][0]);
            const float f1 = X::sig(X::hi(gi[1 % G]) + gh[1 % G].y + bh[1 % G][1]);
            const float g0 = X::tnh(X::lo(gi[2 % G]) + gh[2 % G].x + bh[2 % G][0]);
            const float g1 = X::tnh(X::hi(gi[2 % G]) + gh[2 % G].y + bh[2 % G][1]);
            const float o0 = X::sig(X::lo(gi[3 % G]) + gh[3 % G].x + bh[3 % G][0]);
            const float o1 = X::sig(X::hi(gi[3 % G]) + gh[3 % G].y + bh[3 % G][1]);
            const float c0 = f0 * cprev[q][0] + i0 * g0, c1 = f1 * cprev[q][1] + i1 * g1;
            hn0 = o0 * X::tnh(c0);
            hn1 = o1 * X::tnh(c1);
            cprev[q][0] = c0;
            cprev[q][1] = c1;
            hprev[q][0] = hn0;
            hprev[q][1] = hn1;
            pl[0][0] = i0; pl[0][1] = i1;
            pl[1 % M][0] = f0; pl[1 % M][1] = f1;
            pl[2 % M][0] = g0; pl[2 % M][1] = g1;
            pl[3 % M][0] = o0; pl[3 % M][1] = o1;
            pl[4 % M][0] = c0; pl[4 % M][1] = c1;
          }
        } else {
          if (act) {
            hn0 = X::tnh(X::lo(gi[0]) + gh[0].x + bh[0][0]);
            hn1 = X::tnh(X::hi(gi[0]) + gh[0].y + bh[0][1]);
            hprev[q][0] = hn0;
            hprev[q][1] = hn1;
          }
        }
        if (dead) hn0 = hn1 = hprev[q][0] = hprev[q][1] = QNAN;   // fail loudly downstream
      }
      {
        // the quad's four dwords meet on its lane 0 (all lanes active: rows are uniform within a quad), which publishes the carried
        // state first (inactive samples republish their unchanged state), then re-arms and stores the bookkeeping planes
        const u32x4_t pubv = quad_gather(pay_word(hprev[q][0], hprev[q][1]));
        const u32x4_t hsv = quad_gather(cvt_pk_bf16(hn0, hn1));
        u32x4_t plv[M];
#pragma unroll
        for (int m = 0; m < M; ++m) plv[m] = quad_gather(cvt_pk_bf16(pl[m][0], pl[m][1]));
#ifndef DS2R_QUAD_STORES       // A/B variant: every lane stores its own dword (the form before the quad gathers)
        if (on[q]) {
          pub32(xg + q * SETB + (s & 3) * SLOT + xoff, pubv[0], plain);
          pub32(xg + q * SETB + ((s + 2) & 3) * SLOT + xoff, XSENT2, plain);
          if (!(dbgmask & 2)) {
            *reinterpret_cast<uint32_t*>(hs_ptr[q]) = hsv[0];
#pragma unroll
            for (int m = 0; m < NS; ++m) *reinterpret_cast<uint32_t*>(sv_ptr[q] + (long)m * H) = plv[m][0];
          }
        }
#else
        if (on[q] && dw == 0) {
          pub128(rsrc, q * SETB + (s & 3) * SLOT + xoff, pubv, plain);
          pub128(rsrc, q * SETB + ((s + 2) & 3) * SLOT + xoff, u32x4_t{XSENT2, XSENT2, XSENT2, XSENT2}, plain);   // re-arm step s + 2's slot
          if (!(dbgmask & 2)) {
            st128(hs_ptr[q], hsv);
#pragma unroll
            for (int m = 0; m < NS; ++m) st128(sv_ptr[q] + (long)m * H, plv[m]);
          }
        }
#endif
      }
      if (NS) sv_ptr[q] += sv_stride;
      hs_ptr[q] += hs_stride;
      DS2_PROBE_ONLY(const unsigned long long t3 = __builtin_readcyclecounter(); c_gather += t1 - t0; c_bar += t2 - t1; c_gate += t3 - t2;)
      ++hstep;
    }
  }
#ifdef DS2_PROBE
  if (ra.dbg && p == 0 && (tid == 0 || tid == 255)) {
    const int o = grp * 8 + (tid == 0 ? 0 : 4);
    ra.dbg[o + 0] = c_gather;
    ra.dbg[o + 1] = c_bar;
    ra.dbg[o + 2] = c_gate;
    ra.dbg[o + 3] = rounds;
  }
#endif
  (void)rounds;
#pragma unroll
  for (int q = 0; q < NSET; ++q) {
    if (on[q]) {
      const long so = ((long)d * N + nsmp[q]) * H + j;
      if (a.hn) {
        a.hn[so] = hprev[q][0];
        a.hn[so + 1] = hprev[q][1];
      }
      if (CELL == CELL_LSTM && a.cn) {
        a.cn[so] = cprev[q][0];
        a.cn[so + 1] = cprev[q][1];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BPTT sweep.  dh_t = dOut[t] + carry (elementwise part of dh from the step processed before) + dgates_{t'} * W_hh.
// The workgroup owns the W_hh^T rows of its 32 units (two 16-row tiles), K = G*H: gate g's element j is k = g*H + j.
// ------------------------------------------------------------------------------------------------------------------
template <int CELL, int H, int NSET, int SP, bool SS = false>
__global__ void __launch_bounds__(256, 1) k_rnn_persist3_bwd(RArgs ra) {
  typedef XT<bf16_t> X;
  const QArgs& a = ra.q;
  constexpr int G = CellInfo<CELL>::G, NS = CellInfo<CELL>::NS, M = NS ? NS : 1;
  constexpr int RT = 2;
  constexpr int KTH = H / 32, KT = G * KTH, KSW = (KT + 3) / 4;
  constexpr bool RAGGED = KT % 4 != 0;
  constexpr int PS = SS ? 1 : 0;
  constexpr Plan3 PL = plan3(RT, KSW, PS);
  constexpr int PB = PL.PB, P_TR = PL.TR, P_NLDS = PL.nlds;
  constexpr unsigned P_KM = PL.kmask;
  constexpr int SLOT = KT * 1024, SETB = 4 * SLOT, GATEB = SS ? KTH * 512 : KTH * 1024;   // gate g's k = g * H + j
  static_assert(H % 32 == 0 && PL.ok, "unsupported hidden size");
  static_assert(SP == 1 || NSET == 1, "lane sharing is instantiated for single-set groups only");
  static_assert(!SS || (SP == 1 && KT % 8 == 0 && H % 64 == 0), "structured-sparse sets: whole k-blocks per wave and gate");
  typedef typename std::conditional<SS, Gather3S<RT, KSW, true>, Gather3<RT, KSW, SP, RAGGED, true>>::type GX;
  extern __shared__ __attribute__((aligned(16))) uint4 smem3[];
  uint4* wl = smem3;
  float* part = reinterpret_cast<float*>(smem3 + P_NLDS * 256);
  constexpr int PART_FLOATS = 4 * RT * PT3_TILE;
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int grp, p;
  block_map3(ra, grp, p);
  if (grp >= a.NG) return;                                 // an empty group slot (fewer samples than groups): the whole workgroup leaves
  const int d = grp / a.gpd, slice = grp % a.gpd;
  const int N = a.N, Tp = a.Tp;
  const int Ns = (N - slice + a.gpd - 1) / a.gpd;
  const int li = lane & 15, lq = lane >> 4;
  constexpr long GH = (long)G * H;
  const long ldgi = (long)a.D * GH;
  const int ks0 = wave * KSW;
  const int cnt = RAGGED ? max(0, min(KSW, KT - ks0)) : KSW;

  typename GX::WArr w;         // only the register-resident fragments are ever touched (the others never become registers)
  {
    const bf16_t* WT = (const bf16_t*)a.W + (long)d * H * GH;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const bf16_t* row = WT + (long)(p * 32 + t * 16 + li) * GH + lq * 8;
#pragma unroll
      for (int k = 0; k < KSW; ++k) {
        const uint4 f = (!RAGGED || k < cnt) ? *reinterpret_cast<const uint4*>(row + (long)(ks0 + k) * 32) : make_uint4(0, 0, 0, 0);
        if (in_lds3(P_TR, P_KM, t, k, PS))
          wl[lds_index3(P_TR, P_KM, RT, t, k, PS) * 256 + tid] = f;
        else
          GX::set_frag(w, t, k, f);
      }
    }
    GX::template pin<DS2R_VKEEP_BWD>(w);
  }
  const uint4* wl_lane = wl + tid;
  char* xg = a.xbuf + (long)grp * a.xgroup_bytes;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, NSET * SETB, 0x00020000);
  const int srow = SS ? li : (SP == 2 ? (li & 7) : li), gpart = SP == 2 ? (li >> 3) : 0;
  const int crow = SS ? (li & 7) : srow;                // the clip (row of the set) this lane gathers
  const int gbase = lq * 256 + srow * 16;
  int lo[NSET], hi[NSET], RPS;
  const bool asc = d != 0;                              // BPTT visits t = T' - 1 - s for direction 0
  sched3<NSET, SS ? 8 : 16>(a.lens, slice, a.gpd, Ns, Tp, asc, (ra.skip & 1) != 0, lo, hi, RPS);
  int glen[NSET];
#pragma unroll
  for (int q = 0; q < NSET; ++q)
    glen[q] = (crow < RPS && q * RPS + crow < Ns) ? Tp : 0;

  const int lqw = tid >> 6, grow = (tid >> 2) & 15, dw = tid & 3;
  const int jl = lqw * 8 + dw * 2, j = p * 32 + jl;
  const int xoff = SS ? xsp_pair_bytes(j, grow & 7)                   // gate 0's dword of this pair inside a slot; gate g: + g * GATEB
                      : ((p * 4 + lqw) * 16 + grow) * 16 + dw * 4;
  constexpr int NB = CELL == CELL_GRU ? 4 : G;
  bool on[NSET];
  int len[NSET], nsmp[NSET];
  float car[NSET][2], dc[NSET][2], bsum[NSET][NB][2];
  const bf16_t* do_ptr[NSET];      // the three read pointers run one half-step AHEAD (next prefetch)
  const bf16_t* sv_ptr[NSET];
  const bf16_t* hs_ptr[NSET];
  int tnext[NSET];                 // time index the read pointers of the set stand at
  bf16_t* dgi_ptr[NSET];
  bf16_t* dgh_ptr[NSET];
  const long dstep = d == 0 ? -1 : 1;                       // BPTT walks the direction's time axis backwards
  const long prev_off = d == 0 ? -1 : 1;                    // previous step in FORWARD order of this direction
  constexpr long NSH_ = (long)M * H;
#pragma unroll
  for (int q = 0; q < NSET; ++q) {
    const int t_first = d == 0 ? Tp - 1 - lo[q] : lo[q];      // the set's first executed step
    const int i = q * RPS + grow;
    on[q] = grow < RPS && i < Ns && grow < (SS ? 8 : 16 / SP);
    const int n = on[q] ? slice + a.gpd * i : 0;
    nsmp[q] = n;
    len[q] = on[q] ? a.lens[n] : 0;
    car[q][0] = car[q][1] = dc[q][0] = dc[q][1] = 0.f;
#pragma unroll
    for (int g = 0; g < NB; ++g) bsum[q][g][0] = bsum[q][g][1] = 0.f;
    do_ptr[q] = (const bf16_t*)a.dOut + ((long)t_first * N + n) * H + j;
    sv_ptr[q] = NS ? (const bf16_t*)a.S + (((long)d * Tp + t_first) * N + n) * NSH_ + j : nullptr;
    hs_ptr[q] = (const bf16_t*)a.Hseq + (long)d * a.hseq_dstride + ((long)t_first * N + n) * H + j;   // h_t
    tnext[q] = t_first;
    dgi_ptr[q] = (bf16_t*)a.dGI + ((long)t_first * N + n) * ldgi + (long)d * GH + j;
    dgh_ptr[q] = a.dGH ? (bf16_t*)a.dGH + (((long)d * Tp + t_first) * N + n) * H + j : nullptr;     // dQ
  }
  bool dead = false;
  bool local = false;
  if (ra.xmap)
    local = group_is_xcd_local3(ra.xcc + grp * 32, p, ra.P, tid, a.err, a.lerr, a.startup_ms, dead);
  else
    wait_all_resident(a.lerr, tid, a.err, a.startup_ms, dead);    // groups that span XCDs: no XCC-id handshake, one arrival word
  __syncthreads();

  DS2_PROBE_ONLY(const int dbgmask = ra.dbgmask;)
#ifndef DS2_PROBE
  constexpr int dbgmask = DS2R_VAR;
#endif
  // gate-phase operands of a half-step, loaded one half-step ahead (raw pairs)
  struct Pre {
    uint32_t dout, sp[M], hp, cp;
  };
  auto prefetch = [&](Pre& r, int q, bool valid) {   // reads the set's pointers (they stand at tnext[q]) and advances them
    r.dout = r.hp = r.cp = 0u;
#pragma unroll
    for (int m = 0; m < M; ++m) r.sp[m] = 0u;
    if (on[q] && valid && !(dbgmask & 1)) {
      const int t = tnext[q];
      r.dout = X::ld(do_ptr[q]);
#pragma unroll
      for (int m = 0; m < NS; ++m) r.sp[m] = X::ld(sv_ptr[q] + (long)m * H);
      r.hp = X::ld(hs_ptr[q] + prev_off * N * H);     // guard slots / inactive frames hold zeros: unconditional (tprev in [-1, Tp])
      if (CELL == CELL_LSTM) {
        const bool has_prev = d == 0 ? (t > 0) : (t + 1 < len[q]);
        if (has_prev) r.cp = X::ld(sv_ptr[q] + prev_off * N * NSH_ + 4 * H);
      }
      if (CELL == CELL_RNN) r.hp = X::ld(hs_ptr[q]);
    }
    do_ptr[q] += dstep * N * H;
    if (NS) sv_ptr[q] += dstep * N * NSH_;
    hs_ptr[q] += dstep * N * H;
    tnext[q] += (int)dstep;
  };
  // ... DEP steps of the set ahead, right behind the gather phase (see the forward kernel)
  constexpr int DEP = NSET == 1 ? 2 : 1;
  Pre ring[NSET][DEP];
  unsigned rounds = 0;
  DS2_PROBE_ONLY(unsigned long long c_gather = 0, c_bar = 0, c_gate = 0;)
  const bool plain = local || (dbgmask & 64);
  GX gx;
  gx.spidx = (li & 8) ? 0xEEEE : 0x4444;
#pragma unroll
  for (int q = 0; q < NSET; ++q)
#pragma unroll
    for (int i = 0; i < DEP; ++i) prefetch(ring[q][i], q, lo[q] + i < hi[q]);
  int hstep = 0;
  bool pre = false;             // the first two chunks of the half-step about to run are in flight (NSET == 2)
  int s_lo = lo[0], s_hi = hi[0];
#pragma unroll
  for (int q = 1; q < NSET; ++q) {
    s_lo = min(s_lo, lo[q]);
    s_hi = max(s_hi, hi[q]);
  }
  for (int s = s_lo; s < s_hi; ++s) {
    const int t = d == 0 ? Tp - 1 - s : s;
#pragma unroll
    for (int q = 0; q < NSET; ++q) {
      if (s < lo[q] || s >= hi[q]) continue;     // no clip of this set is inside its sequence at t
      DS2_PROBE_ONLY(const unsigned long long t0 = __builtin_readcyclecounter();)
      const Pre pc = ring[q][0];
      ds2_f32x4 acc2[2 * RT];
#pragma unroll
      for (int tt = 0; tt < 2 * RT; ++tt) acc2[tt] = ds2_f32x4{0.f, 0.f, 0.f, 0.f};
      {
        ds2_f32x4 (&acc)[2 * RT] = acc2;
        DS2R_GATHER_PHASE(false)
      }
      mfma_results_ready<2 * RT>(acc2);
      ds2_f32x4 acc[RT];
#pragma unroll
      for (int tt = 0; tt < RT; ++tt) acc[tt] = acc2[tt] + acc2[tt + RT];
#pragma unroll
      for (int i = 0; i + 1 < DEP; ++i) ring[q][i] = ring[q][i + 1];
      prefetch(ring[q][DEP - 1], q, s + DEP < hi[q]);
      DS2_PROBE_ONLY(const unsigned long long t1 = __builtin_readcyclecounter();)
      float* pp = part + (PB == 2 ? (hstep & 1) * PART_FLOATS : 0);
      if (PB == 1) __syncthreads();
      store_partials3<RT>(pp, acc, wave, lane);
      __syncthreads();
      DS2_PROBE_ONLY(const unsigned long long t2 = __builtin_readcyclecounter();)
      float gx[G][2], gn[2] = {0.f, 0.f};         // gx: the exchanged planes (GRU: dr, dz, dq), gn: GRU's dn (stored, not exchanged)
#pragma unroll
      for (int g = 0; g < G; ++g) gx[g][0] = gx[g][1] = 0.f;
      if (on[q]) {
        const bool act = t < len[q] && !(dbgmask & 256);
        const float2 mp = load_partials3<RT, SS>(pp, jl >> 4, grow, jl & 15);
        const float din0 = car[q][0] + mp.x, din1 = car[q][1] + mp.y;
        car[q][0] = din0;
        car[q][1] = din1;
        if (CELL == CELL_GRU) {
          if (act) {
            const float r0 = X::lo(pc.sp[0]), r1 = X::hi(pc.sp[0]), z0 = X::lo(pc.sp[1 % M]), z1 = X::hi(pc.sp[1 % M]);
            const float n0 = X::lo(pc.sp[2 % M]), n1 = X::hi(pc.sp[2 % M]), q0 = X::lo(pc.sp[3 % M]), q1 = X::hi(pc.sp[3 % M]);
            const float dh0 = X::lo(pc.dout) + din0, dh1 = X::hi(pc.dout) + din1;
            gn[0] = dh0 * (1.f - z0) * (1.f - n0 * n0);
            gn[1] = dh1 * (1.f - z1) * (1.f - n1 * n1);
            gx[1 % G][0] = dh0 * (X::lo(pc.hp) - n0) * z0 * (1.f - z0);
            gx[1 % G][1] = dh1 * (X::hi(pc.hp) - n1) * z1 * (1.f - z1);
            gx[0][0] = gn[0] * q0 * r0 * (1.f - r0);
            gx[0][1] = gn[1] * q1 * r1 * (1.f - r1);
            gx[2 % G][0] = gn[0] * r0;
            gx[2 % G][1] = gn[1] * r1;
            car[q][0] = dh0 * z0;
            car[q][1] = dh1 * z1;
          }
        } else if (CELL == CELL_LSTM) {
          if (act) {
            const float i0 = X::lo(pc.sp[0]), i1 = X::hi(pc.sp[0]), f0 = X::lo(pc.sp[1 % M]), f1 = X::hi(pc.sp[1 % M]);
            const float g0 = X::lo(pc.sp[2 % M]), g1 = X::hi(pc.sp[2 % M]), o0 = X::lo(pc.sp[3 % M]), o1 = X::hi(pc.sp[3 % M]);
            const float tc0 = X::tnh(X::lo(pc.sp[4 % M])), tc1 = X::tnh(X::hi(pc.sp[4 % M]));
            const float dh0 = X::lo(pc.dout) + din0, dh1 = X::hi(pc.dout) + din1;
            const float dcn0 = dc[q][0] + dh0 * o0 * (1.f - tc0 * tc0), dcn1 = dc[q][1] + dh1 * o1 * (1.f - tc1 * tc1);
            gx[0][0] = dcn0 * g0 * i0 * (1.f - i0);
            gx[0][1] = dcn1 * g1 * i1 * (1.f - i1);
            gx[1 % G][0] = dcn0 * X::lo(pc.cp) * f0 * (1.f - f0);
            gx[1 % G][1] = dcn1 * X::hi(pc.cp) * f1 * (1.f - f1);
            gx[2 % G][0] = dcn0 * i0 * (1.f - g0 * g0);
            gx[2 % G][1] = dcn1 * i1 * (1.f - g1 * g1);
            gx[3 % G][0] = dh0 * tc0 * o0 * (1.f - o0);
            gx[3 % G][1] = dh1 * tc1 * o1 * (1.f - o1);
            car[q][0] = car[q][1] = 0.f;
            dc[q][0] = dcn0 * f0;
            dc[q][1] = dcn1 * f1;
          }
        } else {
          if (act) {
            const float h0v = X::lo(pc.hp), h1v = X::hi(pc.hp);
            gx[0][0] = (X::lo(pc.dout) + din0) * (1.f - h0v * h0v);
            gx[0][1] = (X::hi(pc.dout) + din1) * (1.f - h1v * h1v);
            car[q][0] = car[q][1] = 0.f;
          }
        }
        if (dead) gx[0][0] = gx[0][1] = QNAN;
      }
      {
        // packed planes; the quad's four dwords meet on its lane 0 (all lanes active), which publishes, re-arms and stores 16 bytes at
        // a time; every lane keeps the bias sums of its own pair from the ROUNDED values (= the column sums of the stored planes)
        uint32_t pk[G];
#pragma unroll
        for (int g = 0; g < G; ++g) pk[g] = cvt_pk_bf16(gx[g][0], gx[g][1]);
        const uint32_t pkn = cvt_pk_bf16(gn[0], gn[1]);
        u32x4_t pubv[G], stv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          stv[g] = quad_gather(pk[g]);
          pubv[g] = quad_gather(pk[g] == XSENT2 ? 0x7fc07fc0u : pk[g]);
        }
        const u32x4_t nv = quad_gather(pkn);
        if (on[q]) {
#ifndef DS2R_QUAD_STORES
          {
            char* xo = xg + q * SETB + (s & 3) * SLOT + xoff;
            char* xr = xg + q * SETB + ((s + 2) & 3) * SLOT + xoff;
#pragma unroll
            for (int g = 0; g < G; ++g) pub32(xo + g * GATEB, pubv[g][0], plain);
#pragma unroll
            for (int g = 0; g < G; ++g) pub32(xr + g * GATEB, XSENT2, plain);
            if (!(dbgmask & 2)) {
              bf16_t* dgi = dgi_ptr[q];
              if (CELL == CELL_GRU) {
                *reinterpret_cast<uint32_t*>(dgi) = pk[0];
                *reinterpret_cast<uint32_t*>(dgi + H) = pk[1 % G];
                *reinterpret_cast<uint32_t*>(dgi + 2 * H) = pkn;
                *reinterpret_cast<uint32_t*>(dgh_ptr[q]) = pk[2 % G];
              } else {
#pragma unroll
                for (int g = 0; g < G; ++g) *reinterpret_cast<uint32_t*>(dgi + (long)g * H) = pk[g];
              }
            }
          }
          if (false) {
            const int xo = 0, xr = 0;
#pragma unroll
            for (int g = 0; g < G; ++g) pub128(rsrc, xo + g * GATEB, pubv[g], plain);
#else
          if (dw == 0) {
            const int xo = q * SETB + (s & 3) * SLOT + xoff, xr = q * SETB + ((s + 2) & 3) * SLOT + xoff;   // xr: re-armed for step s + 2
#pragma unroll
            for (int g = 0; g < G; ++g) pub128(rsrc, xo + g * GATEB, pubv[g], plain);
#endif
#pragma unroll
            for (int g = 0; g < G; ++g) pub128(rsrc, xr + g * GATEB, u32x4_t{XSENT2, XSENT2, XSENT2, XSENT2}, plain);
            if (!(dbgmask & 2)) {
              bf16_t* dgi = dgi_ptr[q];
              if (CELL == CELL_GRU) {     // dGI = [dr, dz, dn], dQ apart
                st128(dgi, stv[0]);
                st128(dgi + H, stv[1 % G]);
                st128(dgi + 2 * H, nv);
                st128(dgh_ptr[q], stv[2 % G]);
              } else {
#pragma unroll
                for (int g = 0; g < G; ++g) st128(dgi + (long)g * H, stv[g]);
              }
            }
          }
          if (CELL == CELL_GRU) {
            bsum[q][0][0] += bf_lo(pk[0]); bsum[q][0][1] += bf_hi(pk[0]);
            bsum[q][1 % NB][0] += bf_lo(pk[1 % G]); bsum[q][1 % NB][1] += bf_hi(pk[1 % G]);
            bsum[q][2 % NB][0] += bf_lo(pkn); bsum[q][2 % NB][1] += bf_hi(pkn);
            bsum[q][3 % NB][0] += bf_lo(pk[2 % G]); bsum[q][3 % NB][1] += bf_hi(pk[2 % G]);
          } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
              bsum[q][g % NB][0] += bf_lo(pk[g]);
              bsum[q][g % NB][1] += bf_hi(pk[g]);
            }
          }
        }
      }
      dgi_ptr[q] += dstep * N * ldgi;
      if (CELL == CELL_GRU) dgh_ptr[q] += dstep * N * H;
      DS2_PROBE_ONLY(const unsigned long long t3 = __builtin_readcyclecounter(); c_gather += t1 - t0; c_bar += t2 - t1; c_gate += t3 - t2;)
      ++hstep;
    }
  }
#ifdef DS2_PROBE
  if (ra.dbg && p == 0 && (tid == 0 || tid == 255)) {
    const int o = grp * 8 + (tid == 0 ? 0 : 4);
    ra.dbg[o + 0] = c_gather;
    ra.dbg[o + 1] = c_bar;
    ra.dbg[o + 2] = c_gate;
    ra.dbg[o + 3] = rounds;
  }
#endif
  (void)rounds;
  if (a.dBacc) {
#pragma unroll
    for (int q = 0; q < NSET; ++q) {
      if (on[q]) {
        float* bo = a.dBacc + ((long)d * N + nsmp[q]) * NB * H + j;
#pragma unroll
        for (int g = 0; g < NB; ++g) *reinterpret_cast<float2*>(bo + (long)g * H) = make_float2(bsum[q][g][0], bsum[q][g][1]);
      }
    }
  }
}

template <int CELL, int H>
constexpr bool covered3() {
  constexpr int G = CellInfo<CELL>::G;
  return H % 32 == 0 && plan3(2 * G, (H / 32 + 3) / 4).ok && plan3(2, (G * H / 32 + 3) / 4).ok;
}

// Structured-sparse sets (Gather3S): every wave's K-quarter is a whole number of k-blocks of 64, forward (K = H) and BPTT (K = G * H)
template <int CELL, int H>
constexpr bool sparse3() {
  constexpr int G = CellInfo<CELL>::G;
  return H % 256 == 0 && covered3<CELL, H>() && plan3(2 * G, H / 128, 1).ok && plan3(2, G * H / 128, 1).ok;
}

template <int CELL, int H, int NSET, int SP, bool SS = false>
int launch3_one(bool bwd, const RArgs& ra, hipStream_t st) {
  constexpr int G = CellInfo<CELL>::G;
  if constexpr (!covered3<CELL, H>() || (SS && !sparse3<CELL, H>())) {
    return DS2_ERR_ARG;
  } else {
    const size_t shm = bwd ? (size_t)lds_bytes3(2, (G * H / 32 + 3) / 4, SS ? 1 : 0) : (size_t)lds_bytes3(2 * G, (H / 32 + 3) / 4, SS ? 1 : 0);
    static bool attr[2][DS2_MAX_DEVICES];
    const void* fn = bwd ? (const void*)k_rnn_persist3_bwd<CELL, H, NSET, SP, SS> : (const void*)k_rnn_persist3_fwd<CELL, H, NSET, SP, SS>;
    if (ds2_first_use_on_device(attr[bwd ? 1 : 0])) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    const int grid = ra.xmap ? 8 * ra.gx * ra.P : ra.q.NG * ra.P;
    if (bwd)
      hipLaunchKernelGGL((k_rnn_persist3_bwd<CELL, H, NSET, SP, SS>), dim3(grid), dim3(256), shm, st, ra);
    else
      hipLaunchKernelGGL((k_rnn_persist3_fwd<CELL, H, NSET, SP, SS>), dim3(grid), dim3(256), shm, st, ra);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
  }
}

// nset = sample sets (1 or 2).  ra.sparse (the host's choice for groups of <= 16 clips at the widths sparse3 covers): sets of <= 8 clips
// on the structured-sparse products; otherwise single-set groups of <= 8 samples share the gather loads between lane pairs (SP = 2).
// probe: is the combination instantiated?
template <int CELL, int H>
int launch3(bool probe, bool bwd, const RArgs& ra, hipStream_t st) {
  if (!covered3<CELL, H>()) return DS2_ERR_ARG;
  if (probe) return (!ra.sparse || sparse3<CELL, H>()) ? 0 : DS2_ERR_ARG;
  const int ns = (ra.q.N + ra.q.gpd - 1) / ra.q.gpd;
  if (ra.sparse) {
    if (ra.nset == 2) return launch3_one<CELL, H, 2, 1, true>(bwd, ra, st);
    return launch3_one<CELL, H, 1, 1, true>(bwd, ra, st);
  }
  if (ra.nset == 2) return launch3_one<CELL, H, 2, 1>(bwd, ra, st);
  if (ns <= 8) return launch3_one<CELL, H, 1, 2>(bwd, ra, st);
  return launch3_one<CELL, H, 1, 1>(bwd, ra, st);
}

}  // namespace ds2r
