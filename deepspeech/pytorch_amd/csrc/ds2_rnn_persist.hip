// Host side of the persistent recurrent sweeps: support query, scratch sizing, launch.
//   tuned kernels  (ds2_rnn_persist_impl.h,  instantiated per cell in ds2_rnn_persist_{gru,lstm,rnn}.hip): bf16, H = 1024,
//                   8 XCD-local groups of 32 workgroups, <= 16 samples per group -- BASELINE.json config 3;
//   general kernels, round 4 (ds2_rnn_persist3_impl.h, instantiated in ds2_rnn_persist3_*.hip): bf16, GRU / LSTM, H in {512, 768,
//                   800, 1024, 1280, 1536}: 32 units per workgroup, XCD-local groups for H <= 1024, up to 32 samples per group in
//                   two sample sets with their own step schedules -- config 5 and every bf16 width / batch the tuned kernels do not take;
//   general kernels, round 2 (ds2_rnn_persist2_impl.h, instantiated in ds2_rnn_persist2_*.hip): H in {800, 1024, 1280}, bf16 and fp32
//                   storage, GRU / LSTM, up to 64 samples per group -- config 2 (the fp32 parity mode) and bf16 groups of > 32 samples.
#include "ds2_rnn_persist3_impl.h"

namespace ds2p {
int launch_gru(bool bwd, int H, const PArgs& a, hipStream_t st, bool dense);
int launch_lstm(bool bwd, int H, const PArgs& a, hipStream_t st, bool dense);
int launch_rnn(bool bwd, int H, const PArgs& a, hipStream_t st, bool dense);
}  // namespace ds2p
namespace ds2q {
// 0 on success, DS2_ERR_ARG if the combination is not instantiated; `probe` only asks whether it is
int launch_bf16_800(bool probe, bool bwd, int cell, int MT, const QArgs& a, hipStream_t st);
int launch_bf16_1280(bool probe, bool bwd, int cell, int MT, const QArgs& a, hipStream_t st);
int launch_f32_800(bool probe, bool bwd, int cell, int MT, const QArgs& a, hipStream_t st);
int launch_f32_1024(bool probe, bool bwd, int cell, int MT, const QArgs& a, hipStream_t st);
int launch_f32_1280(bool probe, bool bwd, int cell, int MT, const QArgs& a, hipStream_t st);
}  // namespace ds2q

namespace ds2r {
int launch3_384(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
int launch3_640(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
int launch3_896(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
int launch3_1152(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
int launch3_1408(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
int launch3_512(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
int launch3_768(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
int launch3_800(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
int launch3_1024(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
int launch3_1280(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
int launch3_1536(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st);
}  // namespace ds2r

namespace {
using namespace ds2p;

int cu_count() {   // of the CURRENT device (cached per device)
  static int n[DS2_MAX_DEVICES];
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DS2_MAX_DEVICES) return 0;
  if (n[dev] == 0 && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) n[dev] = v;
  return n[dev];
}

// Routing A/B bits (ds2_persist_opts.variant / the `variant` argument of the queries; 0 = the shipping routing): bit 0 = do not use
// the round-4 general kernels; 1 = two-set groups execute every half-step; 2 = unused; 3 = the general kernels take H = 1024 too;
// 4 = the tuned kernels keep 9-16 clips per group; 5 = the 8-clip tuned kernels use dense products (round 4's form) instead of the
// structured-sparse ones; 6 = the general kernels keep dense 16-row tiles for groups of <= 8 clips too (round 5's form); 7 = groups
// of 9-16 clips run as two structured-sparse sets (measured and rejected, kept for A/B).  Rounds 2-5 kept these bits -- and the spin budget -- in process-wide variables behind setter entries;
// they travel with every call now, so the entries are re-entrant and a test that dies cannot re-route the launches after it.
int dispatch(bool bwd, int cell, int H, const PArgs& a, hipStream_t st, unsigned g_variant) {
  const bool dense = (g_variant & 32u) != 0;    // A/B: the 8-clip kernels without the structured-sparse products (round 4's form)
  switch (cell) {
    case CELL_GRU: return launch_gru(bwd, H, a, st, dense);
    case CELL_LSTM: return launch_lstm(bwd, H, a, st, dense);
    case CELL_RNN: return launch_rnn(bwd, H, a, st, dense);
  }
  return DS2_ERR_ARG;
}

// scratch head: [0,1024) cycle counters (-DDS2_PROBE builds only), [1024,3072) XCC-id handshake slots, [3072] the per-launch
// error word, [3076] the launch's spin budget, [3080] the arrival word of the kernels without a handshake; the exchange buffer follows.  All of it is reset to 0xFF bytes before every launch: the payload-only exchanges use
// the all-ones dword as "not published yet"; an all-ones tag never equals a step index, an all-ones handshake slot is not a
// signature, and the error word counts as raised only when it is 1.
constexpr long AUX_BYTES = 4096;

// [3076] of the scratch head: the launch's spin budget (all-ones after the reset = none beyond SPIN_LIMIT)
int set_spin_budget(void* ws, const ds2_persist_opts* o, hipStream_t st) {
  if (!o || o->spin_limit == 0) return 0;
  return (int)hipMemsetD32Async((hipDeviceptr_t)((char*)ws + 3076), (int)o->spin_limit, 1, st);
}
// How long the workgroups of a launch wait for ALL of them to become resident (csrc/ds2_rnn_persist_impl.h, raise_err_startup).
constexpr unsigned STARTUP_MS_DEFAULT = 300;
unsigned startup_ms(const ds2_persist_opts* o) { return o && o->startup_ms ? o->startup_ms : STARTUP_MS_DEFAULT; }
unsigned variant_of(const ds2_persist_opts* o) { return o ? o->variant : 0u; }

int gates(int cell) { return cell == CELL_GRU ? 3 : cell == CELL_LSTM ? 4 : 1; }

long xbuf_bytes(int cell, int H, bool bwd) {
  const long X2 = (bwd ? (long)gates(cell) * H : (long)H) / 2;
  return (long)NGROUPS * 2 * MAXS * X2 * 8;
}


bool tuned_ok(int dtype, int cell, int D, int N, int H, unsigned g_variant) {
  if (dtype != DS2_BF16 || H != 1024 || (D != 1 && D != 2)) return false;
  const int gpd = NGROUPS / D;
  if (g_variant & 8u) return false;                     // A/B: the general kernels take H = 1024 too
  const int ns = N >= 1 ? (N + gpd - 1) / gpd : MAXS + 1;
  // 9-16 clips per group: the round-4 general kernels are faster (GRU bi, 64 clips: 2.7 vs 3.4 us per forward step,
  // profiles/r04n_time_sweeps.txt) unless A/B bit 4 asks for the tuned ones; RNN cells only exist here
  const int cap = ((g_variant & 16u) || cell == CELL_RNN) ? MAXS : 8;
  return ns <= cap && cu_count() >= NGROUPS * 32;       // one workgroup per CU, all 256 co-resident
}

// Geometry of the general kernels: P = H/16 workgroups per group, as many groups per direction as the chip holds.
struct Plan2 {
  int gpd, NG, MT, P;
};
int launch2_any(bool probe, bool bwd, int dtype, int cell, int H, int MT, const ds2q::QArgs& a, hipStream_t st) {
  if (dtype == DS2_BF16 && H == 800) return ds2q::launch_bf16_800(probe, bwd, cell, MT, a, st);
  if (dtype == DS2_BF16 && H == 1280) return ds2q::launch_bf16_1280(probe, bwd, cell, MT, a, st);
  if (dtype == DS2_F32 && H == 800) return ds2q::launch_f32_800(probe, bwd, cell, MT, a, st);
  if (dtype == DS2_F32 && H == 1024) return ds2q::launch_f32_1024(probe, bwd, cell, MT, a, st);
  if (dtype == DS2_F32 && H == 1280) return ds2q::launch_f32_1280(probe, bwd, cell, MT, a, st);
  return DS2_ERR_ARG;
}
bool plan2(int dtype, int cell, int D, int N, int H, Plan2& pl) {
  if ((dtype != DS2_BF16 && dtype != DS2_F32) || (D != 1 && D != 2) || N < 1 || H % 16 != 0) return false;
  const int cus = cu_count();
  pl.P = H / 16;
  if (cus < 256 || pl.P * D > cus) return false;
  pl.gpd = cus / pl.P / D;
  if (pl.gpd > N) pl.gpd = N;
  pl.NG = pl.gpd * D;
  const int ns = (N + pl.gpd - 1) / pl.gpd;
  pl.MT = ns <= 16 ? 1 : ns <= 32 ? 2 : ns <= 64 ? 4 : 0;
  if (pl.MT == 0) return false;
  ds2q::QArgs dummy{};
  return launch2_any(true, false, dtype, cell, H, pl.MT, dummy, nullptr) == 0;
}
// ---- round-4 general kernels (bf16, 32 units per workgroup) -----------------------------------------------------------------
struct Plan3H {
  int gpd, NG, P, xmap, gx, nset, sparse;
};
int launch3_any(bool probe, bool bwd, int cell, int H, const ds2r::RArgs& a, hipStream_t st) {
  switch (H) {
    case 384: return ds2r::launch3_384(probe, bwd, cell, a, st);
    case 640: return ds2r::launch3_640(probe, bwd, cell, a, st);
    case 896: return ds2r::launch3_896(probe, bwd, cell, a, st);
    case 1152: return ds2r::launch3_1152(probe, bwd, cell, a, st);
    case 1408: return ds2r::launch3_1408(probe, bwd, cell, a, st);
    case 512: return ds2r::launch3_512(probe, bwd, cell, a, st);
    case 768: return ds2r::launch3_768(probe, bwd, cell, a, st);
    case 800: return ds2r::launch3_800(probe, bwd, cell, a, st);
    case 1024: return ds2r::launch3_1024(probe, bwd, cell, a, st);
    case 1280: return ds2r::launch3_1280(probe, bwd, cell, a, st);
    case 1536: return ds2r::launch3_1536(probe, bwd, cell, a, st);
  }
  return DS2_ERR_ARG;
}
// cus = compute units to plan for (the device's, or 256 for the "would a full device take this shape" question)
bool plan3h(int dtype, int cell, int D, int N, int H, int cus, Plan3H& pl, unsigned g_variant) {
  if ((g_variant & 1u) || dtype != DS2_BF16 || (D != 1 && D != 2) || N < 1 || H % 32 != 0) return false;
  if (cell != CELL_GRU && cell != CELL_LSTM) return false;
  if (cus < 256) return false;
  pl.P = H / 32;
  int slots;
  if (pl.P <= 32) {            // a group fits one XCD: 8 x floor(32 / P) group slots, block b on XCD b % 8
    pl.xmap = 1;
    pl.gx = 32 / pl.P;
    slots = 8 * pl.gx;
  } else {
    pl.xmap = 0;
    pl.gx = 0;
    slots = 256 / pl.P;
  }
  pl.gpd = slots / D;
  if (pl.gpd < 1) return false;
  if (pl.gpd > N) pl.gpd = N;
  pl.NG = pl.gpd * D;
  const int ns = (N + pl.gpd - 1) / pl.gpd;
  ds2r::RArgs dummy{};
  // groups of <= 8 clips: ONE set on the structured-sparse products where the width has them (H % 256 == 0; variant bit 6: never).
  // 9-16 clips as two sparse sets of <= 8 (variant bit 7, A/B only) LOSE to one dense 16-row set: a half-step is a latency chain
  // whatever its matrix work -- config 5b's groups of 11: 4.07 / 4.54 us per time step against 2.71 / 3.38 (profiles/r06c_time_sweeps.txt)
  dummy.sparse = 1;
  const int sparse_max = (g_variant & 128u) ? 16 : 8;
  // measured (profiles/r06e_sparse_single_set.txt): LSTM-1280, 8 clips per group 2.62 / 2.88 against 2.70 / 3.01 us per time step
  // dense; GRU-768 equal; LSTM-512 forward 1.93 against 1.69 (two k-blocks per wave leave nothing to overlap) -> from H = 1024 on
  pl.sparse = (!(g_variant & 64u) && ns <= sparse_max && (H >= 1024 || (g_variant & 128u)) && launch3_any(true, false, cell, H, dummy, nullptr) == 0) ? 1 : 0;
  pl.nset = pl.sparse ? (ns <= 8 ? 1 : 2) : (ns <= 16 ? 1 : ns <= 32 ? 2 : 0);
  if (pl.nset == 0) return false;
  dummy.sparse = 0;
  return launch3_any(true, false, cell, H, dummy, nullptr) == 0;
}
long xbuf3_bytes(int cell, int H, const Plan3H& pl, bool bwd) {
  const long kt = (bwd ? (long)gates(cell) * H : (long)H) / 32;
  return (long)pl.NG * pl.nset * 4 * kt * 1024;       // per group: nset sets x four payload slots of [k-step][lq][16 rows] x 16 B
}

// XCC-id handshake words of the round-4 general kernels: 32 per group, BEHIND the exchange buffer (up to 32 groups -- 8 x gx -- do
// not fit the 2 KB the tuned kernels' eight groups use inside the head: with 16 groups (H = 512) the words of groups 8-15 used to
// land on the spin budget and on the first group's exchange slots -- an intermittent stale read at the second time step)
long xcc3_bytes(const Plan3H& pl) { return (long)pl.NG * 32 * 8; }

// Two-set groups execute a set's half-steps only while one of its clips is inside its sequence (ds2r::sched3): the padding rows of
// the sweep's outputs are zeroed here instead of by the half-steps left out.  One workgroup per 4 rows of a [T' x N] matrix.
__global__ void __launch_bounds__(256) k_zero_pad3(unsigned char* X, long ld_bytes, int row_bytes, const int* __restrict__ lens, int N, long R) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  if ((int)(r / N) < lens[r % N]) return;
  unsigned char* p = X + r * ld_bytes;
  for (int o = (threadIdx.x & 63) * 16; o < row_bytes; o += 64 * 16) *reinterpret_cast<uint4*>(p + o) = make_uint4(0, 0, 0, 0);
}
void zero_pad3(void* X, long ld_bytes, long row_bytes, const int* lens, int N, int Tp, hipStream_t st) {
  const long R = (long)Tp * N;
  hipLaunchKernelGGL(k_zero_pad3, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, (unsigned char*)X, ld_bytes, (int)row_bytes, lens, N, R);
}

long xbuf2_bytes(int dtype, int cell, int H, const Plan2& pl, bool bwd) {
  const int ksz = dtype == DS2_BF16 ? 32 : 16;
  const long kt = (bwd ? (long)gates(cell) * H : (long)H) / ksz;
  // two parities of tagged granules (2048 B per k-step and m-tile) or four payload-only slots (1024 B): the same bytes
  return (long)pl.NG * 2 * kt * pl.MT * 2048;
}

}  // namespace

extern "C" {

// 1 if a persistent kernel covers this problem on the current device (>= 256 CUs: one workgroup per CU, all co-resident).
int ds2_rnn_persist_supported(int dtype, int cell, int D, int N, int H, unsigned variant) {
  if (cell != CELL_GRU && cell != CELL_LSTM && cell != CELL_RNN) return 0;
  if (tuned_ok(dtype, cell, D, N, H, variant)) return 1;
  Plan3H p3;
  if (plan3h(dtype, cell, D, N, H, cu_count(), p3, variant)) return 1;
  Plan2 pl;
  return plan2(dtype, cell, D, N, H, pl) ? 1 : 0;
}

// Which kernel family ds2_rnn_persist_fwd / _bwd run for the problem on the current device: 0 none, 1 tuned (H = 1024, <= 8 samples
// per group: k_rnn_persist_fwd4 / bwd4), 2 tuned (9-16 samples: k_rnn_persist_fwd / bwd), 3 round-4 general (k_rnn_persist3_*),
// 4 round-2 general (k_rnn_persist2_*).  For measurement tools (bench.py names the rocprofv3 kernel from it).
int ds2_rnn_persist_kind(int dtype, int cell, int D, int N, int H, unsigned variant) {
  if (cell != CELL_GRU && cell != CELL_LSTM && cell != CELL_RNN) return 0;
  if (tuned_ok(dtype, cell, D, N, H, variant)) return (N + NGROUPS / D - 1) / (NGROUPS / D) <= 8 ? 1 : 2;
  Plan3H p3;
  if (plan3h(dtype, cell, D, N, H, cu_count(), p3, variant)) return 3;
  Plan2 pl;
  return plan2(dtype, cell, D, N, H, pl) ? 4 : 0;
}

// 1 if a persistent kernel is INSTANTIATED for this problem, whatever the current device's CU count (what a full 256-CU device
// would run): lets the caller tell "this device is too small" (an error) from "no persistent kernel for this shape" (a warning).
int ds2_rnn_persist_shape_covered(int dtype, int cell, int D, int N, int H) {
  if (cell != CELL_GRU && cell != CELL_LSTM && cell != CELL_RNN) return 0;
  if (dtype == DS2_BF16 && H == 1024 && (D == 1 || D == 2) && N >= 1 && (N + NGROUPS / D - 1) / (NGROUPS / D) <= MAXS) return 1;
  Plan3H p3;
  if (plan3h(dtype, cell, D, N, H, 256, p3, 0u)) return 1;
  if ((dtype != DS2_BF16 && dtype != DS2_F32) || (D != 1 && D != 2) || N < 1 || H % 16 != 0) return 0;
  const int P = H / 16;
  if (P * D > 256) return 0;
  int gpd = 256 / P / D;
  if (gpd > N) gpd = N;
  const int ns = (N + gpd - 1) / gpd;
  const int MT = ns <= 16 ? 1 : ns <= 32 ? 2 : ns <= 64 ? 4 : 0;
  if (MT == 0) return 0;
  ds2q::QArgs dummy{};
  return launch2_any(true, false, dtype, cell, H, MT, dummy, nullptr) == 0 ? 1 : 0;
}

// scratch bytes of one sweep (exchange buffer for the larger of the forward / BPTT needs + the head described above)
long ds2_rnn_persist_ws_bytes(int dtype, int cell, int D, int N, int H, unsigned variant) {
#ifdef DS2_PROBE
  if (tuned_ok(dtype, cell, D, N, H, variant)) return AUX_BYTES + xbuf_bytes(cell, H, true) + 32 * ds2p::TL_N * ds2p::TL_K * 8;   // + timeline
#endif
  if (tuned_ok(dtype, cell, D, N, H, variant)) return AUX_BYTES + xbuf_bytes(cell, H, true);
  Plan3H p3;
  if (plan3h(dtype, cell, D, N, H, cu_count(), p3, variant)) return AUX_BYTES + xbuf3_bytes(cell, H, p3, true) + xcc3_bytes(p3);
  Plan2 pl;
  if (!plan2(dtype, cell, D, N, H, pl)) return 0;
  return AUX_BYTES + xbuf2_bytes(dtype, cell, H, pl, true);
}

// Same contract as ds2_rnn_fwd (ds2_rnn.hip); ws = ds2_rnn_persist_ws_bytes() bytes (reset here).
// err: one device int, set to 1 if a workgroup gave up waiting (outputs are then NaN-poisoned).
int ds2_rnn_persist_fwd(int dtype, int cell, int D, int N, int H, int Tp, const int* lens, const void* GI, const void* Whh,
                        const float* bhh, const float* h0, const float* c0, void* Hseq, long hseq_dstride, void* S, float* hn,
                        float* cn, void* ws, int* err, const ds2_persist_opts* opts, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  const unsigned g_variant = variant_of(opts);
  DS2_REQUIRE(Tp > 0 && Tp < (int)TAG_INIT && ws && err, DS2_ERR_ARG);
  if (tuned_ok(dtype, cell, D, N, H, g_variant)) {
    hipError_t e = hipMemsetAsync(ws, 0xff, AUX_BYTES + xbuf_bytes(cell, H, false), st);   // 0xFF: see gather_mma_tf
    if (e != hipSuccess) return (int)e;
    if (int r = set_spin_budget(ws, opts, st)) return r;
    PArgs a{};
    a.N = N; a.Tp = Tp; a.D = D; a.gpd = NGROUPS / D; a.lens = lens; a.W = (const bf16_t*)Whh; a.bhh = bhh;
    a.GI = (const bf16_t*)GI; a.Hseq = (bf16_t*)Hseq; a.hseq_dstride = hseq_dstride; a.S = (bf16_t*)S; a.h0 = h0; a.c0 = c0;
    a.hn = hn; a.cn = cn; a.xbuf = (u64*)((char*)ws + AUX_BYTES); a.err = err;
    a.xcc = (u64*)((char*)ws + 1024); a.lerr = (int*)((char*)ws + 3072); a.startup_ms = startup_ms(opts);
#ifdef DS2_PROBE
    a.dbg = (unsigned long long*)ws;
    a.tl = (unsigned long long*)((char*)ws + AUX_BYTES + xbuf_bytes(cell, H, true));
    { const char* e_ = getenv("DS2_PERSIST_DBG"); a.dbgmask = e_ ? atoi(e_) : 0; }
#endif
    return dispatch(false, cell, H, a, st, g_variant);
  }
  Plan3H p3;
  if (plan3h(dtype, cell, D, N, H, cu_count(), p3, g_variant)) {
    const long xb = xbuf3_bytes(cell, H, p3, false);
    hipError_t e = hipMemsetAsync(ws, 0xff, AUX_BYTES + xb + xcc3_bytes(p3), st);
    if (e != hipSuccess) return (int)e;
    if (int r = set_spin_budget(ws, opts, st)) return r;
    ds2r::RArgs ra{};
    ds2q::QArgs& a = ra.q;
    a.N = N; a.Tp = Tp; a.D = D; a.gpd = p3.gpd; a.NG = p3.NG; a.lens = lens; a.W = Whh; a.bhh = bhh; a.GI = GI; a.Hseq = Hseq;
    a.hseq_dstride = hseq_dstride; a.S = S; a.h0 = h0; a.c0 = c0; a.hn = hn; a.cn = cn;
    a.xbuf = (char*)ws + AUX_BYTES; a.xgroup_bytes = xb / p3.NG; a.err = err; a.lerr = (int*)((char*)ws + 3072); a.startup_ms = startup_ms(opts);
    ra.xcc = (u64*)((char*)ws + AUX_BYTES + xb); ra.P = p3.P; ra.xmap = p3.xmap; ra.gx = p3.gx; ra.nset = p3.nset; ra.sparse = p3.sparse;
    ra.skip = (p3.nset == 2 && !(g_variant & 2u)) ? 1 : 0;
    if (ra.skip & 1)
      for (int d = 0; d < D; ++d)       // h_t of the padding frames (Hseq points at t = 0)
        zero_pad3((char*)Hseq + (long)d * hseq_dstride * 2, (long)H * 2, (long)H * 2, lens, N, Tp, st);
#ifdef DS2_PROBE
    ra.dbg = (unsigned long long*)ws;
    { const char* e_ = getenv("DS2_PERSIST_DBG"); ra.dbgmask = e_ ? atoi(e_) : 0; }
#endif
    return launch3_any(false, false, cell, H, ra, st);
  }
  Plan2 pl;
  DS2_REQUIRE(plan2(dtype, cell, D, N, H, pl), DS2_ERR_ARG);
  const long xb = xbuf2_bytes(dtype, cell, H, pl, false);
  hipError_t e = hipMemsetAsync(ws, 0xff, AUX_BYTES + xb, st);
  if (e != hipSuccess) return (int)e;
  if (int r = set_spin_budget(ws, opts, st)) return r;
  ds2q::QArgs a{};
  a.N = N; a.Tp = Tp; a.D = D; a.gpd = pl.gpd; a.NG = pl.NG; a.lens = lens; a.W = Whh; a.bhh = bhh; a.GI = GI; a.Hseq = Hseq;
  a.hseq_dstride = hseq_dstride; a.S = S; a.h0 = h0; a.c0 = c0; a.hn = hn; a.cn = cn;
  a.xbuf = (char*)ws + AUX_BYTES; a.xgroup_bytes = xb / pl.NG; a.err = err; a.lerr = (int*)((char*)ws + 3072); a.startup_ms = startup_ms(opts);
  return launch2_any(false, false, dtype, cell, H, pl.MT, a, st);
}

// BPTT sweep.  Inputs as ds2_rnn_bwd (zero initial state).  Outputs: dGI [Tp*N][D*G*H]; for GRU dQ [D][Tp][N][H] = dn * r, the
// only slot of the hidden-side gate gradient that differs from dGI's (d(W_hh h + b_hh) = [dr, dz, dQ]); dBacc [D][N][NB*H] f32
// (may be null) = per-sample sums over time of the gate-gradient planes as stored (NB = 4 for GRU: dr, dz, dn, dQ; G otherwise):
// bias_ih.grad = sum over samples of planes 0..G-1, bias_hh.grad (GRU) = planes 0, 1, 3.
int ds2_rnn_persist_bwd(int dtype, int cell, int D, int N, int H, int Tp, const int* lens, const void* dOut, const void* WhhT,
                        const void* Hseq, long hseq_dstride, const void* S, void* dGI, void* dQ, float* dBacc, int flags, void* ws,
                        int* err, const ds2_persist_opts* opts, ds2_stream_t st_) {
  void* dGH = dQ;
  hipStream_t st = (hipStream_t)st_;
  const unsigned g_variant = variant_of(opts);
  DS2_REQUIRE(Tp > 0 && Tp < (int)TAG_INIT && ws && err, DS2_ERR_ARG);
  DS2_REQUIRE(cell != CELL_GRU || dGH != nullptr, DS2_ERR_ARG);
  if (tuned_ok(dtype, cell, D, N, H, g_variant)) {
    hipError_t e = hipMemsetAsync(ws, 0xff, AUX_BYTES + xbuf_bytes(cell, H, true), st);
    if (e != hipSuccess) return (int)e;
    if (int r = set_spin_budget(ws, opts, st)) return r;
    PArgs a{};
    a.N = N; a.Tp = Tp; a.D = D; a.gpd = NGROUPS / D; a.lens = lens; a.W = (const bf16_t*)WhhT;
    a.Hseq = (bf16_t*)Hseq; a.hseq_dstride = hseq_dstride; a.S = (bf16_t*)S; a.dOut = (const bf16_t*)dOut;
    a.dGI = (bf16_t*)dGI; a.dGH = (bf16_t*)dGH; a.dBacc = dBacc; a.xbuf = (u64*)((char*)ws + AUX_BYTES); a.err = err;
    a.xcc = (u64*)((char*)ws + 1024); a.lerr = (int*)((char*)ws + 3072); a.startup_ms = startup_ms(opts);
#ifdef DS2_PROBE
    a.dbg = (unsigned long long*)ws;
    a.tl = (unsigned long long*)((char*)ws + AUX_BYTES + xbuf_bytes(cell, H, true));
    { const char* e_ = getenv("DS2_PERSIST_DBG"); a.dbgmask = e_ ? atoi(e_) : 0; }
#endif
    return dispatch(true, cell, H, a, st, g_variant);
  }
  Plan3H p3;
  if (plan3h(dtype, cell, D, N, H, cu_count(), p3, g_variant)) {
    const long xb = xbuf3_bytes(cell, H, p3, true);
    hipError_t e = hipMemsetAsync(ws, 0xff, AUX_BYTES + xb + xcc3_bytes(p3), st);
    if (e != hipSuccess) return (int)e;
    if (int r = set_spin_budget(ws, opts, st)) return r;
    ds2r::RArgs ra{};
    ds2q::QArgs& a = ra.q;
    a.N = N; a.Tp = Tp; a.D = D; a.gpd = p3.gpd; a.NG = p3.NG; a.lens = lens; a.W = WhhT; a.Hseq = (void*)Hseq;
    a.hseq_dstride = hseq_dstride; a.S = (void*)S; a.dOut = dOut; a.dGI = dGI; a.dGH = dGH; a.dBacc = dBacc;
    a.xbuf = (char*)ws + AUX_BYTES; a.xgroup_bytes = xb / p3.NG; a.err = err; a.lerr = (int*)((char*)ws + 3072); a.startup_ms = startup_ms(opts);
    ra.xcc = (u64*)((char*)ws + AUX_BYTES + xb); ra.P = p3.P; ra.xmap = p3.xmap; ra.gx = p3.gx; ra.nset = p3.nset; ra.sparse = p3.sparse;
    ra.skip = (p3.nset == 2 && !(g_variant & 2u)) ? 1 : 0;
    if ((ra.skip & 1) && !(flags & 1)) {     // flags bit 0: nobody reads the padding rows (row-list consumers)
      const long GHb = (long)gates(cell) * H * 2;
      zero_pad3(dGI, D * GHb, D * GHb, lens, N, Tp, st);
      if (dGH)
        for (int d = 0; d < D; ++d) zero_pad3((char*)dGH + (long)d * Tp * N * H * 2, (long)H * 2, (long)H * 2, lens, N, Tp, st);
    }
#ifdef DS2_PROBE
    ra.dbg = (unsigned long long*)ws;
    { const char* e_ = getenv("DS2_PERSIST_DBG"); ra.dbgmask = e_ ? atoi(e_) : 0; }
#endif
    return launch3_any(false, true, cell, H, ra, st);
  }
  Plan2 pl;
  DS2_REQUIRE(plan2(dtype, cell, D, N, H, pl), DS2_ERR_ARG);
  const long xb = xbuf2_bytes(dtype, cell, H, pl, true);
  hipError_t e = hipMemsetAsync(ws, 0xff, AUX_BYTES + xb, st);
  if (e != hipSuccess) return (int)e;
  if (int r = set_spin_budget(ws, opts, st)) return r;
  ds2q::QArgs a{};
  a.N = N; a.Tp = Tp; a.D = D; a.gpd = pl.gpd; a.NG = pl.NG; a.lens = lens; a.W = WhhT; a.Hseq = (void*)Hseq;
  a.hseq_dstride = hseq_dstride; a.S = (void*)S; a.dOut = dOut; a.dGI = dGI; a.dGH = dGH; a.dBacc = dBacc;
  a.xbuf = (char*)ws + AUX_BYTES; a.xgroup_bytes = xb / pl.NG; a.err = err; a.lerr = (int*)((char*)ws + 3072); a.startup_ms = startup_ms(opts);
  return launch2_any(false, true, dtype, cell, H, pl.MT, a, st);
}

}  // extern "C"
