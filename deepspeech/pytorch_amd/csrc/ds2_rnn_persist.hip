// Host side of the persistent recurrent sweeps: support query, scratch sizing, launch (kernels: ds2_rnn_persist_impl.h,
// instantiated per cell type in ds2_rnn_persist_{gru,lstm,rnn}.hip).
#include "ds2_rnn_persist_impl.h"

namespace ds2p {
int launch_gru(bool bwd, int H, const PArgs& a, hipStream_t st);
int launch_lstm(bool bwd, int H, const PArgs& a, hipStream_t st);
int launch_rnn(bool bwd, int H, const PArgs& a, hipStream_t st);
}  // namespace ds2p

namespace {
using namespace ds2p;

int cu_count() {   // of the CURRENT device (cached per device)
  static int n[DS2_MAX_DEVICES];
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DS2_MAX_DEVICES) return 0;
  if (n[dev] == 0 && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) n[dev] = v;
  return n[dev];
}

int dispatch(bool bwd, int cell, int H, const PArgs& a, hipStream_t st) {
  switch (cell) {
    case CELL_GRU: return launch_gru(bwd, H, a, st);
    case CELL_LSTM: return launch_lstm(bwd, H, a, st);
    case CELL_RNN: return launch_rnn(bwd, H, a, st);
  }
  return DS2_ERR_ARG;
}

// scratch head: [0,1024) cycle counters (-DDS2_PROBE builds only), [1024,3072) XCC-id handshake slots, [3072] the per-launch
// error word; the exchange buffer follows.  All of it is zeroed before every launch.
constexpr long AUX_BYTES = 4096;

long xbuf_bytes(int cell, int H, bool bwd) {
  const int G = cell == CELL_GRU ? 3 : cell == CELL_LSTM ? 4 : 1;
  const long X2 = (bwd ? (long)G * H : (long)H) / 2;
  return (long)NGROUPS * 2 * MAXS * X2 * 8;
}

}  // namespace

extern "C" {

// 1 if the persistent kernels cover this problem on the current device (bf16 storage only): H = 1024, the
// device has exactly 256 CUs (one workgroup per CU, all co-resident), and a direction's share of the minibatch fits
// the 16-row MFMA tile of each of its 8/D groups.
int ds2_rnn_persist_supported(int dtype, int cell, int D, int N, int H) {
  if (dtype != DS2_BF16 || (cell != CELL_GRU && cell != CELL_LSTM && cell != CELL_RNN)) return 0;
  if (H != 1024) return 0;
  if (D != 1 && D != 2) return 0;
  const int gpd = NGROUPS / D;
  if (N < 1 || (N + gpd - 1) / gpd > MAXS) return 0;
  return cu_count() == 256 ? 1 : 0;
}

// bytes of the exchange buffer (max of the forward and backward sweep needs) + 64 for the error word
long ds2_rnn_persist_ws_bytes(int cell, int H) { return AUX_BYTES + xbuf_bytes(cell, H, true); }

// Same contract as ds2_rnn_fwd (ds2_rnn.hip) for dtype bf16; ws = ds2_rnn_persist_ws_bytes() bytes (zeroed here).
// err: one device int, set to 1 if a workgroup gave up waiting (outputs are then NaN-poisoned).
int ds2_rnn_persist_fwd(int cell, int D, int N, int H, int Tp, const int* lens, const void* GI, const void* Whh,
                        const float* bhh, const float* h0, const float* c0, void* Hseq, long hseq_dstride, void* S, float* hn,
                        float* cn, void* ws, int* err, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(ds2_rnn_persist_supported(DS2_BF16, cell, D, N, H), DS2_ERR_ARG);
  DS2_REQUIRE(Tp > 0 && Tp < (int)TAG_INIT && ws && err, DS2_ERR_ARG);
  hipError_t e = hipMemsetAsync(ws, 0, AUX_BYTES + xbuf_bytes(cell, H, false), st);
  if (e != hipSuccess) return (int)e;
  PArgs a{};
  a.N = N; a.Tp = Tp; a.D = D; a.gpd = NGROUPS / D; a.lens = lens; a.W = (const bf16_t*)Whh; a.bhh = bhh;
  a.GI = (const bf16_t*)GI; a.Hseq = (bf16_t*)Hseq; a.hseq_dstride = hseq_dstride; a.S = (bf16_t*)S; a.h0 = h0; a.c0 = c0;
  a.hn = hn; a.cn = cn; a.xbuf = (u64*)((char*)ws + AUX_BYTES); a.err = err;
  a.xcc = (u64*)((char*)ws + 1024); a.lerr = (int*)((char*)ws + 3072);
#ifdef DS2_PROBE
  a.dbg = (unsigned long long*)ws;
  { const char* e_ = getenv("DS2_PERSIST_DBG"); a.dbgmask = e_ ? atoi(e_) : 0; }
#endif
  return dispatch(false, cell, H, a, st);
}

// Same contract as ds2_rnn_bwd for dtype bf16 (zero initial state).
int ds2_rnn_persist_bwd(int cell, int D, int N, int H, int Tp, const int* lens, const void* dOut, const void* WhhT,
                        const void* Hseq, long hseq_dstride, const void* S, void* dGI, void* dGH, void* ws, int* err,
                        ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(ds2_rnn_persist_supported(DS2_BF16, cell, D, N, H), DS2_ERR_ARG);
  DS2_REQUIRE(Tp > 0 && Tp < (int)TAG_INIT && ws && err, DS2_ERR_ARG);
  DS2_REQUIRE(cell != CELL_GRU || dGH != nullptr, DS2_ERR_ARG);
  hipError_t e = hipMemsetAsync(ws, 0, AUX_BYTES + xbuf_bytes(cell, H, true), st);
  if (e != hipSuccess) return (int)e;
  PArgs a{};
  a.N = N; a.Tp = Tp; a.D = D; a.gpd = NGROUPS / D; a.lens = lens; a.W = (const bf16_t*)WhhT;
  a.Hseq = (bf16_t*)Hseq; a.hseq_dstride = hseq_dstride; a.S = (bf16_t*)S; a.dOut = (const bf16_t*)dOut;
  a.dGI = (bf16_t*)dGI; a.dGH = (bf16_t*)dGH; a.xbuf = (u64*)((char*)ws + AUX_BYTES); a.err = err;
  a.xcc = (u64*)((char*)ws + 1024); a.lerr = (int*)((char*)ws + 3072);
#ifdef DS2_PROBE
  a.dbg = (unsigned long long*)ws;
  { const char* e_ = getenv("DS2_PERSIST_DBG"); a.dbgmask = e_ ? atoi(e_) : 0; }
#endif
  return dispatch(true, cell, H, a, st);
}

}  // extern "C"
