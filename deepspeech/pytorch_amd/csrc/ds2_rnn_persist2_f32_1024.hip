// Instantiations of the general persistent recurrent kernels (ds2_rnn_persist2_impl.h): storage float, hidden size 1024.
// One translation unit per (type, H) keeps hipcc's time per file bounded; the m-tile counts are the ones the BASELINE
// configurations need (cfg2: 1, cfg5b: 2, cfg5a: 4).
#include "ds2_rnn_persist2_impl.h"

namespace ds2q {
int launch_f32_1024(bool probe, bool bwd, int cell, int MT, const QArgs& a, hipStream_t st) {
#define DS2Q_CASE(CELL, M) if (cell == CELL && MT == M) return probe ? 0 : launch2<CELL, float, 1024, M>(bwd, a, st);
  DS2Q_CASE(CELL_GRU, 1) DS2Q_CASE(CELL_LSTM, 1) DS2Q_CASE(CELL_RNN, 1)   // tanh cells: round 6 (fp32 mode had them on one launch per time step)
#undef DS2Q_CASE
  return DS2_ERR_ARG;
}
}  // namespace ds2q
