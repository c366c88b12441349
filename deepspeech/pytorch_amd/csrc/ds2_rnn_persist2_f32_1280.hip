// Instantiations of the general persistent recurrent kernels (ds2_rnn_persist2_impl.h): storage float, hidden size 1280 (round 6: the
// fp32 parity mode of config 5's width no longer falls to one launch per time step; the reference leaves hidden_size free,
// train_config.py:46-50).
#include "ds2_rnn_persist2_impl.h"

namespace ds2q {
int launch_f32_1280(bool probe, bool bwd, int cell, int MT, const QArgs& a, hipStream_t st) {
#define DS2Q_CASE(CELL, M) if (cell == CELL && MT == M) return probe ? 0 : launch2<CELL, float, 1280, M>(bwd, a, st);
  DS2Q_CASE(CELL_GRU, 1) DS2Q_CASE(CELL_LSTM, 1) DS2Q_CASE(CELL_RNN, 1)
#undef DS2Q_CASE
  return DS2_ERR_ARG;
}
}  // namespace ds2q
