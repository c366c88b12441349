// Instantiations of the round-4 general persistent recurrent kernels (ds2_rnn_persist3_impl.h: bf16 storage, 32 hidden units per
// workgroup): hidden size 640, GRU and LSTM.  One translation unit per width keeps hipcc's time per file bounded.  (Round 5: the
// widths between the round-4 ones, so that every bf16 hidden size up to 1536 reaches a persistent kernel after at most 128 units of
// zero padding -- model.DeepSpeech._padded_hidden.)
#include "ds2_rnn_persist3_impl.h"

namespace ds2r {
int launch3_640(bool probe, bool bwd, int cell, const RArgs& a, hipStream_t st) {
  if (cell == CELL_GRU) return launch3<CELL_GRU, 640>(probe, bwd, a, st);
  if (cell == CELL_LSTM) return launch3<CELL_LSTM, 640>(probe, bwd, a, st);
  return DS2_ERR_ARG;
}
}  // namespace ds2r
