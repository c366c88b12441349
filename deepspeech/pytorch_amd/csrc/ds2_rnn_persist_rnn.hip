// Instantiations of the persistent recurrent kernels (ds2_rnn_persist_impl.h) for the RNN cell -- one translation unit per
// cell keeps hipcc's time and memory per file bounded (the kernels are fully unrolled around register-resident weights).
#include "ds2_rnn_persist_impl.h"

namespace ds2p {
int launch_rnn(bool bwd, int H, const PArgs& a, hipStream_t st, bool dense) {
  if (H == 1024) return launch<CELL_RNN, 1024, 32>(bwd, a, st, dense);
  return DS2_ERR_ARG;
}
}  // namespace ds2p
