// Greedy (arg-max) CTC decoding on the device: what validation_step does through GreedyDecoder.decode (reference
// decoder.py:164-181 over process_string :146-162, called from model.py:256): per frame the arg-max class, then per sample the
// collapse of repeats and the removal of blanks.  The reference moves the full (N, T', C) probability tensor to the host and
// walks it with one .item() per frame; here only the surviving labels and their frame offsets travel.
//
// One wave per sample; frames in chunks of 64 (one per lane): arg-max over the C classes of the lane's frame (first
// maximum wins, like torch.max), keep = label != blank && (t == 0 || label != label of frame t-1) && t < size, compaction by
// ballot + popcount prefix.  HBM-bound: reads N*T'*C floats once (the rows are 116-256 bytes, each lane streams its own row).
#include "ds2_common.h"

namespace {

__global__ void __launch_bounds__(64) k_greedy_decode(const float* __restrict__ x, long stride_n, long stride_t, int T, int C,
                                                      const int* __restrict__ sizes, int blank, int* __restrict__ tokens,
                                                      int* __restrict__ offsets, int* __restrict__ counts) {
  const int n = blockIdx.x, lane = threadIdx.x;
  int size = sizes ? sizes[n] : T;
  size = size < 0 ? 0 : (size > T ? T : size);
  const float* xn = x + (long)n * stride_n;
  int* tok = tokens + (long)n * T;
  int* off = offsets + (long)n * T;
  int count = 0;
  int carry = -1;                                      // arg-max of the frame before this chunk (none before frame 0)
  for (int t0 = 0; t0 < size; t0 += 64) {
    const int t = t0 + lane;
    int best = -1;
    if (t < size) {
      const float* row = xn + (long)t * stride_t;
      float bv = row[0];
      best = 0;
      for (int c = 1; c < C; ++c) {
        const float v = row[c];
        if (v > bv) {
          bv = v;
          best = c;
        }
      }
    }
    int prev = __shfl_up(best, 1, 64);
    if (lane == 0) prev = carry;
    const bool keep = t < size && best != blank && (t == 0 || best != prev);
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
      tok[pos] = best;
      off[pos] = t;
    }
    count += __popcll(m);
    carry = __shfl(best, 63, 64);
  }
  if (lane == 0) counts[n] = count;
}

}  // namespace

extern "C" {

// x: scores (probabilities or logits) of sample n, frame t, class c at x[n*stride_n + t*stride_t + c] (f32; any (N,T',C)
// view whose class dimension is contiguous).  sizes: [N] valid frames per sample (device int32, may be null = T).
// Outputs (device int32): tokens [N][T], offsets [N][T] (first `counts[n]` entries of row n are valid), counts [N].
int ds2_greedy_decode(const float* x, long stride_n, long stride_t, int N, int T, int C, const int* sizes, int blank,
                      int* tokens, int* offsets, int* counts, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(N > 0 && T > 0 && C > 0 && blank >= 0 && blank < C, DS2_ERR_ARG);
  DS2_REQUIRE(x && tokens && offsets && counts, DS2_ERR_ARG);
  hipLaunchKernelGGL(k_greedy_decode, dim3(N), dim3(64), 0, st, x, stride_n, stride_t, T, C, sizes, blank, tokens, offsets, counts);
  DS2_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
