// Log-spectrogram front-end on the device: what SpectrogramParser.compute_spectrogram does per
// utterance on the CPU loader workers (reference loader/data_loader.py:73-94: librosa.stft(n_fft = win_length = 320, hop 160,
// hamming window, center = True) -> magnitude -> log1p -> (x - mean) / std over the whole utterance) and what _collate_fn does
// with the results (:247-270: zero-padded (N, 1, 161, Tmax) batch), producing the model's input tensor directly.
//
// The 320-point real DFT of every frame is one fp32 MFMA GEMM: the frames of an utterance are the rows of a matrix with row
// stride = hop (overlapping rows of the centre-padded waveform, no frame copy), the basis [2*161][320] carries the window.
//   k_spect_pad   : waveform -> centre-padded copy (zeros = librosa >= 0.10 default, or reflect = older default)
//   ds2_gemm_nt   : C[t][0:161] = Re, C[t][161:322] = Im      (v_mfma_f32_32x32x2_f32, exact fp32 products)
//   k_spect_stats : per utterance sum / sum of squares of log1p(|X|) over its own frames (fp64 partials, fixed order)
//   k_spect_write : normalise, transpose to [f][t] through LDS, zero the padding frames
// Roofline: MFMA fp32 (2*320*322 flop per frame) against ~1.3 KB of HBM traffic per frame: compute-bound on the fp32 matrix
// pipe (157 TFLOP/s), ~0.1 ms for a 32 x 15 s batch; the CPU path spends tens of ms per clip.
#include "ds2_common.h"

extern "C" int ds2_gemm_nt(int dtype, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda,
                           long ldb, long ldc, int out_f32, int batch, long strideA, long strideB, long strideC, long strideBias,
                           int splitk, ds2_stream_t st);

namespace {

constexpr int NFFT = 320, HOP = 160, NBIN = 161, LDC = 336;   // LDC: row stride of the GEMM output (322 rounded up to 16)
constexpr int STAT_BLOCKS = 32;

__global__ void __launch_bounds__(256) k_spect_pad(const float* __restrict__ wav, long ldw, const int* __restrict__ nsamp, int reflect,
                                                   float* __restrict__ ypad, long lpad) {
  const int n = blockIdx.y;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= lpad) return;
  const int L = nsamp[n];
  long s = i - NFFT / 2;
  float v = 0.f;
  if (i < (long)L + NFFT) {              // inside this utterance's padded extent
    if (reflect) {
      if (s < 0) s = -s;
      if (s >= L) s = 2 * ((long)L - 1) - s;
    }
    if (s >= 0 && s < L) v = wav[(long)n * ldw + s];
  }
  ypad[(long)n * lpad + i] = v;
}

__device__ __forceinline__ float logmag(const float* row, int f) {
  const float re = row[f], im = row[NBIN + f];
  return log1pf(sqrtf(re * re + im * im));
}

// grid (STAT_BLOCKS, N): block b of sample n sums its share of the frames
__global__ void __launch_bounds__(256) k_spect_stats(const float* __restrict__ C, long strideC, const int* __restrict__ nsamp,
                                                     double* __restrict__ partial) {
  __shared__ double red[2][4];
  const int n = blockIdx.y;
  const int T = 1 + nsamp[n] / HOP;
  const float* Cn = C + (long)n * strideC;
  double s = 0.0, q = 0.0;
  for (int t = blockIdx.x; t < T; t += STAT_BLOCKS) {
    const float* row = Cn + (long)t * LDC;
    if (threadIdx.x < NBIN) {
      const double v = (double)logmag(row, threadIdx.x);
      s += v;
      q += v * v;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[((long)n * STAT_BLOCKS + blockIdx.x) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[((long)n * STAT_BLOCKS + blockIdx.x) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// mean / 1/std per sample: torch's spect.mean() and spect.std() (unbiased, data_loader.py:88-92)
__global__ void k_spect_finalize(const double* __restrict__ partial, const int* __restrict__ nsamp, int N, float* __restrict__ ms) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < STAT_BLOCKS; ++b) {
    s += partial[((long)n * STAT_BLOCKS + b) * 2];
    q += partial[((long)n * STAT_BLOCKS + b) * 2 + 1];
  }
  const double cnt = (double)NBIN * (1 + nsamp[n] / HOP);
  const double mean = s / cnt;
  double var = (q - cnt * mean * mean) / (cnt - 1.0);
  if (var < 0.0) var = 0.0;
  ms[2 * n] = (float)mean;
  ms[2 * n + 1] = (float)(1.0 / sqrt(var));
}

// grid (ceil(Tmax/64), 3, N): 64 frames x 64 bins per block through LDS; out[n][0][f][t]
__global__ void __launch_bounds__(256) k_spect_write(const float* __restrict__ C, long strideC, const int* __restrict__ nsamp,
                                                     const float* __restrict__ ms, int normalize, float* __restrict__ out, int Tmax) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, t0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
  const int T = 1 + nsamp[n] / HOP;
  const float mean = normalize ? ms[2 * n] : 0.f, rstd = normalize ? ms[2 * n + 1] : 1.f;
  const float* Cn = C + (long)n * strideC;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int t = t0 + ty * 16 + i, f = f0 + tx;
    float v = 0.f;
    if (t < T && f < NBIN) v = (logmag(Cn + (long)t * LDC, f) - mean) * rstd;
    tile[ty * 16 + i][tx] = v;
  }
  __syncthreads();
  float* on = out + (long)n * NBIN * Tmax;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int f = f0 + ty * 16 + i, t = t0 + tx;
    if (f < NBIN && t < Tmax) on[(long)f * Tmax + t] = tile[tx][ty * 16 + i];
  }
}

inline long pad_len(int Lmax) { return (((long)Lmax + NFFT + 3) / 4) * 4 + NFFT; }

}  // namespace

extern "C" {

// frames of a waveform of `nsamples` samples (librosa.stft, center = True): 1 + nsamples / hop
int ds2_spect_frames(int nsamples) { return 1 + nsamples / HOP; }
// bytes of scratch: padded waveforms + GEMM output + statistics
long ds2_spect_ws_bytes(int N, int Lmax) {
  const long Tmax = 1 + Lmax / HOP;
  return ((long)N * pad_len(Lmax) + (long)N * Tmax * LDC + 2L * N + 16) * 4 + (long)N * STAT_BLOCKS * 2 * 8;
}

// wav [N][ldw] f32 (utterance n = first nsamples[n] entries of row n), nsamples [N] device int32, 16 kHz / 20 ms / 10 ms
// geometry (n_fft 320, hop 160: the geometry the conv kernels are specialised for).  basis [322][320] f32: rows 0..160
// window[k] * cos(2 pi f k / 320), rows 161..321 -window[k] * sin(...) (built by the binding, any window).  reflect: 0 =
// zero centre padding, 1 = reflect.  normalize: (x - mean) / std per utterance (unbiased std, as torch .std()).
// out (N, 1, 161, Tmax) f32 with Tmax = 1 + Lmax/160, frames >= the utterance's own count are zero (_collate_fn layout).
int ds2_spectrogram(const float* wav, long ldw, const int* nsamples, int N, int Lmax, const float* basis, int reflect,
                    int normalize, float* out, void* ws, ds2_stream_t st_) {
  hipStream_t st = (hipStream_t)st_;
  DS2_REQUIRE(wav && nsamples && basis && out && ws && N > 0 && Lmax > 0 && ldw >= Lmax, DS2_ERR_ARG);
  const long lpad = pad_len(Lmax);
  const int Tmax = 1 + Lmax / HOP;
  float* ypad = (float*)ws;
  float* Cbuf = ypad + (long)N * lpad;
  float* ms = Cbuf + (long)N * Tmax * LDC;
  double* partial = (double*)(((uintptr_t)(ms + 2L * N) + 15) & ~(uintptr_t)15);
  hipLaunchKernelGGL(k_spect_pad, dim3(ds2_cdiv(lpad, 256), N), dim3(256), 0, st, wav, ldw, nsamples, reflect, ypad, lpad);
  DS2_CHECK_LAUNCH();
  int rc = ds2_gemm_nt(DS2_F32, ypad, basis, Cbuf, nullptr, Tmax, 2 * NBIN, NFFT, HOP, NFFT, LDC, 1, N, lpad, 0, (long)Tmax * LDC, 0, 1, st_);
  if (rc != 0) return rc;
  if (normalize) {
    hipLaunchKernelGGL(k_spect_stats, dim3(STAT_BLOCKS, N), dim3(256), 0, st, (const float*)Cbuf, (long)Tmax * LDC, nsamples, partial);
    DS2_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_spect_finalize, dim3(ds2_cdiv(N, 64)), dim3(64), 0, st, (const double*)partial, nsamples, N, ms);
    DS2_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_spect_write, dim3(ds2_cdiv(Tmax, 64), 3, N), dim3(256), 0, st, (const float*)Cbuf, (long)Tmax * LDC, nsamples,
                     (const float*)ms, normalize, out, Tmax);
  DS2_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
