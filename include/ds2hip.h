/* ds2hip.h -- C ABI of libds2hip.so: MI355X (gfx950) kernels for the DeepSpeech2 train-step hot path.
 *
 * The reference (SeanNaren/deepspeech.pytorch) is pure Python and has NO FFI/plugin interface: its boundary for this
 * path is the Python class deepspeech_pytorch.model.DeepSpeech (model.py:138-310).  This header is the C-ABI seam
 * placed directly beneath that class: one entry per stage of DeepSpeech.forward / training_step, each citing the
 * reference lines it replaces.  A maintainer binds it with ctypes; the shipped binding is
 * deepspeech/pytorch_amd/_lib.py and the drop-in class is deepspeech/pytorch_amd/model.py.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer unless named *_host.  The library never allocates or frees device memory and
 *    never synchronises: the caller owns all buffers (torch's caching allocator) and passes the stream (a hipStream_t) to launch
 *    on (torch.cuda.current_stream().cuda_stream).  Re-entrant, no thread-local state (backward runs on autograd's
 *    worker thread).
 *  - Return value: 0 = ok; >0 = hipError_t of a failed launch; DS2_ERR_* (>= 1000) = argument errors.
 *  - dtype selects the activation STORAGE type (and the MFMA operand type): DS2_F32 or DS2_BF16 (raw uint16 bits).
 *    Accumulation, statistics, gate math, CTC and all parameter gradients are fp32.
 *  - "T" below means the storage type selected by dtype.
 *  - Activation layouts (internal to this library; only the class boundary must match the reference):
 *      conv activations   NFTC : [N][F][T'][32]   (channel fastest)                 reference: NCHW (N,32,F,T')
 *      sequence matrices  rows = t*N + n, features fastest: [T'*N][ld]              reference: (T',N,H)
 *      conv->RNN features are ordered f*32+c (reference model.py:219-221 orders c*41+f; the binding permutes the
 *      columns of rnns.0 weight_ih accordingly, parameters themselves keep the reference layout).
 */
#ifndef DS2HIP_H
#define DS2HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* a hipStream_t, passed as a plain pointer so that the header needs no HIP headers */
typedef void* ds2_stream_t;

enum { DS2_F32 = 0, DS2_BF16 = 1 };
enum { DS2_CELL_GRU = 0, DS2_CELL_LSTM = 1, DS2_CELL_RNN_TANH = 2 };
enum {
  DS2_OK = 0,
  DS2_ERR_DTYPE = 1001, /* unknown dtype */
  DS2_ERR_ARG = 1002,   /* bad dimension / null pointer / unsupported combination */
  DS2_ERR_ALIGN = 1003  /* pointer or leading dimension not 16-byte aligned / not a multiple of the vector width */
};

int ds2_version(void);
const char* ds2_error_string(int code);

/* ---- dense contraction (MFMA): C[M][ldc] = A[M][lda] * B[N][ldb]^T (+ bias[N]) -------------------------------------
 * Replaces the GEMMs inside torch's GRU/LSTM input projection (model.py:97-99), nn.Linear of the head (model.py:197)
 * and their autograd backward.  out_f32 != 0 -> C is float regardless of dtype.  splitk > 1: C is f32 and is zeroed by the call (memset nodes on `stream`). */
int ds2_gemm_nt(int dtype, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb,
                long ldc, int out_f32, int batch, long strideA, long strideB, long strideC, long strideBias, int splitk,
                ds2_stream_t stream);

/* same contract as ds2_gemm_nt; selects the low-register kernel variant whose waves can share a CU with the persistent
 * recurrent kernels (weight-gradient GEMMs issued on a second stream while a sweep runs). */
/* A as two row blocks (rows [0,m_split) from A, the rest from A2, same lda); coresident != 0: the low-register kernel. */
int ds2_gemm_nt_rows2(int dtype, const void* A, const void* A2, int m_split, const void* B, void* C, int M, int N, int K, long lda,
                      long ldb, long ldc, int out_f32, int splitk, int coresident, ds2_stream_t stream);
int ds2_gemm_nt_coresident(int dtype, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda,
                           long ldb, long ldc, int out_f32, int batch, long strideA, long strideB, long strideC, long strideBias,
                           int splitk, ds2_stream_t stream);

/* 256x256-tile bf16 kernel with a phase-split schedule (csrc/ds2_gemm8.hip) for the large contractions of the step.
 * ds2_gemm8_nt: same product as ds2_gemm_nt (bf16 operands; C bf16 or f32); K % 64 == 0, lda/ldb % 8 == 0.  The input
 * projections of torch's GRU/LSTM and the dX products of their backward (model.py:97-99).
 * ds2_gemm8_tn_grouped: n_problems (<= 6) products C_i[M_i][ldc_i] f32 = sum_{k<K} At_i[k][m] * Bt_i[k][n] in ONE launch -- both
 * operands K-major, i.e. stored as the activations are ([T'*N rows][features]): the weight gradients dW_ih = dGI^T X and
 * dW_hh = dGH^T h_prev of the recurrent layers' backward without operand transposes.  At2_i (may be NULL): output rows >=
 * m_split_i (a multiple of 256) take their operand columns from At2_i (column m - m_split_i, leading dimension lda2_i) -- the GRU's hidden-side
 * gate gradient [dr, dz | dQ] lives in two buffers.  Any K (no row past K - 1 is read); M_i, N_i, lda, ldb % 8 == 0. */
int ds2_gemm8_nt(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb, long ldc, int out_f32,
                 ds2_stream_t stream);
int ds2_gemm8_tn_grouped(int n_problems, const void* const* At, const void* const* At2, const int* m_split, const void* const* Bt, void* const* C,
                         const int* M, const int* N, const long* lda, const long* lda2, const long* ldb, const long* ldc, int K, ds2_stream_t stream);

/* ds2_gemm8_wgrad_dx: the TN problems of ds2_gemm8_tn_grouped (n_tn <= 5) AND one NT product (bf16 out, no bias; the dX of the
 * layer: same contract as ds2_gemm8_nt) in ONE launch, long tiles first -- the backward of one recurrent layer behind its sweep. */
int ds2_gemm8_wgrad_dx(int n_tn, const void* const* At, const void* const* At2, const int* m_split, const void* const* Bt, void* const* C,
                       const int* M, const int* N, const long* lda, const long* lda2, const long* ldb, const long* ldc, int K_tn,
                       const void* A_nt, const void* B_nt, void* C_nt, int M_nt, int N_nt, int K_nt, long lda_nt, long ldb_nt, long ldc_nt,
                       ds2_stream_t stream);

/* Row lists: the `_rows` forms of the three entries above visit only the listed rows of the dimension that runs over the
 * [T' x batch] frames of a padded sequence matrix -- the frames t < length of every clip, i.e. what pack_padded_sequence keeps
 * (model.py:96; the reference's nn.GRU / nn.LSTM never see the padding, model.py:97-100).  rows: device int32 [n_rows], values in
 * [0, n_phys); null = every row.
 *   ds2_gemm8_nt_rows:          C[rows[r]][:] = A[rows[r]][:] * B^T (+ bias), r < n_rows; the other rows of C are NOT written.
 *   ds2_gemm8_tn_grouped_rows:  C_i = sum over k < n_rows of At_i[rows[k]][m] * Bt_i[rows[k]][n].
 *   ds2_gemm8_wgrad_dx_rows:    both (K_tn == M_nt == n_phys).
 * ds2_zero_pad_rows: X[(t*N + n)][0..cols) = 0 for t >= lens[n] (the rows a `_rows` product leaves unwritten). */
int ds2_gemm8_nt_rows(const void* A, const void* B, void* C, const float* bias, int n_phys, int N, int K, long lda, long ldb, long ldc,
                      int out_f32, const int* rows, int n_rows, ds2_stream_t stream);
int ds2_gemm8_tn_grouped_rows(int n_problems, const void* const* At, const void* const* At2, const int* m_split, const void* const* Bt,
                              void* const* C, const int* M, const int* N, const long* lda, const long* lda2, const long* ldb, const long* ldc,
                              int n_phys, const int* rows, int n_rows, ds2_stream_t stream);
int ds2_gemm8_wgrad_dx_rows(int n_tn, const void* const* At, const void* const* At2, const int* m_split, const void* const* Bt, void* const* C,
                            const int* M, const int* N, const long* lda, const long* lda2, const long* ldb, const long* ldc, int K_tn,
                            const void* A_nt, const void* B_nt, void* C_nt, int M_nt, int N_nt, int K_nt, long lda_nt, long ldb_nt, long ldc_nt,
                            const int* rows, int n_rows, ds2_stream_t stream);
int ds2_zero_pad_rows(int dtype, void* X, long ld, int cols, const int* lens, int Tp, int N, ds2_stream_t stream);

/* ---- BatchNorm (model.py:159,162 BatchNorm2d in NFTC; model.py:28-33,86,196 SequenceWise BatchNorm1d) ---------------
 * mode 0: sequence matrix X[R][ldx], C features.   mode 1: conv activation NFTC (R = N*F*Tp rows, C = 32): output also
 * gets Hardtanh(0,20) (model.py:160,163) and the MaskConv time mask (model.py:61-68; t >= lens[n] -> 0).
 * mode 2: like 1 but Y (fwd) / G (bwd) are in sequence layout [(t*N+n)][f*32+c] (fuses model.py:219-221).
 * training != 0: batch statistics over ALL rows (biased variance), running stats updated with `momentum` (unbiased
 * variance), num_batches_tracked += 1.  save_* (C floats each) are kept for backward.  ws: 2*ds2_norm_partials(R)*C floats.*/
int ds2_norm_partials(long R);
int ds2_bn_fwd(int dtype, int mode, int training, const void* X, void* Y, long R, int C, long ldx, long ldy, int F, int Tp,
               int N, const int* lens, const float* gamma, const float* beta, float* running_mean, float* running_var,
               long long* num_batches_tracked, float eps, float momentum, float* save_mean, float* save_rstd,
               float* save_scale, float* save_shift, float* ws, ds2_stream_t stream);
/* G = upstream gradient (gated by Hardtanh' and the mask in modes 1/2), X = the forward input of the BN.
 * ws: (2*ds2_norm_partials(R) + 2) * C floats. */
int ds2_bn_bwd(int dtype, int mode, const void* G, const void* X, void* DX, long R, int C, long ldg, long ldx, long lddx,
               int F, int Tp, int N, const int* lens, const float* save_mean, const float* save_rstd,
               const float* save_scale, const float* save_shift, float* dgamma, float* dbeta, float* ws,
               ds2_stream_t stream);
/* out[c] = scale * sum_r X[r][c]  (bias gradients; ws: ds2_norm_partials(R)*C floats) */
int ds2_colsum(int dtype, const void* X, long R, int C, long ld, float* out, float scale, float* ws, ds2_stream_t stream);

/* ---- conv front-end (MaskConv over model.py:157-164) ------------------------------------------------------------------
 * F0 = input frequency bins = sample_rate * window_size / 2 + 1 (model.py:166: 161 at 16 kHz / 20 ms, 81 at 8 kHz); F1, F2 = rows
 *      after the two convolutions (model.py:167-168; ds2_conv_rows: 161 -> 81 -> 41).  The bf16 matrix-pipe kernels are specialised
 *      for 161 bins; every other geometry runs the general kernels of the same storage type.
 * conv1: Conv2d(1,32,(41,11),stride (2,2),pad (20,5)) on x (N,1,F0,T) f32 -> y1 NFTC [N][F1][Tp][32] (T), bias added,
 *        time mask applied.  w1k = weight re-laid as [41*11][32] f32 (tap-major, out-channel fastest).
 * conv2: Conv2d(32,32,(21,11),stride (2,1),pad (10,5)) on a1 NFTC [N][F1][Tp][32] -> y2 NFTC [N][F2][Tp][32];
 *        w2t = weight re-laid as [21*11][32 out][32 in] (T).
 * dgrad: da1 = conv2^T(dy2) with w2d = the two stride-parity sub-kernels, layout documented in _lib.py/prep. */
int ds2_conv_rows(int F0, int* F1, int* F2);
int ds2_conv1_fwd(int dtype, const float* x, const float* w1k, const float* b1, const int* lens, void* y1, int N, int F0, int T,
                  int Tp, ds2_stream_t stream);
/* dw1k [451][32] f32 = sum over positions of dy1 (NFTC, T) x input taps.  ws: ds2_conv1_wgrad_ws_floats(N,F0,Tp) floats. */
long ds2_conv1_wgrad_ws_floats(int N, int F0, int Tp);
int ds2_conv1_wgrad(int dtype, const float* x, const void* dy1, float* dw1k, int N, int F0, int T, int Tp, float* ws,
                    ds2_stream_t stream);
/* ws: ds2_conv2_fwd_ws_bytes(dtype,N,F0,Tp) bytes (bf16 storage at 161 bins: fp32 partial sums of the even-kernel-row half of the
 * layer pass through it; 0 and ws may be NULL otherwise). */
long ds2_conv2_fwd_ws_bytes(int dtype, int N, int F0, int Tp);
int ds2_conv2_fwd(int dtype, const void* a1, const void* w2t, const float* b2, const int* lens, void* y2, int N, int F0, int Tp,
                  void* ws, ds2_stream_t stream);
int ds2_conv2_dgrad(int dtype, const void* dy2, const void* w2d_even, const void* w2d_odd, void* da1, int N, int F0, int Tp,
                    ds2_stream_t stream);
/* dw2t [231][32 out][32 in] f32.  ws: ds2_conv2_wgrad_ws_floats(N,Tp) floats. */
long ds2_conv2_wgrad_ws_floats(int N, int Tp);
int ds2_conv2_wgrad(int dtype, const void* dy2, const void* a1, float* dw2t, int N, int F0, int Tp, float* ws, ds2_stream_t stream);

/* ---- recurrent sweeps (BatchRNN, model.py:94-102; nn.GRU / nn.LSTM / nn.RNN(tanh), enums.py:17-21) ---------------------
 * See csrc/ds2_rnn.hip for the buffer shapes.  H % 16 == 0. */
int ds2_rnn_gates(int cell);
int ds2_rnn_saved_planes(int cell);
long ds2_rnn_state_bytes(int D, int N, int H);
int ds2_rnn_fwd(int dtype, int cell, int D, int N, int H, int Tp, const int* lens, const void* GI, const void* Whh,
                const float* bhh, const float* h0, const float* c0, void* Hseq, long hseq_dstride, void* S, float* hn, float* cn,
                void* state, ds2_stream_t stream);
/* h0 / c0 [D][N][H] f32 (may be NULL = zeros): the initial state the forward was given (`hs`, model.py:224-230); dh0 / dc0 (may be
 * NULL): d loss / d h0, d c0.  Only the launch-per-time-step BPTT takes them (training through a given `hs` is the rare path; the
 * persistent sweeps assume a zero initial state in backward). */
int ds2_rnn_bwd(int dtype, int cell, int D, int N, int H, int Tp, const int* lens, const void* dOut, const void* WhhT,
                const void* Hseq, long hseq_dstride, const void* S, void* dGI, void* dGH, const float* h0, const float* c0,
                float* dh0, float* dc0, void* state, ds2_stream_t stream);

/* Persistent variant (csrc/ds2_rnn_persist.hip): one launch per sweep, all time steps inside the kernel, W_hh resident in
 * registers, h exchanged between the workgroups of a group through a polled exchange buffer inside ws (pure payload in four
 * slots with an all-ones "not published yet" dword, or tagged 8-byte granules: csrc/ds2_rnn_persist_impl.h); same buffer contract as
 * ds2_rnn_fwd / ds2_rnn_bwd.  Covered (ds2_rnn_persist_supported): bf16 with H = 1024 (any cell, <= 16 samples per group of an
 * 8-group chip: BASELINE config 3); GRU / LSTM with bf16 H in {512, 768, 800, 1024, 1280, 1536} (LSTM: not 1536) and <= 32
 * samples per group (round-4 general kernels: 32 units per workgroup, groups inside one XCD for H <= 1024 -- config 5); GRU /
 * LSTM with fp32 H in {800, 1024} or bf16 H in {800, 1280} and <= 64 samples per group (round-2 general kernels: config 2, the 1e-3
 * parity mode); everything else runs ds2_rnn_fwd / _bwd.
 * ws: ds2_rnn_persist_ws_bytes() bytes of scratch (reset by every call on `stream`); err: one device int, sticky (maximum): 1 = a
 * workgroup gave up waiting for its peers in the middle of a sweep, 2 = the launch's workgroups never became co-resident within
 * opts->startup_ms; the outputs are then NaN-poisoned. */
/* Per-launch options of the persistent sweeps; pass NULL for the defaults.  (Rounds 2-5 had process-wide setters here --
 * ds2_rnn_persist_set_variant / _set_spin_limit; options travel with the call now: the entries are re-entrant, backward may run on
 * autograd's worker thread beside a forward, and nothing a caller sets outlives its call.)
 *   variant     routing A/B bits for measurements, 0 = the shipping routing: bit 0 the shapes of the round-4 general kernels
 *               (csrc/ds2_rnn_persist3_impl.h) run on the round-2 general kernels (or one launch per time step) instead, 1 two-set
 *               groups execute every half-step, 3 the round-4 kernels also take H = 1024 with <= 8 clips per group, 4 the tuned kernels
 *               keep 9-16 clips per group, 5 the 8-clip tuned kernels use dense products instead of the structured-sparse ones.
 *               The queries below take the same bits, so that a caller sizes the scratch for the routing it will launch.
 *   spin_limit  fault injection: polls a workgroup may spend on ONE mid-sweep exchange wait before it gives up (raises *err = 1,
 *               NaN-poisons its outputs, ends); 0 = the built-in budget (~seconds).
 *   startup_ms  how long (wall clock) the workgroups of a launch wait for ALL of them to become resident before the first exchange;
 *               a sweep needs every workgroup on a CU of its own at the same time, so a kernel of another process -- or an RCCL
 *               collective on another stream that waits for a late peer rank -- keeps it from starting.  0 = 300 ms: right for a
 *               single process (nothing legitimate holds CUs that long; *err = 2 names the cause).  Under data parallelism pass
 *               the process group's time-out: a sweep behind a collective must wait, as any stock kernel would simply queue
 *               (configs/librispeech.yaml:14 `strategy: ddp`; loader/data_loader.py:320-360 hands ranks unequal batches). */
typedef struct ds2_persist_opts {
  unsigned variant;
  unsigned spin_limit;
  unsigned startup_ms;
} ds2_persist_opts;
int ds2_rnn_persist_supported(int dtype, int cell, int D, int N, int H, unsigned variant);
/* 1 if a persistent kernel exists for the problem on a FULL device (256 CUs), whatever the current device exposes: tells "this
 * device is too small for the persistent sweeps" (the caller raises) from "no persistent kernel for this shape" (it warns). */
int ds2_rnn_persist_shape_covered(int dtype, int cell, int D, int N, int H);
/* Kernel family the persistent entries run for the problem on the current device: 0 none (ds2_rnn_fwd / _bwd), 1 / 2 tuned H = 1024
 * (<= 8 / 9-16 samples per group), 3 round-4 general (k_rnn_persist3_*), 4 round-2 general (k_rnn_persist2_*). */
int ds2_rnn_persist_kind(int dtype, int cell, int D, int N, int H, unsigned variant);
long ds2_rnn_persist_ws_bytes(int dtype, int cell, int D, int N, int H, unsigned variant);
int ds2_rnn_persist_fwd(int dtype, int cell, int D, int N, int H, int Tp, const int* lens, const void* GI, const void* Whh,
                        const float* bhh, const float* h0, const float* c0, void* Hseq, long hseq_dstride, void* S, float* hn,
                        float* cn, void* ws, int* err, const ds2_persist_opts* opts, ds2_stream_t stream);
/* BPTT: dGI as ds2_rnn_bwd; GRU: dQ [D][Tp][N][H] = dn*r (the one slot of the hidden-side gate gradient [dr,dz,dQ] that differs
 * from dGI's; null for LSTM / RNN); dBacc [D][N][NB*H] f32 (may be null): per-sample sums over time of the stored gate-gradient
 * planes (NB = 4 for GRU: dr, dz, dn, dQ; G otherwise) -- the bias gradients are their sums over the samples. */
/* flags bit 0: the caller never reads the padding rows (t >= lens[n]) of dGI / dQ -- its products over the frames run on a row list
 * (ds2_gemm8_*_rows) and its bias gradients come from dBacc -- so a sweep whose half-steps leave those rows unwritten (two-set groups
 * of the round-4 general kernels) skips zeroing them. */
int ds2_rnn_persist_bwd(int dtype, int cell, int D, int N, int H, int Tp, const int* lens, const void* dOut, const void* WhhT,
                        const void* Hseq, long hseq_dstride, const void* S, void* dGI, void* dQ, float* dBacc, int flags, void* ws,
                        int* err, const ds2_persist_opts* opts, ds2_stream_t stream);

/* ---- small sequence ops ---------------------------------------------------------------------------------------------------
 * add2: out = a + b (direction sum, model.py:101).  transpose: dst[C][ldd] = src[R][lds]^T, zero-filling r in [R, ldd).
 * lookahead (model.py:105-135, uni-directional models): y[t] = hardtanh(sum_k w[h][k] * x[t+k]), x/y [Tp*N][H] (T),
 * w [H][ctx] f32, any ctx >= 1 (ctx = 20, the reference default, takes the sliding-window kernels: one load per operand and frame);
 * `pre` keeps the pre-Hardtanh value for backward (may be null in eval).
 * bwd: dx and dw (dw via ws of ds2_lookahead_ws_floats). */
int ds2_add2(int dtype, const void* a, const void* b, void* out, long n, ds2_stream_t stream);
/* out[n] f32 = the sum of `slices` consecutive [n] blocks of src, added in index order: the fixed-order reduction of a product whose
 * contraction was cut into K-slices (ds2_gemm_nt with the slices as its batch) -- the fp32-mode dX of nn.GRU / nn.LSTM (model.py:97-99
 * backward: few output tiles, long K) without the run-to-run summation order of atomic split-K.  n % 4 == 0. */
int ds2_sum_slices(const float* src, float* out, long n, int slices, ds2_stream_t stream);
int ds2_transpose(int dtype, const void* src, void* dst, long R, int C, long lds, long ldd, ds2_stream_t stream);
/* fp32 operand [rows][K] (row stride lds) -> bf16 [rows][3 * Kp] (row stride ldd; Kp = K rounded up to 64, zero beyond K): the K
 * segments [hi | hi | lo] (mode 0, the A operand of a product) or [hi | lo | hi] (mode 1, the B operand) with hi = bf16(x),
 * lo = bf16(x - hi).  One bf16 GEMM over K' = 3 Kp of two such operands = a_hi b_hi + a_hi b_lo + a_lo b_hi: the fp32-mode (1e-3
 * parity) input projections / dX / weight gradients of nn.GRU / nn.LSTM (model.py:97-99) on the bf16 matrix pipe.
 * Modes 2 / 3 (round 6): the same three segments stacked by ROWS -- dst [3 * rows][ldd], segment s at row s * rows, Kp = K rounded up
 * to 8 -- i.e. the operands of a TN product (ds2_gemm8_tn_grouped: the contraction index is the row), so that the fp32-mode weight
 * gradients run over the activations as stored, without operand transposes. */
int ds2_split3_bf16(const float* src, long lds, long rows, int K, int Kp, int mode, void* dst, long ldd, ds2_stream_t stream);
/* Weight re-layout of the recurrent layers (what nn.GRU/LSTM/RNN.flatten_parameters + the autocast weight casts do in the
 * reference, model.py:97-99): fp32 src[R][C] -> bf16 dst[R][ldd] and/or bf16 transpose dstT[Cout][lddT] in one pass.
 * perm_c > 0: output column j = f*perm_c + c takes source column c*perm_f + f (rnns.0 reads the conv features in the
 * kernels' [f][c] order; the reference flattens [c][f], model.py:219-220); columns [C, Cout) are zero.  R % 16 == 0. */
int ds2_cast_transpose_bf16(const float* src, long lds, int R, int C, int perm_c, int perm_f, int Cout, void* dst, long ldd,
                            void* dstT, long lddT, ds2_stream_t stream);
/* Kernel layouts of the small weights in one launch (the per-step re-layout of conv.seq_module.{0,3}.weight and fc weight after
 * the optimizer step; model.py:158,161,197): w1k [451][32] f32 = conv1 weight tap-major; w2t [21][11][co][ci] (T) conv2 forward;
 * w2d0 [11][11][ci][co] / w2d1 [10][11][ci][co] (T) = the flipped even / odd kernel-row sub-kernels of the conv2 data gradient;
 * wfcp [32][H] (T) = head weight zero-padded to 32 classes, wfcT [H][32] (T) its transpose.
 * ds2_scale_by: x[i] *= *s (s on the device) -- the upstream gradient of the summed CTC loss. */
int ds2_small_weight_layouts(int dtype, const float* w1, const float* w2, const float* wfc, int C, int H, float* w1k, void* w2t,
                             void* w2d0, void* w2d1, void* wfcp, void* wfcT, ds2_stream_t stream);
int ds2_scale_by(float* x, const float* s, long n, ds2_stream_t stream);

/* ds2_copy_words: dst[0..n_words) = src[0..n_words) (4-byte words), as a kernel on `stream`.  Either pointer may be pinned
 * (device-mapped) host memory.  Replaces the reference's implicit host->device transfers of the per-step index tables -- the
 * lengths `.cpu()` / BoolTensor masks `.cuda()` of MaskConv (model.py:61-68,215) and the CTC target table (model.py:248) -- and the
 * device->host read of the persistent sweeps' error word: hipMemcpyAsync would hand these to the SDMA engines, whose transfers run
 * under the kernels of earlier steps when the host thread is ahead and cost a latency-bound sweep 1-2 ms each. */
int ds2_copy_words(const void* src, void* dst, long n_words, ds2_stream_t stream);
int ds2_lookahead_fwd(int dtype, const void* x, const float* w, void* y, void* pre, int Tp, int N, int H, int ctx,
                      ds2_stream_t stream);
long ds2_lookahead_ws_floats(int Tp, int N, int H, int ctx);
int ds2_lookahead_bwd(int dtype, const void* x, const float* w, const void* pre, const void* dy, void* dx, float* dw, int Tp,
                      int N, int H, int ctx, float* ws, ds2_stream_t stream);
/* bias_ih.grad [D][G*H] and bias_hh.grad [D][G*H] of a recurrent layer (torch GRU/LSTM/RNN backward, model.py:97-99) from the
 * per-sample sums dBacc [D][N][NB*H] that ds2_rnn_persist_bwd accumulates (cell: 0 GRU, 1 LSTM, 2 tanh RNN). */
int ds2_rnn_bias_grads(int cell, int D, int N, int H, const float* dBacc, float* dbih, float* dbhh, ds2_stream_t stream);

/* probs = softmax(logits) row-wise (InferenceBatchSoftmax, model.py:72-77), f32 [rows][C] */
int ds2_softmax_rows(const float* logits, float* probs, long rows, int C, long ld_in, long ld_out, ds2_stream_t stream);

/* ---- log_softmax + CTC loss + gradient (model.py:246,203,248) -----------------------------------------------------------
 * logits [Tp*N][ldl] f32 (row = t*N+n, C classes), targets int32 concatenated with target_offsets[N] (start of each
 * sample's labels), input_lengths/target_lengths int32 [N].  blank index `blank`; reduction 'sum'; zero_infinity: an
 * infeasible sample contributes loss 0 and gradient 0.  Outputs: nll [N] f32 (per-sample), loss_sum [1] f32,
 * dlogits [Tp*N][ldg] f32 = grad_scale * d(loss_sum)/d(logits) (log_softmax backward fused; zero rows for t >= length;
 * columns >= C are zero-filled up to ldg).  ws: ds2_ctc_ws_floats(...) floats.  max_target_len: max over target_lengths.
 * recursion: which recursion kernel runs (bit-identical results; for A/B runs and tests): 0 = the default choice (the pair-tile
 * kernel for targets of up to 783 labels, the four-wave kernel beyond), 1 = always the four-wave kernel, 2 = the one-wave kernel up to
 * 255 labels (<= 32 classes), 3 = the choice of rounds 3-5 (one wave up to 63 labels, four waves beyond). */
long ds2_ctc_ws_floats(int Tp, int N, int C, int max_target_len);
int ds2_ctc_loss_grad(const float* logits, long ldl, const int* targets, const int* target_offsets, const int* input_lengths,
                      const int* target_lengths, int Tp, int N, int C, int blank, int max_target_len, float grad_scale,
                      float* nll, float* loss_sum, float* dlogits, long ldg, float* ws, int recursion, ds2_stream_t stream);

/* ---- optimizer step (configure_optimizers, model.py:273-297; Lightning's gradient_clip_val, configs/an4.yaml:12) -----------
 * ds2_clip_coef: out[0] = global L2 norm of `count` fp32 gradient tensors, out[1] = min(1, max_norm / (norm + 1e-6)) -- the
 * coefficient torch.nn.utils.clip_grad_norm_ multiplies the gradients by; computed on the device, deterministic order.
 * ds2_opt_multi / ds2_opt_matrix: mode 0 = AdamW, 1 = SGD with Nesterov momentum, fp32, torch's single-tensor arithmetic;
 * hp[7] = {AdamW: 1-lr*wd, 1-beta1, beta2, 1-beta2, sqrt(1-beta2^t), eps, -lr/(1-beta1^t) | SGD: wd, momentum, 0, 0, 1, 0, -lr};
 * first = 1 on SGD's first step; clip = out of ds2_clip_coef (device) or null; v is ignored for SGD.  ds2_opt_matrix also
 * writes the bf16 copy / transpose of the updated matrix (same layout parameters as ds2_cast_transpose_bf16). */
int ds2_opt_max_tensors(void);
long ds2_clip_ws_floats(int count, const long* n);
int ds2_clip_coef(int count, const float* const* g, const long* n, float max_norm, float* out, float* ws, ds2_stream_t stream);
int ds2_opt_multi(int mode, int count, float* const* p, const float* const* g, float* const* m, float* const* v, const long* n,
                  const float* hp, int first, const float* clip, ds2_stream_t stream);
int ds2_opt_matrix(int mode, float* p, const float* g, float* m, float* v, int R, int C, int perm_c, int perm_f, int Cout,
                   void* dst, long ldd, void* dstT, long lddT, const float* hp, int first, const float* clip, ds2_stream_t stream);
/* ds2_opt_matrix for `count` matrices of ONE parameter group (arrays of the per-matrix arguments): every matrix without a column
 * permutation (perm_c == 0, Cout == C, C % 4 == 0, 16-byte aligned) is updated in one launch per 36 of them, the others one by one. */
int ds2_opt_matrices(int mode, int count, float* const* p, const float* const* g, float* const* m, float* const* v, const int* R,
                     const int* C, const int* perm_c, const int* perm_f, const int* Cout, void* const* dst, const long* ldd,
                     void* const* dstT, const long* lddT, const float* hp, int first, const float* clip, ds2_stream_t stream);

/* ---- log-spectrogram front-end (SpectrogramParser.compute_spectrogram, loader/data_loader.py:73-94, + the padded batch layout
 * of _collate_fn, :247-270).  wav [N][ldw] f32 waveforms (utterance n = first nsamples[n] entries of row n; nsamples on the
 * device), n_fft 320 / hop 160 / center = True; basis [322][320] f32 = window-weighted cos rows then -sin rows; reflect: 0 zero
 * centre padding (librosa >= 0.10), 1 reflection; normalize: (x - mean) / unbiased std per utterance.  out (N,1,161,Tmax) f32,
 * Tmax = ds2_spect_frames(Lmax) = 1 + Lmax/160, zero beyond an utterance's own frames.  ws: ds2_spect_ws_bytes(N, Lmax). */
int ds2_spect_frames(int nsamples);
long ds2_spect_ws_bytes(int N, int Lmax);
int ds2_spectrogram(const float* wav, long ldw, const int* nsamples, int N, int Lmax, const float* basis, int reflect,
                    int normalize, float* out, void* ws, ds2_stream_t stream);

/* ---- greedy CTC decoding on the device (validation_step, model.py:256 -> GreedyDecoder.decode, decoder.py:164-181) -------
 * x[n*stride_n + t*stride_t + c] f32 scores (probabilities or logits), C <= 64; sizes [N] int32 on the device (null = T).
 * Per sample: arg-max per frame (first maximum), repeats collapsed, blanks dropped.  tokens / offsets [N][T] int32 (the first
 * counts[n] entries of row n are valid: label index and its frame), counts [N]. */
int ds2_greedy_decode(const float* x, long stride_n, long stride_t, int N, int T, int C, const int* sizes, int blank,
                      int* tokens, int* offsets, int* counts, ds2_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DS2HIP_H */
