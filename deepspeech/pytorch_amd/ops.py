"""Torch-tensor wrappers around the C ABI (include/ds2hip.h).  torch is used here only as the owner of device
memory and streams; every computation is a ds2hip kernel.  All tensors must be CUDA(HIP) tensors -- there is no CPU
path (a CPU tensor raises)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import F32, BF16, call, query

CELLS = {"gru": _lib.CELL_GRU, "lstm": _lib.CELL_LSTM, "rnn": _lib.CELL_RNN_TANH}
GATES = {"gru": 3, "lstm": 4, "rnn": 1}
SAVED_PLANES = {"gru": 4, "lstm": 5, "rnn": 0}


def dt(t):
    d = t if isinstance(t, torch.dtype) else t.dtype
    if d == torch.float32:
        return F32
    if d == torch.bfloat16:
        return BF16
    raise TypeError("ds2hip supports float32 and bfloat16 activations, got %s" % d)


def vecw(dtype):
    return 4 if dtype == torch.float32 else 8


def P(t):
    """device pointer of a tensor (None -> NULL); refuses CPU tensors: the HIP path is the only path."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise _lib.Ds2HipError("ds2hip ops need CUDA(HIP) tensors; got a %s tensor -- there is no CPU fallback" % t.device)
    return C.c_void_p(t.data_ptr())


def PF(t):
    """device pointer of an fp32 parameter / buffer / statistics tensor.  The kernels read these as float: a module cast with
    .half() / .to(bfloat16) (or a 'bf16-true' precision plugin) must fail loudly instead of being read as garbage."""
    if t is not None and t.dtype != torch.float32:
        raise _lib.Ds2HipError("ds2hip kernels keep parameters, BatchNorm statistics and carried states in float32 (bf16 is an "
                               "activation/operand type selected by `precision`/autocast); got %s" % t.dtype)
    return P(t)


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rup(x, m):
    return (x + m - 1) // m * m


# Optional measurement hook (bench.py): when set to a list, every recurrent sweep appends
# (tag, n_launches, start_event, end_event) recorded on the stream the kernels are launched on.
SWEEP_EVENTS = None


class _sweep_timer:
    def __init__(self, tag, launches):
        self.tag, self.launches = tag, launches

    def __enter__(self):
        if SWEEP_EVENTS is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if SWEEP_EVENTS is not None:
            self.e1.record()
            SWEEP_EVENTS.append((self.tag, self.launches, self.e0, self.e1))
        return False


# ---------------------------------------------------------------------------------------------------------------
def pad_ld(n, dtype=torch.bfloat16):
    """Leading dimension for a matrix with n columns that the large GEMMs stream by rows: a row stride that is a multiple of
    1 KiB maps every row of a tile onto the same few memory channels (measured: the K = 1024 input projection runs 714 TFLOP/s
    with ld 1024 and 800-846 with ld 1088 / 1152, profiles/r03b_gemm8_ld.txt) -- such widths get 64 extra columns."""
    esz = 2 if dtype == torch.bfloat16 else 4
    return n + 64 if (n * esz) % 1024 == 0 else n


def empty_padded(rows, cols, dtype, device):
    """[rows][cols] view (row stride pad_ld(cols)) of a fresh allocation."""
    ld = pad_ld(cols, dtype)
    return torch.empty((rows, ld), dtype=dtype, device=device)[:, :cols] if ld != cols else torch.empty((rows, cols), dtype=dtype, device=device)


GEMM8_ENABLED = True     # tests flip this to run the 128x128-tile kernels on shapes the 256x256 kernel covers


def gemm8_nt_shape_ok(dtype, M, N, K, lda, ldb):
    """The shape part of gemm8_nt_ok (decidable before the operands exist)."""
    return (GEMM8_ENABLED and dtype == torch.bfloat16 and K % 64 == 0 and lda % 8 == 0 and ldb % 8 == 0 and
            M >= 1024 and N >= 256 and rup(M, 256) // 256 * (rup(N, 256) // 256) >= 160 and M * lda < (1 << 31) and N * ldb < (1 << 31))


def gemm8_nt_ok(A, B, M, N, K, lda, ldb):
    """Shapes the 256x256 phase-split kernel takes (and wins on): bf16, whole K-tiles, enough tiles to fill the chip."""
    return (A.dtype == B.dtype and gemm8_nt_shape_ok(A.dtype, M, N, K, lda, ldb) and
            A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def wgrad_tn_ok(dtype, R, M, N, lda=None, ldb=None):
    """Weight gradients as grouped TN products (no operand transposes): bf16 activations, a long contraction, 8-aligned widths, and
    operands whose R x ld elements stay inside the kernels' 32-bit offsets (ds2_gemm8_tn_grouped / _wgrad_dx require K * ld < 2^31:
    beyond it -- e.g. R = T'N = 256 000 rows of a BiLSTM-1280's dGI, ld 10 240 -- the transpose + NT path takes over)."""
    lda = M if lda is None else lda
    ldb = N if ldb is None else ldb
    return (GEMM8_ENABLED and dtype == torch.bfloat16 and R >= 512 and M % 8 == 0 and N % 8 == 0 and M >= 256 and N >= 256 and
            R * lda < (1 << 31) and R * ldb < (1 << 31))


def zero_pad_rows(X, lens, Tp, N):
    """X[(t*N + n)][:] = 0 for t >= lens[n]: the rows a row-list product leaves unwritten."""
    call("ds2_zero_pad_rows", dt(X), P(X), X.stride(0), X.shape[1], P(lens), Tp, N, S())
    return X


def gemm_nt(A, B, bias=None, out_dtype=None, M=None, N=None, K=None, lda=None, ldb=None, out=None, ldc=None, splitk=1,
            batch=1, sA=0, sB=0, sC=0, sBias=0, coresident=False, rows=None, zero_pad=None):
    """C[M][N] = A[M][K] * B[N][K]^T (+bias).  A, B: 2-D row-major tensors (or explicit M/N/K/ld for views).
    rows (int32 device tensor or None): on the 256x256 kernel only the listed rows of A / C are visited (the frames of a padded
    sequence matrix that pack_padded_sequence keeps, model.py:96); zero_pad = (lens, Tp, N): the other rows of C are then zero,
    without it they are unspecified (unwritten by the 256x256 kernel; the kernels without row lists compute every row)."""
    M = A.shape[-2] if M is None else M
    N = B.shape[-2] if N is None else N
    K = A.shape[-1] if K is None else K
    lda = A.stride(-2) if lda is None else lda
    ldb = B.stride(-2) if ldb is None else ldb
    out_dtype = out_dtype or A.dtype
    if out is None and splitk == 1 and batch == 1 and not coresident and gemm8_nt_ok(A, B, M, N, K, lda, ldb):
        res = gemm8_nt(A, B, bias=bias, out_dtype=out_dtype, M=M, N=N, K=K, lda=lda, ldb=ldb, rows=rows)
        if rows is not None and zero_pad is not None:
            zero_pad_rows(res, *zero_pad)
        return res
    ks = fp32_ksplit(A.dtype, M, N, K) if (out is None and splitk == 1 and batch == 1 and not coresident and bias is None and
                                            out_dtype == torch.float32 and A.dtype == torch.float32) else 1
    if ks > 1:
        # fp32 mode, few output tiles and a long contraction (a layer's dX at config 2: 49 tiles of K = 4 800 on 256 CUs): the
        # K-slices run as the kernel's BATCH into a [slices][M][N] scratch and are added in index order (ds2_sum_slices) -- split-K
        # without atomics, i.e. with a fixed summation order
        Ks = K // ks
        ws = torch.empty((ks, M, N), dtype=torch.float32, device=A.device)
        call("ds2_gemm_nt", dt(A), P(A), P(B), P(ws), None, M, N, Ks, lda, ldb, N, 1, ks, Ks, Ks, M * N, 0, 1, S())
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
        call("ds2_sum_slices", P(ws), P(out), M * N, ks, S())
        if rows is not None and zero_pad is not None:
            zero_pad_rows(out, *zero_pad)
        return out
    if out is None:
        shape = (batch, M, N) if batch > 1 else (M, N)
        out = torch.empty(shape, dtype=out_dtype, device=A.device)       # split-K: the entry zeroes C itself
        ldc = N
        sC = M * N
    out_f32 = 1 if out.dtype == torch.float32 else 0
    call("ds2_gemm_nt_coresident" if coresident else "ds2_gemm_nt", dt(A), P(A), P(B), P(out), PF(bias), M, N, K, lda, ldb, ldc,
         out_f32, batch, sA, sB, sC, sBias, splitk, S())
    if rows is not None and zero_pad is not None and batch == 1:
        zero_pad_rows(out, *zero_pad)          # same contract whichever kernel ran: unlisted rows are zero
    return out


FP32_KSPLIT = __import__("os").environ.get("DS2_FP32_KSPLIT", "1") != "0"    # 0: every exact-fp32 product in one pass (round 5)


def fp32_ksplit(dtype, M, N, K):
    """K-slices of an exact-fp32 product (ops.gemm_nt): > 1 when its 128x128 tiles fill less than half of the chip and the
    contraction is long; a divisor of the K-tile count (32 elements) so that every slice is whole tiles of the same length."""
    if not FP32_KSPLIT or dtype != torch.float32 or K % 32 != 0 or (M * N) % 4 != 0:
        return 1
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    nkt = K // 32
    if tiles >= 128 or nkt < 32:
        return 1
    want = min(8, max(1, 256 // tiles))
    for s_ in range(want, 1, -1):
        if nkt % s_ == 0 and nkt // s_ >= 8:
            return s_
    return 1


def gemm_nt_rows2(A, A2, m_split, B, M, N, K, lda, ldb, splitk=1, coresident=False):
    """C[M][N] f32 = [A rows 0..m_split-1 ; A2 rows 0..M-m_split-1][.][K] * B[N][K]^T: the A operand split over two buffers."""
    out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    call("ds2_gemm_nt_rows2", dt(A), P(A), P(A2), m_split, P(B), P(out), M, N, K, lda, ldb, N, 1, splitk, 1 if coresident else 0, S())
    return out


def gemm8_nt(A, B, bias=None, out_dtype=None, M=None, N=None, K=None, lda=None, ldb=None, rows=None):
    """C[M][N] = A[M][K] * B[N][K]^T (+bias) on the 256x256 phase-split kernel (bf16 operands, K % 64 == 0); rows: see gemm_nt."""
    M = A.shape[-2] if M is None else M
    N = B.shape[-2] if N is None else N
    K = A.shape[-1] if K is None else K
    lda = A.stride(-2) if lda is None else lda
    ldb = B.stride(-2) if ldb is None else ldb
    out = torch.empty((M, N), dtype=out_dtype or A.dtype, device=A.device)
    if rows is not None:
        call("ds2_gemm8_nt_rows", P(A), P(B), P(out), PF(bias), M, N, K, lda, ldb, N, 1 if out.dtype == torch.float32 else 0, P(rows),
             rows.numel(), S())
        return out
    call("ds2_gemm8_nt", P(A), P(B), P(out), PF(bias), M, N, K, lda, ldb, N, 1 if out.dtype == torch.float32 else 0, S())
    return out


def gemm8_tn_grouped(problems, K, dx=None, rows=None, zero_pad=None):
    """problems: list of dicts(At, Bt, M, N, lda, ldb[, At2, lda2, m_split][, out]); every product C[M][N] f32 = sum_{k<K} At[k][m] Bt[k][n]
    in one launch.  Returns the list of outputs.  dx = (A, B): additionally the NT product A[M][Kx] * B[N][Kx]^T (bf16) in the SAME
    launch (the layer's dX beside its weight gradients); returns (outputs, dX) then.  rows: the contraction (and the rows of dX)
    runs over the listed rows only; zero_pad = (lens, Tp, N) zeroes the unlisted rows of dX."""
    rp, nr = (P(rows), rows.numel()) if rows is not None else (None, 0)
    n = len(problems)
    outs = []
    for pr in problems:
        o = pr.get("out")
        if o is None:
            o = torch.empty((pr["M"], pr["N"]), dtype=torch.float32, device=pr["At"].device)
        outs.append(o)
    vp = C.c_void_p * n
    ia, la = C.c_int * n, C.c_long * n
    At = vp(*[pr["At"].data_ptr() for pr in problems])
    At2 = vp(*[(pr["At2"].data_ptr() if pr.get("At2") is not None else 0) for pr in problems])
    ms = ia(*[int(pr.get("m_split", 0)) for pr in problems])
    Bt = vp(*[pr["Bt"].data_ptr() for pr in problems])
    Cs = vp(*[o.data_ptr() for o in outs])
    Ms, Ns = ia(*[pr["M"] for pr in problems]), ia(*[pr["N"] for pr in problems])
    lda, ldb = la(*[pr["lda"] for pr in problems]), la(*[pr["ldb"] for pr in problems])
    lda2 = la(*[int(pr.get("lda2", 0)) for pr in problems])
    ldc = la(*[o.stride(0) for o in outs])
    for pr in problems:
        P(pr["At"]), P(pr["Bt"])            # refuses CPU tensors
    if dx is not None:
        A, B = dx
        Mx, Nx, Kx = A.shape[0], B.shape[0], A.shape[1]
        out_x = torch.empty((Mx, Nx), dtype=A.dtype, device=A.device)
        call("ds2_gemm8_wgrad_dx_rows", n, At, At2, ms, Bt, Cs, Ms, Ns, lda, lda2, ldb, ldc, K, P(A), P(B), P(out_x), Mx, Nx, Kx, A.stride(0),
             B.stride(0), Nx, rp, nr, S())
        if rows is not None and zero_pad is not None:
            zero_pad_rows(out_x, *zero_pad)
        return outs, out_x
    call("ds2_gemm8_tn_grouped_rows", n, At, At2, ms, Bt, Cs, Ms, Ns, lda, lda2, ldb, ldc, K, rp, nr, S())
    return outs


def colsum(X, R=None, Cc=None, ld=None, scale=1.0):
    R = X.shape[0] if R is None else R
    Cc = X.shape[1] if Cc is None else Cc
    ld = X.stride(0) if ld is None else ld
    out = torch.empty(Cc, dtype=torch.float32, device=X.device)
    ws = torch.empty(query("ds2_norm_partials", R) * Cc, dtype=torch.float32, device=X.device)
    call("ds2_colsum", dt(X), P(X), R, Cc, ld, P(out), float(scale), P(ws), S())
    return out


# fp32 (1e-3 parity) mode: a large GEMM can run as ONE bf16 GEMM over three K-segments of split operands (a_hi b_hi + a_hi b_lo +
# a_lo b_hi, fp32 accumulation: ~2e-5 of |a||b| per entry against 6e-8 of the exact-fp32 MFMA, which runs at 1/16 of the bf16 rate).
# Default ("wgrad"): only the WEIGHT-GRADIENT products, whose results are leaves -- 2e-5 stays 2e-5.  The input projections and dX
# feed the recurrence / the layers below, and a deep stack amplifies their noise: with every GEMM split the conv1 weight gradient of
# the 7-layer LSTM-1280 fixture lands 3.8e-3 from the reference (bar 1e-3), although logits and loss stay inside 1e-3
# (DS2_FP32_GEMM=split3: config 2 10.8 ms instead of 12.3; =exact: no split anywhere).
import os as _os
FP32_GEMM_MODE = _os.environ.get("DS2_FP32_GEMM", "wgrad")
FP32_SPLIT3 = FP32_GEMM_MODE != "exact"


def split3_ok(dtype, M, N, K, leaf=False):
    """leaf: the product is a parameter gradient (its error does not propagate)."""
    if dtype != torch.float32 or FP32_GEMM_MODE == "exact" or not (M >= 256 and N >= 256 and K >= 256):
        return False
    return leaf or FP32_GEMM_MODE == "split3"


def split3(X, mode, out=None):
    """fp32 [rows][K] (any row stride) -> bf16 [rows][3 * Kp], Kp = K rounded up to 64: K-segments [hi | hi | lo] (mode 0: the A
    operand of a product) or [hi | lo | hi] (mode 1: the B operand).  `out`: a row window of a larger [.., 3 * Kp] buffer."""
    rows, K = X.shape
    Kp = rup(K, 64)
    if out is None:
        out = torch.empty((rows, 3 * Kp), dtype=torch.bfloat16, device=X.device)
    assert out.shape[0] == rows and out.shape[1] == 3 * Kp and X.dtype == torch.float32 and X.stride(1) == 1
    call("ds2_split3_bf16", P(X), X.stride(0), rows, K, Kp, mode, P(out), out.stride(0), S())
    return out


def split3_rows(X, mode):
    """fp32 [rows][K] (any row stride) -> bf16 [3 * rows][K8] (K8 = K rounded up to 8, zero beyond K): the split segments stacked by
    rows, [hi; hi; lo] (mode 0: the At operand of a TN product) or [hi; lo; hi] (mode 1: the Bt operand).  The contraction of
    ds2_gemm8_tn_grouped over the 3 * rows rows of two such operands = a_hi b_hi + a_hi b_lo + a_lo b_hi."""
    rows, K = X.shape
    K8 = rup(K, 8)
    out = torch.empty((3 * rows, K8), dtype=torch.bfloat16, device=X.device)
    assert X.dtype == torch.float32 and X.stride(1) == 1
    call("ds2_split3_bf16", P(X), X.stride(0), rows, K, K8, 2 + mode, P(out), K8, S())
    return out


def transpose(X, R=None, Cc=None, lds=None):
    """dst[C][ldd] = X[R][C]^T with ldd = roundup(R, 64) (the K-tile of the MFMA GEMM) and zero fill of the pad columns."""
    R = X.shape[0] if R is None else R
    Cc = X.shape[1] if Cc is None else Cc
    lds = X.stride(0) if lds is None else lds
    ldd = rup(R, 64)
    out = torch.empty((Cc, ldd), dtype=X.dtype, device=X.device)
    call("ds2_transpose", dt(X), P(X), P(out), R, Cc, lds, ldd, S())
    return out


def cast_transpose_bf16(src, dst=None, ldd=0, dstT=None, lddT=0, perm=None, cout=None):
    """fp32 parameter matrix -> bf16 copy (window `dst`, row stride ldd) and/or bf16 transpose (window `dstT`, row stride
    lddT); `perm` = (c, f) re-orders the conv features of rnns.0, `cout` pads the columns with zeros."""
    R, Cc = src.shape
    pc, pf = perm if perm is not None else (0, 0)
    call("ds2_cast_transpose_bf16", PF(src), src.stride(0), R, Cc, pc, pf, Cc if cout is None else cout,
         P(dst) if dst is not None else None, ldd, P(dstT) if dstT is not None else None, lddT, S())


def small_weight_layouts(w1, w2, wfc, dtype):
    """(w1k, w2t, [w2d_even, w2d_odd], wfc_padded, wfcT) in kernel layout from the fp32 parameters -- one launch."""
    dev, Cc, H = w1.device, wfc.shape[0], wfc.shape[1]
    w1k = torch.empty((451, 32), dtype=torch.float32, device=dev)
    w2t = torch.empty((21, 11, 32, 32), dtype=dtype, device=dev)
    w2d0 = torch.empty((11, 11, 32, 32), dtype=dtype, device=dev)
    w2d1 = torch.empty((10, 11, 32, 32), dtype=dtype, device=dev)
    wp = torch.empty((32, H), dtype=dtype, device=dev)
    wpT = torch.empty((H, 32), dtype=dtype, device=dev)
    call("ds2_small_weight_layouts", dt(dtype), PF(w1), PF(w2), PF(wfc), Cc, H, P(w1k), P(w2t), P(w2d0), P(w2d1), P(wp), P(wpT), S())
    return w1k, w2t, [w2d0, w2d1], wp, wpT


def scale_by_(x, s):
    """x *= s[0] in place (x f32 contiguous, s a device scalar tensor)."""
    call("ds2_scale_by", P(x), PF(s.reshape(1)), x.numel(), S())
    return x


def add2(a, b):
    out = torch.empty_like(a)
    call("ds2_add2", dt(a), P(a), P(b), P(out), a.numel(), S())
    return out


class BnSaved:
    __slots__ = ("mean", "rstd", "scale", "shift")

    def __init__(self, Cc, device):
        buf = torch.empty((4, Cc), dtype=torch.float32, device=device)
        self.mean, self.rstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


def bn_fwd(X, mode, training, gamma, beta, rmean, rvar, nbt, R, Cc, ldx, Y, ldy, F=0, Tp=0, N=0, lens=None, eps=1e-5,
           momentum=0.1):
    sv = BnSaved(Cc, X.device)
    ws = torch.empty(2 * query("ds2_norm_partials", R) * Cc, dtype=torch.float32, device=X.device)
    call("ds2_bn_fwd", dt(X), mode, 1 if training else 0, P(X), P(Y), R, Cc, ldx, ldy, F, Tp, N, P(lens), PF(gamma), PF(beta),
         PF(rmean), PF(rvar), P(nbt) if training else C.c_void_p(0), float(eps), float(momentum), P(sv.mean), P(sv.rstd),
         P(sv.scale), P(sv.shift), P(ws), S())
    return sv


def bn_bwd(G, X, DX, mode, sv, R, Cc, ldg, ldx, lddx, F=0, Tp=0, N=0, lens=None):
    dgamma = torch.empty(Cc, dtype=torch.float32, device=X.device)
    dbeta = torch.empty(Cc, dtype=torch.float32, device=X.device)
    ws = torch.empty((2 * query("ds2_norm_partials", R) + 2) * Cc, dtype=torch.float32, device=X.device)
    call("ds2_bn_bwd", dt(X), mode, P(G), P(X), P(DX), R, Cc, ldg, ldx, lddx, F, Tp, N, P(lens), P(sv.mean), P(sv.rstd),
         P(sv.scale), P(sv.shift), P(dgamma), P(dbeta), P(ws), S())
    return dgamma, dbeta


# ---------------------------------------------------------------------------------------------------------------
def conv_rows(F0):
    """(F1, F2): frequency rows after conv1 / conv2 for F0 input bins (reference model.py:166-168): 161 -> (81, 41), 81 -> (41, 21)."""
    F1 = (F0 + 2 * 20 - 41) // 2 + 1
    return F1, (F1 + 2 * 10 - 21) // 2 + 1


def conv1_fwd(x, w1k, b1, lens, Tp, dtype):
    N, _, F0, T = x.shape
    y1 = torch.empty((N, conv_rows(F0)[0], Tp, 32), dtype=dtype, device=x.device)
    call("ds2_conv1_fwd", dt(dtype), PF(x), PF(w1k), PF(b1), P(lens), P(y1), N, F0, T, Tp, S())
    return y1


def conv1_wgrad(x, dy1, Tp):
    N, _, F0, T = x.shape
    dw = torch.empty((451, 32), dtype=torch.float32, device=x.device)
    ws = torch.empty(query("ds2_conv1_wgrad_ws_floats", N, F0, Tp), dtype=torch.float32, device=x.device)
    call("ds2_conv1_wgrad", dt(dy1), P(x), P(dy1), P(dw), N, F0, T, Tp, P(ws), S())
    return dw


def conv2_fwd(a1, w2t, b2, lens, F0=161):
    N, F1, Tp, _ = a1.shape
    assert F1 == conv_rows(F0)[0]
    y2 = torch.empty((N, conv_rows(F0)[1], Tp, 32), dtype=a1.dtype, device=a1.device)
    nws = query("ds2_conv2_fwd_ws_bytes", dt(a1), N, F0, Tp)
    ws = torch.empty(nws, dtype=torch.uint8, device=a1.device) if nws else None
    call("ds2_conv2_fwd", dt(a1), P(a1), P(w2t), PF(b2), P(lens), P(y2), N, F0, Tp, P(ws), S())
    return y2


def conv2_dgrad(dy2, w2d_even, w2d_odd, F0=161):
    N, F2, Tp, _ = dy2.shape
    assert F2 == conv_rows(F0)[1]
    da1 = torch.empty((N, conv_rows(F0)[0], Tp, 32), dtype=dy2.dtype, device=dy2.device)
    call("ds2_conv2_dgrad", dt(dy2), P(dy2), P(w2d_even), P(w2d_odd), P(da1), N, F0, Tp, S())
    return da1


def conv2_wgrad(dy2, a1, F0=161):
    N, _, Tp, _ = dy2.shape
    dw = torch.empty((231, 32, 32), dtype=torch.float32, device=dy2.device)
    ws = torch.empty(query("ds2_conv2_wgrad_ws_floats", N, Tp), dtype=torch.float32, device=dy2.device)
    call("ds2_conv2_wgrad", dt(dy2), P(dy2), P(a1), P(dw), N, F0, Tp, P(ws), S())
    return dw


# ---------------------------------------------------------------------------------------------------------------
_PERSIST_ERR = {}
LAST_PERSIST_WS = None   # the most recent persistent sweep's scratch (its tail holds the kernel's cycle counters)

# ---- per-launch options of the persistent sweeps (ds2_persist_opts) -----------------------------------------------------------
# What the host decides per launch: the routing A/B bits and the fault-injection spin budget (tests / tools only, through the
# persist_options() context, which restores them on exit whatever happens inside), and the START-UP budget -- how long the sweep's
# workgroups wait for all of them to become resident (a sweep needs every CU at once):
#   * single process: 300 ms.  Nothing legitimate holds compute units that long; fail fast, error code 2 names the cause.
#   * data parallel (torch.distributed initialised with > 1 rank, or dist.wrap_data_parallel told model.py): the process group's
#     time-out (torch's default for "nccl": 10 min).  An RCCL all-reduce on DDP's stream stays resident until its SLOWEST peer
#     arrives -- the reference's sampler hands ranks unequal batches (loader/data_loader.py:320-360), rank 0 saves checkpoints,
#     loaders stall -- and a sweep launched behind it must WAIT, exactly as a stock kernel would queue; if a peer is really gone
#     the collective's own watchdog ends the job, not the sweep.
#   * DS2_PERSIST_STARTUP_MS overrides both.
STARTUP_MS_SINGLE = 300
STARTUP_MS_DP_MIN, STARTUP_MS_DP_MAX = 30_000, 3_600_000
_OPTS = {"variant": 0, "spin_limit": 0, "startup_ms": None}
FORCE_DATA_PARALLEL_BUDGET = [False]      # model.PER_LAYER_NODES_FOR_DDP's twin: dist.wrap_data_parallel sets it for DS2_FORCE_DDP runs


class persist_options:
    """with ops.persist_options(variant=..., spin_limit=..., startup_ms=...): every persistent sweep launched inside (forward on this
    thread, backward on autograd's) carries these ds2_persist_opts; the previous values come back on exit."""

    def __init__(self, **kw):
        assert set(kw) <= set(_OPTS), kw
        self.kw = kw

    def __enter__(self):
        self.old = dict(_OPTS)
        _OPTS.update(self.kw)
        return self

    def __exit__(self, *exc):
        _OPTS.clear()
        _OPTS.update(self.old)
        return False


def data_parallel_ranks():
    """World size of the initialised default process group (1 if none)."""
    try:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            return int(_dist.get_world_size())
    except Exception:  # noqa: BLE001
        pass
    return 1


def persist_startup_ms():
    """The start-up budget the next sweep is launched with (see above)."""
    if _OPTS["startup_ms"] is not None:
        return int(_OPTS["startup_ms"])
    env = _os.environ.get("DS2_PERSIST_STARTUP_MS", "")
    if env:
        return max(1, int(env))
    if data_parallel_ranks() > 1 or FORCE_DATA_PARALLEL_BUDGET[0]:
        ms = 600_000
        try:
            import torch.distributed as _dist
            from torch.distributed.distributed_c10d import _get_default_group
            if _dist.is_initialized():
                to = getattr(_get_default_group(), "_timeout", None) or getattr(_dist, "default_pg_timeout", None)
                if to is not None:
                    ms = int(to.total_seconds() * 1000)
        except Exception:  # noqa: BLE001
            pass
        return min(max(ms, STARTUP_MS_DP_MIN), STARTUP_MS_DP_MAX)
    return STARTUP_MS_SINGLE


def _persist_opts():
    """(ctypes struct, byref) for one launch; the struct must stay alive until the call returns."""
    o = _lib.PersistOpts(int(_OPTS["variant"]), int(_OPTS["spin_limit"]), persist_startup_ms())
    return o


def persist_kind(dtype, kind, D, N, H):
    """Kernel family the persistent entries run for the problem on the current device under the current routing (0 none, 1 / 2 tuned
    H = 1024, 3 round-4 general, 4 round-2 general)."""
    return query("ds2_rnn_persist_kind", dt(dtype), CELLS[kind], D, N, H, int(_OPTS["variant"]))


def _persist_err(dev):
    """One int32 word per device that the persistent recurrent kernels raise when a workgroup times out."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _PERSIST_ERR.get(key)
    if t is None:
        t = _PERSIST_ERR[key] = torch.zeros(16, dtype=torch.int32, device=dev)
    return t


def _persist_error_message(code, key):
    """The sticky error word of the persistent sweeps: 1 = a workgroup gave up waiting for a peer's data in the middle of a sweep,
    2 = the start-up wait never completed, i.e. the sweep's workgroups (one whole CU each) were not all resident."""
    if code == 2:
        return ("a persistent recurrent sweep on device %d could not start: its workgroups never became co-resident within its start-up "
                "budget (%.1f s now; 0.3 s single-process, the process group's time-out under data parallelism, DS2_PERSIST_STARTUP_MS "
                "overrides) -- another kernel holds compute units (a second process on this GPU, e.g. a validation job or two ranks with "
                "the same LOCAL_RANK; or a long-running kernel on another stream).  The sweep gave up, its outputs were NaN-poisoned"
                % (key, persist_startup_ms() / 1000.0))
    return ("a persistent recurrent kernel timed out on device %d waiting for its peer workgroups in the middle of a sweep "
            "(are other kernels occupying CUs?); its outputs were NaN-poisoned" % key)


def check_persistent_kernels():
    """Synchronises and raises if any persistent recurrent kernel launched so far gave up waiting for its peers."""
    for key, t in _PERSIST_ERR.items():
        code = int(t[0].item())
        if code != 0:
            raise _lib.Ds2HipError(_persist_error_message(code, key))


_ERR_MIRROR = {}   # device index -> [pinned host int32 tensor, event of the copy in flight or None]


def poll_persistent_error(dev):
    """Surfaces a timed-out persistent sweep WITHOUT a host synchronisation (DeepSpeech.training_step calls this once per
    step): raises if the asynchronous copy of the device's error word enqueued by an earlier call has landed non-zero,
    then enqueues a fresh copy behind the work queued so far.  A stalled sweep therefore raises one step late at the
    latest; its outputs are NaN-poisoned meanwhile (and the CTC loss stays NaN), never silently zero."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _PERSIST_ERR.get(key)
    if t is None:
        return
    m = _ERR_MIRROR.get(key)
    if m is None:
        m = _ERR_MIRROR[key] = [torch.zeros(1, dtype=torch.int32).pin_memory(), None]
    if m[1] is not None:
        if not m[1].query():
            return                                   # the previous copy has not landed yet: look again next step
        m[1] = None
        if int(m[0][0]) != 0:
            raise _lib.Ds2HipError(_persist_error_message(int(m[0][0]), key))
    call("ds2_copy_words", P(t), C.c_void_p(m[0].data_ptr()), 1, S())      # in-stream kernel -> pinned host word (no SDMA transfer)
    ev = torch.cuda.Event()
    ev.record()
    m[1] = ev


class PinnedRing:
    """Host staging for the small per-step int32 tables (lengths + row list, CTC target table): ONE pinned allocation per ring, made at
    first use, cut into `depth` slots that are reused round-robin behind an event each.

    Why not ``torch.empty(..., pin_memory=True)`` per step: torch's pinned-block cache hands a block out again only after the copy
    that read it has finished.  A training loop's host thread runs several steps ahead of the device until the hardware queue is
    full, so during the first second of a run every step allocated FRESH pinned blocks (hipHostMalloc: device page-table edits under
    the running kernels), and the copies themselves went to the SDMA engines, where the transfers of LATER steps run under the
    kernels of earlier ones: a recurrent sweep (latency-bound, all 256 CUs polling L2) hit by one took 2.1-3.0 ms instead of 1.0-1.3
    -- 11 of the first 19 timed steps of a bench run, i.e. most of the window the driver times (tools/step_jitter.py,
    profiles/r05c_step_jitter.txt, r05d_step_jitter_env.txt).  The slot is therefore copied by an in-stream kernel (ds2_copy_words)."""

    def __init__(self, depth=32):
        self.depth = depth
        self._per_device = {}          # device index -> _RingState: slots, events and the copy kernel's stream belong to ONE device
        self._lock = __import__("threading").Lock()

    class _RingState:
        def __init__(self):
            self.cap, self.buf, self.events, self.i = 0, None, None, 0

    def stage(self, dev, parts):
        """parts: 1-D int32 numpy arrays -> their concatenation as ONE int32 device tensor (asynchronous copy on `dev`'s current
        stream; `dev` need not be the current device)."""
        n = int(sum(p.size for p in parts))
        dev = torch.device(dev)
        if dev.type != "cuda":
            return torch.from_numpy(np.concatenate([np.asarray(p, dtype=np.int32).reshape(-1) for p in parts]) if parts else np.zeros(0, np.int32))
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        with self._lock, torch.cuda.device(key):
            st = self._per_device.setdefault(key, PinnedRing._RingState())
            if st.buf is None or n > st.cap:
                if st.events is not None:
                    for e in st.events:               # the old buffer must outlive the copies that still read it
                        if e is not None:
                            e.synchronize()
                st.cap = max(1024, 1 << (max(n, 1) - 1).bit_length())
                st.buf = torch.empty(self.depth * st.cap, dtype=torch.int32, pin_memory=True)
                st.events = [None] * self.depth
                st.i = 0
            k = st.i
            st.i = (k + 1) % self.depth
            if st.events[k] is not None and not st.events[k].query():
                st.events[k].synchronize()            # the host is a whole ring ahead of the device: wait for the slot's last copy
            slot = st.buf[k * st.cap:k * st.cap + n]
            hv = slot.numpy()
            o = 0
            for p in parts:
                hv[o:o + p.size] = np.asarray(p).reshape(-1)
                o += p.size
            out = torch.empty(n, dtype=torch.int32, device=torch.device("cuda", key))
            if n:
                # a kernel that reads the (device-mapped) pinned slot, not hipMemcpyAsync: see ds2_copy_words for what SDMA transfers
                # of later steps cost the sweeps of earlier ones.  The kernel needs the slot's DEVICE address: equal to the host
                # address for torch's default pinned allocator (hipHostMalloc, unified addressing); with
                # PYTORCH_CUDA_ALLOC_CONF=pinned_use_cuda_host_register the two can differ, and the copy goes through torch instead
                dptr = _pinned_device_ptr(slot)
                if dptr is not None:
                    call("ds2_copy_words", C.c_void_p(dptr), P(out), n, S())
                else:
                    out.copy_(slot, non_blocking=True)
            if st.events[k] is None:
                st.events[k] = torch.cuda.Event()
            st.events[k].record()
            return out


_HIPRT = []


def _pinned_device_ptr(t):
    """Device address of a pinned host tensor (hipHostGetDevicePointer), or None if the runtime cannot give one."""
    if not _HIPRT:
        try:
            lib = C.CDLL("libamdhip64.so")
            lib.hipHostGetDevicePointer.restype = C.c_int
            lib.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
            _HIPRT.append(lib)
        except OSError:
            _HIPRT.append(None)
    lib = _HIPRT[0]
    if lib is None:
        return t.data_ptr()                           # no runtime library to ask: unified addressing is the default
    dp = C.c_void_p(0)
    rc = lib.hipHostGetDevicePointer(C.byref(dp), C.c_void_p(t.data_ptr()), 0)
    return dp.value if rc == 0 and dp.value else None


POISON_UNWRITTEN = False  # tests: the BPTT outputs start as NaN, so a consumer that reads a row the sweep was allowed to leave
                          # unwritten (rnn_bwd(pad_rows_unread=True)) shows up as NaN gradients
PERSIST_ENABLED = True   # tests flip this to run the per-time-step kernels on shapes the persistent kernels cover


_WARNED_CUS = set()
_WARNED_SHAPES = set()
PERSIST_CLIFF_MIN_H = 256      # below this width a time step is launch-bound either way: no warning


def use_persistent(kind, dtype, D, N, H):
    """True if a persistent sweep (one launch per layer and direction pair) takes this problem on the current device.  Never a
    silent cliff: when the answer is no because the DEVICE exposes fewer than 256 CUs although the shape is covered (partition /
    CU-mask modes) that is an error unless DS2_ALLOW_LAUNCH_PER_STEP=1 -- a CU-masked rank of a data-parallel job must not quietly
    become the straggler; when no persistent kernel is instantiated for the shape (the reference leaves hidden_size free,
    train_config.py:49) the launch-per-time-step kernels run, 5-8x slower per step, and a warning says so once per shape."""
    if not PERSIST_ENABLED:
        return False
    ok = bool(query("ds2_rnn_persist_supported", dt(dtype), CELLS[kind], D, N, H, int(_OPTS["variant"])))
    if ok:
        return True
    import os
    import warnings
    covered = bool(query("ds2_rnn_persist_shape_covered", dt(dtype), CELLS[kind], D, N, H))
    if covered:
        dev = torch.cuda.current_device()
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        if os.environ.get("DS2_ALLOW_LAUNCH_PER_STEP", "0") in ("", "0"):
            raise _lib.Ds2HipError("ds2hip: device %d exposes %d compute units (< 256): the persistent recurrent kernels need one "
                                   "workgroup per CU, all co-resident.  Set DS2_ALLOW_LAUNCH_PER_STEP=1 to run the launch-per-time-step "
                                   "kernels instead (several times slower)." % (dev, cus))
        if dev not in _WARNED_CUS:
            _WARNED_CUS.add(dev)
            warnings.warn("ds2hip: device %d exposes %d compute units (< 256): the persistent recurrent kernels are disabled, the "
                          "recurrent sweeps use one launch per time step" % (dev, cus))
    elif H >= PERSIST_CLIFF_MIN_H:
        key = (kind, str(dtype), D, N, H)
        if key not in _WARNED_SHAPES:
            _WARNED_SHAPES.add(key)
            warnings.warn("ds2hip: no persistent recurrent kernel is instantiated for %s %s hidden=%d, %d direction(s), batch %d: the "
                          "sweeps run one launch per time step (5-8x slower per step).  Persistent kernels exist for bf16 GRU / LSTM "
                          "with hidden in {384, 512, 640, 768, 800, 896, 1024, 1152, 1280, 1408, 1536} (LSTM: not 1408 / 1536) up to 32 "
                          "clips per group, bf16 hidden 1024 (any cell), and fp32 GRU / LSTM / RNN with hidden in {800, 1024, 1280}; the model class "
                          "zero-pads other hidden sizes up to the nearest of these when that is at most 1.5x as wide (DS2_PAD_HIDDEN)."
                          % (str(dtype).replace("torch.", ""), kind, H, D, N))
    return False


def rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp, h0=None, c0=None, save=True):
    """Returns (HseqExt [D][Tp+2][N][H] with zero guard slots, S, hn [D,N,H] f32, cn or None)."""
    dev, dtype = GI.device, GI.dtype
    hext = torch.empty((D, Tp + 2, N, H), dtype=dtype, device=dev)
    persistent = use_persistent(kind, dtype, D, N, H)
    if not persistent:                    # the persistent kernels zero the guard slots themselves
        hext[:, 0].zero_()
        hext[:, Tp + 1].zero_()
    ns = SAVED_PLANES[kind]
    Sv = torch.empty((D, Tp, N, max(ns, 1) * H) if ns else (1,), dtype=dtype, device=dev)
    hn = torch.empty((D, N, H), dtype=torch.float32, device=dev)
    cn = torch.empty((D, N, H), dtype=torch.float32, device=dev) if kind == "lstm" else None
    if persistent:
        ws = torch.empty(query("ds2_rnn_persist_ws_bytes", dt(dtype), CELLS[kind], D, N, H, int(_OPTS["variant"])), dtype=torch.uint8, device=dev)
        global LAST_PERSIST_WS
        LAST_PERSIST_WS = ws
        po = _persist_opts()
        with _sweep_timer("rnn_fwd_persistent", Tp):
            call("ds2_rnn_persist_fwd", dt(dtype), CELLS[kind], D, N, H, Tp, P(lens), P(GI), P(Whh), PF(bhh), PF(h0), PF(c0),
                 P(hext[:, 1]), (Tp + 2) * N * H, P(Sv), P(hn), P(cn), P(ws), P(_persist_err(dev)), C.byref(po), S())
        return hext, Sv, hn, cn
    state = torch.empty(query("ds2_rnn_state_bytes", D, N, H), dtype=torch.uint8, device=dev)
    with _sweep_timer("rnn_fwd", Tp):
        call("ds2_rnn_fwd", dt(dtype), CELLS[kind], D, N, H, Tp, P(lens), P(GI), P(Whh), PF(bhh), PF(h0), PF(c0),
             P(hext[:, 1]), (Tp + 2) * N * H, P(Sv), P(hn), P(cn), P(state), S())
    return hext, Sv, hn, cn


class RnnGrads:
    """What a BPTT sweep hands to the weight-gradient stage.  dGI [Tp*N][D*G*H]: gradient of the input projection (= of the
    hidden-side pre-activations too, except the GRU's n slot).  GRU only: either dGH [D][Tp][N][3H] (per-time-step kernels: the
    full hidden-side gradient [dr, dz, dQ]) or dQ [D][Tp][N][H] (persistent kernels: only the slot that differs from dGI).
    bacc [D][N][NB*H] f32 or None: per-sample sums over time of the gate-gradient planes (persistent kernels)."""
    __slots__ = ("dGI", "dGH", "dQ", "bacc", "dh0", "dc0")

    def __init__(self, dGI, dGH=None, dQ=None, bacc=None):
        self.dGI, self.dGH, self.dQ, self.bacc = dGI, dGH, dQ, bacc
        self.dh0 = self.dc0 = None          # d loss / d initial state (rnn_bwd(want_dstate=True))

    def tensors(self):
        return [t for t in (self.dGI, self.dGH, self.dQ, self.bacc) if t is not None]


def rnn_bwd(kind, dOut, WhhT, hext, Sv, lens, D, N, H, Tp, pad_rows_unread=False, h0=None, c0=None, want_dstate=False):
    """pad_rows_unread: the caller reads dGI / dQ only through a row list of the real frames (and takes the bias gradients from the
    sweep's per-sample sums): a persistent sweep that leaves the padding rows unwritten then does not zero them.
    h0 / c0 [D][N][H] f32: the initial state the forward was given (reference model.py:224-230); want_dstate: also return d loss /
    d h0 (and d c0) as RnnGrads.dh0 / .dc0.  Either routes the sweep to the launch-per-time-step BPTT (the persistent sweeps assume a
    zero initial state in backward; training through a given `hs` is the rare path)."""
    dev, dtype = dOut.device, dOut.dtype
    G = GATES[kind]
    dGI = torch.empty((Tp * N, D * G * H), dtype=dtype, device=dev)
    with_state = h0 is not None or c0 is not None or want_dstate
    if not with_state and use_persistent(kind, dtype, D, N, H):
        dQ = torch.empty((D, Tp, N, H), dtype=dtype, device=dev) if kind == "gru" else None
        if POISON_UNWRITTEN:
            dGI.fill_(float("nan"))
            if dQ is not None:
                dQ.fill_(float("nan"))
        bacc = torch.empty((D, N, (4 if kind == "gru" else G) * H), dtype=torch.float32, device=dev)
        ws = torch.empty(query("ds2_rnn_persist_ws_bytes", dt(dtype), CELLS[kind], D, N, H, int(_OPTS["variant"])), dtype=torch.uint8, device=dev)
        global LAST_PERSIST_WS
        LAST_PERSIST_WS = ws
        po = _persist_opts()
        with _sweep_timer("rnn_bwd_persistent", Tp):
            call("ds2_rnn_persist_bwd", dt(dtype), CELLS[kind], D, N, H, Tp, P(lens), P(dOut), P(WhhT), P(hext[:, 1]), (Tp + 2) * N * H,
                 P(Sv), P(dGI), P(dQ), P(bacc), 1 if pad_rows_unread else 0, P(ws), P(_persist_err(dev)), C.byref(po), S())
        return RnnGrads(dGI, dQ=dQ, bacc=bacc)
    dGH = torch.empty((D, Tp, N, G * H), dtype=dtype, device=dev) if kind == "gru" else None
    state = torch.empty(query("ds2_rnn_state_bytes", D, N, H), dtype=torch.uint8, device=dev)
    dh0 = torch.empty((D, N, H), dtype=torch.float32, device=dev) if want_dstate else None
    dc0 = torch.empty((D, N, H), dtype=torch.float32, device=dev) if (want_dstate and kind == "lstm") else None
    with _sweep_timer("rnn_bwd", Tp):
        call("ds2_rnn_bwd", dt(dtype), CELLS[kind], D, N, H, Tp, P(lens), P(dOut), P(WhhT), P(hext[:, 1]), (Tp + 2) * N * H,
             P(Sv), P(dGI), P(dGH), PF(h0), PF(c0), P(dh0), P(dc0), P(state), S())
    rg = RnnGrads(dGI, dGH=dGH)
    rg.dh0, rg.dc0 = dh0, dc0
    return rg


# ---------------------------------------------------------------------------------------------------------------
def rnn_bias_grads(kind, bacc, D, N, H):
    """(bias_ih.grad [D*G*H], bias_hh.grad [D][G*H]) from the BPTT sweep's per-sample sums bacc [D][N][NB*H] -- one launch."""
    G = GATES[kind]
    dbih = torch.empty(D * G * H, dtype=torch.float32, device=bacc.device)
    dbhh = torch.empty((D, G * H), dtype=torch.float32, device=bacc.device)
    call("ds2_rnn_bias_grads", CELLS[kind], D, N, H, PF(bacc), P(dbih), P(dbhh), S())
    return dbih, dbhh


def lookahead_fwd(x, w, Tp, N, H, save=True):
    y = torch.empty_like(x)
    pre = torch.empty_like(x) if save else None
    call("ds2_lookahead_fwd", dt(x), P(x), PF(w), P(y), P(pre), Tp, N, H, w.shape[1], S())
    return y, pre


def lookahead_bwd(x, w, pre, dy, Tp, N, H):
    ctx = w.shape[1]
    dx = torch.empty_like(x)
    dw = torch.empty((H, ctx), dtype=torch.float32, device=x.device)
    ws = torch.empty(query("ds2_lookahead_ws_floats", Tp, N, H, ctx), dtype=torch.float32, device=x.device)
    call("ds2_lookahead_bwd", dt(x), P(x), P(w), P(pre), P(dy), P(dx), P(dw), Tp, N, H, ctx, P(ws), S())
    return dx, dw


def softmax_rows(logits, Cc):
    rows = logits.shape[0]
    out = torch.empty((rows, Cc), dtype=torch.float32, device=logits.device)
    call("ds2_softmax_rows", P(logits), P(out), rows, Cc, logits.stride(0), Cc, S())
    return out


def greedy_decode(scores, sizes, blank):
    """scores: (N, T', C) f32 CUDA tensor (any strides with a contiguous class dimension); sizes: [N] int tensor.
    Returns host lists (tokens per sample, frame offsets per sample) -- reference decoder.py:164-181."""
    if scores.dtype != torch.float32:
        scores = scores.float()
    if scores.stride(2) != 1:
        scores = scores.contiguous()
    N, T, Cc = scores.shape
    dev = scores.device
    sz = sizes.to(dev, torch.int32) if sizes is not None else None
    buf = torch.empty((2, N, T), dtype=torch.int32, device=dev)
    counts = torch.empty(N, dtype=torch.int32, device=dev)
    call("ds2_greedy_decode", P(scores), scores.stride(0), scores.stride(1), N, T, Cc, P(sz), blank, P(buf[0]), P(buf[1]), P(counts),
         S())
    cnt = counts.cpu()
    width = int(cnt.max().item()) if N > 0 else 0
    host = buf[:, :, :max(width, 1)].cpu()           # only the surviving labels + offsets travel
    toks = [host[0, i, :int(cnt[i])].tolist() for i in range(N)]
    offs = [host[1, i, :int(cnt[i])] for i in range(N)]
    return toks, offs


CTC_RECURSION = 0    # tests / A-B tools: 0 = pair tiles (default), 1 = always the four-wave recursion kernel, 2 = the one-wave kernel up to 255 labels, 3 = rounds 3-5


def ctc_loss_grad(logits, targets_i32, target_offsets, input_lengths, target_lengths, Tp, N, Cc, blank, max_target_len,
                  ldg=None, recursion=None):
    """logits [Tp*N][ld] f32.  Returns (loss_sum [1], nll [N], dlogits [Tp*N][ldg] f32 with unit upstream gradient)."""
    dev = logits.device
    ldg = logits.shape[1] if ldg is None else ldg
    nll = torch.empty(N, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dlogits = torch.empty((Tp * N, ldg), dtype=torch.float32, device=dev)
    ws = torch.empty(query("ds2_ctc_ws_floats", Tp, N, Cc, max_target_len), dtype=torch.float32, device=dev)
    call("ds2_ctc_loss_grad", P(logits), logits.stride(0), P(targets_i32), P(target_offsets), P(input_lengths),
         P(target_lengths), Tp, N, Cc, blank, max_target_len, 1.0, P(nll), P(loss), P(dlogits), ldg, P(ws),
         int(CTC_RECURSION if recursion is None else recursion), S())
    return loss, nll, dlogits
