"""GPU parity tests, kernel by kernel: every C-ABI entry of libds2hip.so against the numpy oracle on seeded inputs.
fp32 storage: tight tolerances (the 1e-3 north-star bar with margin).  bf16 storage: the oracle is fed bf16-rounded
inputs; tolerance is bf16 rounding of the stored outputs (2^-8 relative) plus accumulation noise."""
import numpy as np
import pytest
import torch

from oracle import ds2_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from deepspeech.pytorch_amd import ops as _ops
    return _ops


def cu(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype).contiguous()


def rnd(a, dtype):
    """what the device sees after storing `a` in `dtype` (numpy float64 in, float64 out)"""
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(torch.float64).numpy()


def np64(t):
    return t.detach().to(torch.float64).cpu().numpy()


def relerr(got, ref):
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2}


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(200, 77, 96), (128, 128, 64), (257, 300, 1312), (29, 64, 40), (1000, 29, 64),
                                   (384, 256, 192), (130, 70, 1344), (1500, 1312, 512)])
def test_gemm_nt(dtype, M, N, K):
    rs = np.random.RandomState(M + N + K)
    A, B, bias = rs.standard_normal((M, K)), rs.standard_normal((N, K)), rs.standard_normal(N)
    ref = rnd(A, dtype) @ rnd(B, dtype).T + bias
    o = ops()
    got = o.gemm_nt(cu(A, dtype), cu(B, dtype), bias=cu(bias), out_dtype=torch.float32)
    assert relerr(np64(got), ref) < (1e-5 if dtype == torch.float32 else 1e-4)   # fp32 accumulate, fp32 out
    got_t = o.gemm_nt(cu(A, dtype), cu(B, dtype), bias=cu(bias))
    assert relerr(np64(got_t), ref) < TOL[dtype]
    # the low-register variant used for GEMMs that share the CUs with a persistent recurrent sweep
    got_c = o.gemm_nt(cu(A, dtype), cu(B, dtype), bias=cu(bias), coresident=True)
    assert relerr(np64(got_c), ref) < TOL[dtype]


@pytest.mark.parametrize("M,N,K", [(808, 800, 4800), (300, 1024, 6144), (128, 128, 2048)])
def test_gemm_nt_fp32_k_slices_are_exact_and_repeatable(M, N, K):
    """Round 6: an exact-fp32 product with few output tiles and a long contraction (a layer's dX in the 1e-3 parity mode: 49 tiles
    of K = 4 800 on 256 CUs) runs its K-slices as the kernel's batch and adds them in index order (ds2_sum_slices) -- split-K
    without atomics: fp32-exact against float64, and bit-identical from launch to launch."""
    rs = np.random.RandomState(M + N + K)
    A, B = rs.standard_normal((M, K)), rs.standard_normal((N, K))
    o = ops()
    ks = o.fp32_ksplit(torch.float32, M, N, K)
    assert ks > 1
    Ad, Bd = cu(A, torch.float32), cu(B, torch.float32)
    got = o.gemm_nt(Ad, Bd)
    ref = rnd(A, torch.float32) @ rnd(B, torch.float32).T
    assert relerr(np64(got), ref) < 1e-5
    for _ in range(3):
        assert torch.equal(o.gemm_nt(Ad, Bd), got)
    # against the single-pass kernel: the same products, another (fixed) summation order
    old = o.FP32_KSPLIT
    o.FP32_KSPLIT = False
    try:
        one = o.gemm_nt(Ad, Bd)
    finally:
        o.FP32_KSPLIT = old
    assert relerr(np64(one), np64(got)) < 1e-5


@pytest.mark.parametrize("coresident", [False, True])
@pytest.mark.parametrize("M,ms,N,K,sk", [(3072, 2048, 1024, 1536, 1), (192, 128, 96, 640, 1), (160, 96, 64, 4096, 4)])
def test_gemm_nt_rows2(coresident, M, ms, N, K, sk):
    """A operand given as two row blocks in separate buffers (the GRU's [dr, dz | dQ] hidden-side gradient)."""
    rs = np.random.RandomState(M + N + K)
    A, B = rs.standard_normal((M, K)), rs.standard_normal((N, K))
    ref = rnd(A, torch.bfloat16) @ rnd(B, torch.bfloat16).T
    o = ops()
    Ad = cu(A, torch.bfloat16)
    a1, a2 = Ad[:ms].contiguous(), Ad[ms:].contiguous()
    got = o.gemm_nt_rows2(a1, a2, ms, cu(B, torch.bfloat16), M, N, K, K, K, splitk=sk, coresident=coresident)
    assert relerr(np64(got), ref) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_nt_transposed_operands_and_splitk(dtype):
    """A asymmetric, non-square: catches row/col swaps of the MFMA C layout (cdna guide: always test asymmetric)."""
    rs = np.random.RandomState(3)
    M, N, K = 96, 160, 1000
    A, B = rs.standard_normal((M, K)), rs.standard_normal((N, K)) * np.arange(1, N + 1)[:, None] / N
    ref = rnd(A, dtype) @ rnd(B, dtype).T
    o = ops()
    got = o.gemm_nt(cu(A, dtype), cu(B, dtype), out_dtype=torch.float32, splitk=4)
    assert relerr(np64(got), ref) < 1e-4
    # batched
    Ab, Bb = rs.standard_normal((3, 40, 64)), rs.standard_normal((3, 50, 64))
    refb = np.einsum("bmk,bnk->bmn", rnd(Ab, dtype), rnd(Bb, dtype))
    gotb = o.gemm_nt(cu(Ab, dtype), cu(Bb, dtype), out_dtype=torch.float32, batch=3, sA=40 * 64, sB=50 * 64)
    assert relerr(np64(gotb), refb) < 1e-4


def test_cast_transpose_bf16_windows_and_feature_permutation():
    """The weight re-layout kernel: bf16 copy and transpose written into windows of direction-stacked operands, bit-exact
    against torch's round-to-nearest-even cast; rnns.0's column permutation (reference c*41+f -> internal f*32+c) + pad."""
    o = ops()
    g = torch.Generator().manual_seed(12)
    for R, Cc, perm, cout in [(96, 64, None, None), (2400, 800, None, None), (48, 1312, (32, 41), 1344), (16, 16, None, None),
                              (3072, 1024, None, None)]:
        W = [torch.randn((R, Cc), generator=g).cuda() for _ in range(2)]
        Co = cout or Cc
        dst = torch.full((2 * R, Co), 7.0, dtype=torch.bfloat16, device="cuda")
        dstT = torch.full((Co, 2 * R), 7.0, dtype=torch.bfloat16, device="cuda")
        for d in range(2):
            o.cast_transpose_bf16(W[d], dst[d * R:], Co, dstT[:, d * R:], 2 * R, perm=perm, cout=Co)
        ref = torch.cat(W, 0)
        if perm is not None:
            full = ref.new_zeros((2 * R, Co))
            full[:, :Cc] = ref.reshape(2 * R, perm[0], perm[1]).permute(0, 2, 1).reshape(2 * R, Cc)
            ref = full
        ref = ref.to(torch.bfloat16)
        assert torch.equal(dst, ref)
        assert torch.equal(dstT, ref.t())
        only = torch.empty_like(dst)
        for d in range(2):
            o.cast_transpose_bf16(W[d], only[d * R:], Co, None, 0, perm=perm, cout=Co)      # forward-only: no transpose
        assert torch.equal(only, ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose_colsum_add(dtype):
    rs = np.random.RandomState(4)
    o = ops()
    for R, Cc in [(93, 96), (64, 64), (300, 1312), (7, 8)]:
        X = rs.standard_normal((R, Cc))
        t = o.transpose(cu(X, dtype))
        assert t.shape == (Cc, (R + 63) // 64 * 64)     # padded to the K-tile of the MFMA GEMM, zero filled
        got = np64(t)
        assert np.array_equal(got[:, :R], rnd(X, dtype).T)
        assert np.all(got[:, R:] == 0)
        cs = o.colsum(cu(X, dtype))
        assert relerr(np64(cs), rnd(X, dtype).sum(0)) < 1e-5
    a, b = rs.standard_normal(4096), rs.standard_normal(4096)
    s = o.add2(cu(a, dtype), cu(b, dtype))
    assert relerr(np64(s), rnd(a, dtype) + rnd(b, dtype)) < TOL[dtype]


# ---------------------------------------------------------------------------------------------------------------
def nftc(a):   # (N,C,F,T) -> [N][F][T][C]
    return np.ascontiguousarray(a.transpose(0, 2, 3, 1))


def nchw(a):
    return np.ascontiguousarray(a.transpose(0, 3, 1, 2))


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_sequence(dtype):
    rs = np.random.RandomState(5)
    R, Cc = 93, 64
    X = rs.standard_normal((R, Cc)) * 2 + 0.5
    X[80:] = 0   # zero pad rows take part in the statistics
    gamma, beta = rs.uniform(0.5, 1.5, Cc), rs.uniform(-0.2, 0.2, Cc)
    rm, rv = rs.uniform(-0.1, 0.1, Cc), rs.uniform(0.5, 1.5, Cc)
    Xr = rnd(X, dtype)
    yref, cache = O.bn_train_fwd(Xr, gamma, beta, (0,))
    rm2, rv2 = O.bn_running_update(rm, rv, cache)
    o = ops()
    Xd, Y = cu(X, dtype), torch.empty((R, Cc), dtype=dtype, device=DEV)
    rmd, rvd, nbt = cu(rm), cu(rv), torch.zeros(1, dtype=torch.int64, device=DEV)
    sv = o.bn_fwd(Xd, 0, True, cu(gamma), cu(beta), rmd, rvd, nbt, R, Cc, Cc, Y, Cc)
    assert relerr(np64(Y), yref) < TOL[dtype]
    assert relerr(np64(rmd), rm2) < 1e-5 and relerr(np64(rvd), rv2) < 1e-5 and int(nbt.item()) == 1
    G = rs.standard_normal((R, Cc))
    dxref, dgref, dbref = O.bn_train_bwd(rnd(G, dtype), gamma, cache)
    DX = torch.empty((R, Cc), dtype=dtype, device=DEV)
    dg, db = o.bn_bwd(cu(G, dtype), Xd, DX, 0, sv, R, Cc, Cc, Cc, Cc)
    assert relerr(np64(DX), dxref) < TOL[dtype]
    assert relerr(np64(dg), dgref) < 1e-4 and relerr(np64(db), dbref) < 1e-4
    # eval mode uses the running statistics
    Y2 = torch.empty((R, Cc), dtype=dtype, device=DEV)
    o.bn_fwd(Xd, 0, False, cu(gamma), cu(beta), cu(rm), cu(rv), None, R, Cc, Cc, Y2, Cc)
    assert relerr(np64(Y2), O.bn_eval_fwd(Xr, gamma, beta, rm, rv, 1)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("seq_out", [False, True])
def test_bn_conv_modes(dtype, seq_out):
    """BatchNorm2d + Hardtanh + time mask on an NFTC activation (modes 1/2), forward and backward."""
    rs = np.random.RandomState(6)
    N, F, Tp, Cc = 3, 5, 19, 32
    lens = np.array([19, 12, 7], dtype=np.int32)
    x = rs.standard_normal((N, Cc, F, Tp)) * 3
    m = O.time_mask(x.shape, lens)
    x[m] = 0
    gamma, beta = rs.uniform(3.0, 9.0, Cc), rs.uniform(-0.2, 4.0, Cc)   # large gain: both clamp sides are hit
    xr = nchw(rnd(nftc(x), dtype))
    z, cache = O.bn_train_fwd(xr, gamma, beta, (0, 2, 3))
    z[m] = 0
    a = O.hardtanh_fwd(z)
    a[m] = 0
    assert (z >= 20).any() and (z <= 0).any()
    o = ops()
    R = N * F * Tp
    Xd = cu(nftc(x), dtype).view(R, Cc)
    lens_d = torch.from_numpy(lens).to(DEV)
    rmd, rvd, nbt = cu(np.zeros(Cc)), cu(np.ones(Cc)), torch.zeros(1, dtype=torch.int64, device=DEV)
    if seq_out:
        Y = torch.full((Tp * N, F * Cc), 7.0, dtype=dtype, device=DEV)
        sv = o.bn_fwd(Xd, 2, True, cu(gamma), cu(beta), rmd, rvd, nbt, R, Cc, Cc, Y, F * Cc, F=F, Tp=Tp, N=N, lens=lens_d)
        ref = a.transpose(3, 0, 2, 1).reshape(Tp * N, F * Cc)   # [(t,n)][f*32+c]
    else:
        Y = torch.full((R, Cc), 7.0, dtype=dtype, device=DEV)
        sv = o.bn_fwd(Xd, 1, True, cu(gamma), cu(beta), rmd, rvd, nbt, R, Cc, Cc, Y, Cc, F=F, Tp=Tp, N=N, lens=lens_d)
        ref = nftc(a).reshape(R, Cc)
    assert np.abs(np64(Y) - ref).max() < TOL[dtype] * 20
    # backward
    g = rs.standard_normal(x.shape)
    d = rnd(g, dtype).copy()
    d[m] = 0
    d = O.hardtanh_bwd(z, d)
    d[m] = 0
    dx, dgam, dbet = O.bn_train_bwd(d, gamma, cache)
    dx[m] = 0
    DX = torch.empty((R, Cc), dtype=dtype, device=DEV)
    if seq_out:
        Gd = cu(g.transpose(3, 0, 2, 1).reshape(Tp * N, F * Cc), dtype)
        dg, db = o.bn_bwd(Gd, Xd, DX, 2, sv, R, Cc, F * Cc, Cc, Cc, F=F, Tp=Tp, N=N, lens=lens_d)
    else:
        Gd = cu(nftc(g).reshape(R, Cc), dtype)
        dg, db = o.bn_bwd(Gd, Xd, DX, 1, sv, R, Cc, Cc, Cc, Cc, F=F, Tp=Tp, N=N, lens=lens_d)
    # elements whose BN output sits within rounding distance of a clamp edge may legitimately flip in bf16
    tol = TOL[dtype] * (1 if dtype == torch.float32 else 8)
    assert relerr(np64(DX), nftc(dx).reshape(R, Cc)) < tol
    assert relerr(np64(dg), dgam) < tol and relerr(np64(db), dbet) < tol


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,T,F0", [(2, 37, 161), (3, 150, 161), (2, 523, 161),          # 523 frames: three 128-frame blocks, ragged tail
                                    (2, 70, 81), (3, 41, 129), (1, 33, 257), (2, 29, 41)])  # other SpectConfig geometries (general kernels)
def test_conv1_fwd_and_wgrad(dtype, N, T, F0):
    rs = np.random.RandomState(7 + T)
    x = rs.standard_normal((N, 1, F0, T))
    w, b = rs.uniform(-0.05, 0.05, (32, 1, 41, 11)), rs.uniform(-0.1, 0.1, 32)
    lens_in = np.array([T, max(T - 9, 1), max(T // 2, 1)][:N])
    lens = O.seq_lens(lens_in)
    Tp = int(O.seq_lens(np.array([T]))[0])
    # bf16 storage: the layer runs on the matrix pipes with bf16 input and weight and fp32 sums (what torch.autocast does with
    # it); the oracle gets the same rounded operands, so the bar stays that of the output rounding alone
    xq = rnd(x.astype(np.float32).astype(np.float64), dtype)
    y = O.conv2d_fwd(xq, rnd(w.astype(np.float32).astype(np.float64), dtype),
                     b.astype(np.float32).astype(np.float64), (2, 2), (20, 5))
    m = O.time_mask(y.shape, lens)
    y[m] = 0
    o = ops()
    xd, lens_d = cu(x), torch.from_numpy(lens).to(DEV)
    w1k = cu(w.reshape(32, 451).T)
    y1 = o.conv1_fwd(xd, w1k, cu(b), lens_d, Tp, dtype)
    assert relerr(np64(y1), nftc(y)) < TOL[dtype]
    # weight gradient: dy masked like the real pipeline
    dy = rs.standard_normal(y.shape)
    dy[m] = 0
    dyr = nchw(rnd(nftc(dy), dtype))
    # the general kernels (geometries other than 161 bins) multiply the fp32 input itself; the matrix-pipe kernel its bf16 rounding
    xw = xq if F0 == 161 else x.astype(np.float32).astype(np.float64)
    _, dw, _ = O.conv2d_bwd(xw, w, dyr, (2, 2), (20, 5), need_dx=False)
    got = o.conv1_wgrad(xd, cu(nftc(dy), dtype), Tp)
    assert relerr(np64(got), dw.reshape(32, 451).T) < 1e-4


def test_conv1_fwd_persistent_blocks_equal_single_block_runs():
    """The bf16 conv1 forward is a persistent kernel: a workgroup walks several (sample, frame block, row block) tiles through one
    LDS patch.  A batch of 20 long clips is 1 080 tiles on 256 workgroups (4-5 tiles each); every sample run ALONE is 54 tiles (one
    per workgroup).  Same arithmetic per tile, so the two must agree bit for bit (the kernel tests above are all single-tile)."""
    rs = np.random.RandomState(77)
    N, T = 20, 1501
    x = torch.from_numpy(rs.standard_normal((N, 1, 161, T)).astype(np.float32)).to(DEV)
    w1k = cu(rs.uniform(-0.05, 0.05, (451, 32)))
    b = cu(rs.uniform(-0.1, 0.1, 32))
    lens_in = np.array([T - 37 * i for i in range(N)])
    lens = O.seq_lens(lens_in)
    Tp = int(O.seq_lens(np.array([T]))[0])
    o = ops()
    full = o.conv1_fwd(x, w1k, b, torch.from_numpy(lens).to(DEV), Tp, torch.bfloat16)
    for n in (0, 7, 19):
        one = o.conv1_fwd(x[n:n + 1].contiguous(), w1k, b, torch.from_numpy(lens[n:n + 1]).to(DEV), Tp, torch.bfloat16)
        assert torch.equal(full[n:n + 1], one), n
    assert float(full.float().abs().sum()) > 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,Tp,F0", [(2, 19, 161), (3, 75, 161), (2, 300, 161),
                                     (2, 40, 81), (3, 21, 129), (1, 17, 257), (2, 15, 41)])   # F1 = 41 / 65 / 129 / 21 (odd and even)
def test_conv2_fwd_dgrad_wgrad(dtype, N, Tp, F0):
    rs = np.random.RandomState(8 + Tp)
    a1 = np.maximum(rs.standard_normal((N, 32, ops().conv_rows(F0)[0], Tp)), 0)
    lens = np.array([Tp, max(Tp - 5, 1), max(Tp // 3, 1)][:N], dtype=np.int32)
    a1[O.time_mask(a1.shape, lens)] = 0
    w, b = rs.uniform(-0.02, 0.02, (32, 32, 21, 11)), rs.uniform(-0.1, 0.1, 32)
    a1r, wr = nchw(rnd(nftc(a1), dtype)), rnd(w, dtype)
    y = O.conv2d_fwd(a1r, wr, b.astype(np.float32).astype(np.float64), (2, 1), (10, 5))
    m = O.time_mask(y.shape, lens)
    y[m] = 0
    o = ops()
    lens_d = torch.from_numpy(lens).to(DEV)
    a1d = cu(nftc(a1), dtype)
    w2t = cu(w.transpose(2, 3, 0, 1), dtype)            # [kf][kt][co][ci]
    y2 = o.conv2_fwd(a1d, w2t, cu(b), lens_d, F0)
    assert tuple(y2.shape) == (N, o.conv_rows(F0)[1], Tp, 32)
    assert relerr(np64(y2), nftc(y)) < TOL[dtype]
    # backward
    dy = rs.standard_normal(y.shape)
    dy[m] = 0
    dyr = nchw(rnd(nftc(dy), dtype))
    dx, dw, _ = O.conv2d_bwd(a1r, wr, dyr, (2, 1), (10, 5))
    wt = torch.from_numpy(w)
    w2d = [wt[:, :, q::2, :].flip(2, 3).permute(2, 3, 1, 0).contiguous().to(DEV).to(dtype) for q in (0, 1)]
    dyd = cu(nftc(dy), dtype)
    da1 = o.conv2_dgrad(dyd, w2d[0], w2d[1], F0)
    assert relerr(np64(da1), nftc(dx)) < TOL[dtype]
    # wgrad contracts bf16-exact products in fp32: compare against the oracle on the rounded operands
    dw2 = o.conv2_wgrad(dyd, a1d, F0)
    assert relerr(np64(dw2).reshape(21, 11, 32, 32), dw.transpose(2, 3, 0, 1)) < 1e-4


@pytest.mark.parametrize("M,N,K", [(808, 4800, 1344), (808, 800, 4800), (4800, 1344, 808), (300, 256, 328)])
def test_gemm_split3_is_fp32_class(M, N, K):
    """fp32-mode GEMMs as ONE bf16 GEMM over three K-segments of split operands (ops.split3: a_hi b_hi + a_hi b_lo + a_lo b_hi):
    2e-5 of the product's scale against float64 -- two orders inside the 1e-3 parity bar of the fp32 mode, where a plain bf16
    product sits at 4e-3 -- and the segments themselves bit-exact (hi = bf16(x), lo = bf16(x - hi), zero pad)."""
    o = ops()
    rs = np.random.RandomState(M + N + K)
    A = rs.standard_normal((M, K)) * np.exp(rs.uniform(-3, 3, (M, 1)))       # rows of very different scale, like gradients
    B = rs.standard_normal((N, K))
    Ad, Bd = cu(A), cu(B)
    A3, B3 = o.split3(Ad, 0), o.split3(Bd, 1)
    Kp = (K + 63) // 64 * 64
    assert A3.shape == (M, 3 * Kp) and B3.shape == (N, 3 * Kp)
    a32 = torch.from_numpy(A.astype(np.float32))
    hi = a32.to(torch.bfloat16)
    lo = (a32 - hi.float()).to(torch.bfloat16)
    got = A3.cpu()
    assert torch.equal(got[:, :K], hi) and torch.equal(got[:, Kp:Kp + K], hi) and torch.equal(got[:, 2 * Kp:2 * Kp + K], lo)
    assert not got[:, K:Kp].any() and not got[:, 2 * Kp + K:].any()
    b32 = torch.from_numpy(B.astype(np.float32))
    bhi = b32.to(torch.bfloat16)
    blo = (b32 - bhi.float()).to(torch.bfloat16)
    gotb = B3.cpu()
    assert torch.equal(gotb[:, :K], bhi) and torch.equal(gotb[:, Kp:Kp + K], blo) and torch.equal(gotb[:, 2 * Kp:2 * Kp + K], bhi)
    C = np64(o.gemm_nt(A3, B3, out_dtype=torch.float32))
    ref = A.astype(np.float32).astype(np.float64) @ B.astype(np.float32).astype(np.float64).T
    scale = np.sqrt((A ** 2).sum(1))[:, None] * np.sqrt((B ** 2).sum(1))[None, :]          # |a| |b| per entry
    assert (np.abs(C - ref) / scale).max() < 2e-5, (np.abs(C - ref) / scale).max()
    plain = np64(o.gemm_nt(Ad.to(torch.bfloat16).contiguous(), Bd.to(torch.bfloat16).contiguous(), out_dtype=torch.float32)) \
        if K % 64 == 0 else None
    if plain is not None:
        assert (np.abs(plain - ref) / scale).max() > 10 * (np.abs(C - ref) / scale).max()


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
@pytest.mark.parametrize("D,N,H,Tp", [(2, 3, 32, 13), (1, 20, 48, 9), (2, 37, 32, 7)])
def test_rnn_sweeps(dtype, kind, D, N, H, Tp):
    _rnn_sweep_case(dtype, kind, D, N, H, Tp, 0.3)


@pytest.mark.parametrize("kind,D,N,H,Tp", [("gru", 2, 11, 1024, 13), ("gru", 1, 32, 1024, 8), ("gru", 2, 32, 1024, 9),
                                           ("lstm", 2, 13, 1024, 7), ("lstm", 1, 5, 1024, 6), ("rnn", 2, 9, 1024, 11),
                                           ("gru", 2, 1, 1024, 7), ("gru", 2, 64, 1024, 5), ("lstm", 2, 64, 1024, 4)])
def test_rnn_persistent_sweeps(kind, D, N, H, Tp):
    """The persistent recurrent kernels (one launch per sweep, W_hh resident in registers, tagged-granule exchange) against
    the oracle, and against the per-time-step kernels on the same inputs."""
    o = ops()
    assert o.use_persistent(kind, torch.bfloat16, D, N, H), "persistent path not selected on this device"
    res_p = _rnn_sweep_case(torch.bfloat16, kind, D, N, H, Tp, 1.0 / np.sqrt(H))
    o.check_persistent_kernels()
    o.PERSIST_ENABLED = False
    try:
        res_s = _rnn_sweep_case(torch.bfloat16, kind, D, N, H, Tp, 1.0 / np.sqrt(H))
    finally:
        o.PERSIST_ENABLED = True
    for a, b in zip(res_p, res_s):   # same bf16 operands; only the fp32 summation order differs
        assert np.abs(a - b).max() <= 2e-2 * max(1.0, np.abs(b).max()), np.abs(a - b).max()


@pytest.mark.parametrize("kind,D,N,H,Tp", [("gru", 2, 32, 1024, 9), ("gru", 2, 11, 1024, 13), ("lstm", 1, 5, 1024, 6), ("rnn", 2, 9, 1024, 11),
                                           ("gru", 2, 1, 1024, 7)])
def test_rnn_persistent_sparse_and_dense_forms_agree(kind, D, N, H, Tp):
    """Groups of <= 8 clips (round 5): the sweeps' products run on the structured-sparse instruction -- a clip's k = 0, 1 (mod 4)
    elements on tile row s, its k = 2, 3 (mod 4) elements on row s + 8 -- instead of leaving half of the dense tile to padding.
    Both forms multiply the same bf16 operands and are held to the oracle by _rnn_sweep_case; against each other only the fp32
    summation order may differ."""
    o = ops()
    from deepspeech.pytorch_amd._lib import query
    assert o.use_persistent(kind, torch.bfloat16, D, N, H)
    res_sp = _rnn_sweep_case(torch.bfloat16, kind, D, N, H, Tp, 1.0 / np.sqrt(H))
    o.check_persistent_kernels()
    with o.persist_options(variant=32):                    # bit 5: dense products (round 4's form)
        res_de = _rnn_sweep_case(torch.bfloat16, kind, D, N, H, Tp, 1.0 / np.sqrt(H))
        o.check_persistent_kernels()
    for a, b in zip(res_sp, res_de):
        assert np.abs(a - b).max() <= 2e-2 * max(1.0, np.abs(b).max()), np.abs(a - b).max()


GENERAL_PERSISTENT_CASES = [
    # fp32 storage (the 1e-3 parity mode): BASELINE config 2's width (H = 800; 25 / 150 k-steps: ragged K split) and 1024
    (torch.float32, "gru", 2, 8, 800, 9), (torch.float32, "lstm", 1, 5, 800, 7), (torch.float32, "gru", 2, 32, 1024, 6),
    (torch.float32, "lstm", 2, 3, 1024, 5), (torch.float32, "gru", 1, 1, 800, 4),
    # round 6: fp32 tanh cells and the fp32 width 1280 (LSTM: 80 KB of weights per wave) -- no fp32 shape of the test suite is left on
    # the launch-per-time-step kernels
    (torch.float32, "rnn", 2, 5, 1024, 7), (torch.float32, "rnn", 1, 9, 800, 6), (torch.float32, "lstm", 1, 18, 1280, 5),
    (torch.float32, "lstm", 2, 3, 1280, 4), (torch.float32, "gru", 2, 8, 1280, 5), (torch.float32, "rnn", 2, 12, 1280, 4),
    # bf16 storage: H = 800 and config 5's LSTM-1280 with 1, 2 and 4 m-tiles (up to 64 samples per group)
    (torch.bfloat16, "gru", 2, 8, 800, 9), (torch.bfloat16, "lstm", 2, 13, 800, 6), (torch.bfloat16, "gru", 2, 7, 1280, 8),
    (torch.bfloat16, "lstm", 1, 64, 1280, 5), (torch.bfloat16, "lstm", 2, 64, 1280, 5), (torch.bfloat16, "lstm", 2, 40, 1280, 4),
    (torch.bfloat16, "gru", 2, 64, 1280, 4), (torch.bfloat16, "lstm", 1, 3, 1280, 6),
]


# round-4 general kernels (ds2_rnn_persist3_impl.h: bf16, 32 units per workgroup): one / two sample sets, lane-shared
# gathers (<= 8 samples per group), XCD-local groups (H <= 1024) and groups that span XCDs (1280, 1536), ragged K splits (800:
# 25 k-steps over 4 waves), LDS-resident weight tails (LSTM-1280 forward 3 of 10 k-steps, BPTT 12 of 40), empty group slots (N = 3)
PERSIST3_CASES = [
    ("lstm", 2, 64, 1280, 6), ("lstm", 1, 64, 1280, 5), ("lstm", 2, 40, 1280, 4), ("gru", 2, 64, 1280, 4), ("lstm", 1, 3, 1280, 6),
    ("gru", 2, 7, 1280, 8), ("gru", 2, 8, 800, 9), ("lstm", 2, 13, 800, 6), ("lstm", 2, 128, 800, 4), ("gru", 2, 3, 512, 7),
    ("lstm", 1, 100, 512, 5), ("gru", 2, 96, 1024, 4), ("lstm", 2, 128, 1024, 3), ("gru", 1, 20, 768, 6), ("lstm", 2, 50, 768, 5),
    ("gru", 2, 30, 1536, 4), ("gru", 1, 9, 1536, 5),
    # round 5: the widths in between (every bf16 hidden size up to 1536 is at most 128 zero units away from a persistent kernel)
    ("gru", 2, 9, 384, 6), ("lstm", 1, 40, 384, 5), ("gru", 2, 33, 640, 5), ("lstm", 2, 8, 640, 6), ("gru", 1, 7, 896, 6),
    ("lstm", 2, 64, 896, 4), ("gru", 2, 64, 1152, 4), ("lstm", 2, 11, 1152, 5), ("gru", 2, 16, 1408, 5), ("gru", 1, 40, 1408, 4),
    # round 6: sets of <= 8 clips on the structured-sparse products (H % 256 == 0, <= 16 clips per group): one set (<= 8 clips) and
    # two sets (8 + the rest), XCD-local groups and groups across XCDs, LDS-resident k-blocks in BPTT (LSTM-1280: 6 of 20 per wave)
    ("lstm", 1, 64, 1280, 6), ("lstm", 2, 64, 1024, 5), ("gru", 2, 48, 512, 6), ("gru", 1, 100, 768, 5), ("lstm", 1, 90, 1536 // 2, 4),
    ("gru", 2, 27, 1280, 7), ("lstm", 2, 32, 1280, 5),
]
SPARSE3_CASES = [("lstm", 1, 64, 1280, 9), ("lstm", 1, 3, 1280, 6), ("gru", 2, 30, 1536, 4), ("gru", 2, 64, 1024, 6), ("lstm", 2, 48, 512, 7),
                 ("gru", 1, 100, 768, 5), ("gru", 2, 7, 1280, 8)]


@pytest.mark.parametrize("kind,D,N,H,Tp", SPARSE3_CASES)
def test_rnn_persist3_sparse_and_dense_sets_agree(kind, D, N, H, Tp):
    """Round 6: sets of <= 8 clips run on the structured-sparse instruction in the general kernels too (tile row s = a clip's
    k = 0, 1 (mod 4) elements, row s + 8 its k = 2, 3 (mod 4) ones): by default groups of <= 8 clips (one set); routing bit 7 also
    cuts groups of 9-16 into two sparse sets (measured slower than one dense set, kept for A/B and covered here).  Same bf16 operands
    as the dense 16-row tiles (routing bit 6 keeps those); every form is held to the oracle by _rnn_sweep_case; against each other
    only the fp32 summation order differs."""
    o = ops()
    assert o.use_persistent(kind, torch.bfloat16, D, N, H) and o.persist_kind(torch.bfloat16, kind, D, N, H) == 3
    with o.persist_options(variant=128):                   # sparse sets wherever they exist (<= 16 clips per group)
        res_sp = _rnn_sweep_case(torch.bfloat16, kind, D, N, H, Tp, 1.0 / np.sqrt(H))
        o.check_persistent_kernels()
    with o.persist_options(variant=64):                    # dense tiles everywhere
        assert o.persist_kind(torch.bfloat16, kind, D, N, H) == 3
        res_de = _rnn_sweep_case(torch.bfloat16, kind, D, N, H, Tp, 1.0 / np.sqrt(H))
        o.check_persistent_kernels()
    for a, b in zip(res_sp, res_de):
        assert np.abs(a - b).max() <= 2e-2 * max(1.0, np.abs(b).max()), np.abs(a - b).max()


@pytest.mark.parametrize("kind,D,N,H,Tp", PERSIST3_CASES)
def test_rnn_persist3_sweeps(kind, D, N, H, Tp):
    """Round-4 general persistent kernels against the oracle, against the launch-per-time-step kernels and (where the round-2
    general kernels cover the shape) against those, on the same inputs."""
    o = ops()
    dtype = torch.bfloat16
    from deepspeech.pytorch_amd._lib import query
    assert o.use_persistent(kind, dtype, D, N, H), "persistent path not selected on this device"
    res_p = _rnn_sweep_case(dtype, kind, D, N, H, Tp, 1.0 / np.sqrt(H))
    o.check_persistent_kernels()
    others = []
    with o.persist_options(variant=1):                     # round-2 general kernels, if they take the shape
        if query("ds2_rnn_persist_supported", o.dt(dtype), o.CELLS[kind], D, N, H, 1):
            others.append(_rnn_sweep_case(dtype, kind, D, N, H, Tp, 1.0 / np.sqrt(H)))
            o.check_persistent_kernels()
    o.PERSIST_ENABLED = False
    try:
        others.append(_rnn_sweep_case(dtype, kind, D, N, H, Tp, 1.0 / np.sqrt(H)))
    finally:
        o.PERSIST_ENABLED = True
    for res_s in others:
        for a_, b_ in zip(res_p, res_s):   # same operands; only the fp32 summation order differs
            assert np.abs(a_ - b_).max() <= 2e-2 * max(1.0, np.abs(b_).max()), np.abs(a_ - b_).max()


def _repeat_sweeps_bit_identically(kind, D, N, H, Tp, lens_np, rounds):
    """`rounds` forward + BPTT sweeps on the same operands: every launch bit-identical to the first, no time-out, and the first equal
    to the launch-per-step kernels."""
    o = ops()
    G = o.GATES[kind]
    dev = "cuda"
    torch.manual_seed(0)
    GI = torch.randn(Tp * N, D * G * H, device=dev).to(torch.bfloat16)
    Whh = ((torch.rand(D, G * H, H, device=dev) * 2 - 1) / H ** 0.5).to(torch.bfloat16)
    WhhT = Whh.transpose(1, 2).contiguous()
    bhh = torch.zeros(D, G * H, device=dev)
    lens = torch.from_numpy(lens_np.astype(np.int32)).to(dev)
    dout = torch.randn(Tp, N, H, device=dev).to(torch.bfloat16)
    assert o.use_persistent(kind, torch.bfloat16, D, N, H)
    o.PERSIST_ENABLED = False
    try:
        hext_r, Sv_r, _, _ = o.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
        dGI_r = o.rnn_bwd(kind, dout, WhhT, hext_r, Sv_r, lens, D, N, H, Tp).dGI
    finally:
        o.PERSIST_ENABLED = True
    first = None
    for it in range(rounds):
        hext, Sv, _, _ = o.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
        dGI = o.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp).dGI
        torch.cuda.synchronize()
        o.check_persistent_kernels()
        if first is None:
            first = (hext.clone(), dGI.clone())
            assert (hext.float() - hext_r.float()).abs().max().item() <= 2e-2
            assert (dGI.float() - dGI_r.float()).abs().max().item() <= 2e-2 * max(1.0, dGI_r.float().abs().max().item())
        else:
            assert torch.equal(hext, first[0]) and torch.equal(dGI, first[1]), it


@pytest.mark.parametrize("kind,D", [("gru", 1), ("lstm", 2)])
def test_rnn_persist3_sixteen_groups_repeat_bit_identically(kind, D):
    """H = 512 runs TWO groups per XCD -- up to 16 groups, the only width with more than eight.  Their XCC-id handshake words used to
    overflow the 2 KB of the scratch head that eight groups need, onto the spin budget and the first group's exchange slots: about
    one launch in ten read a stale word at the second time step (found by a repeated whole-model step, round 4).  Ten forward +
    BPTT sweeps over 451 ragged steps: every launch bit-identical to the first and equal to the launch-per-step kernels."""
    N, Tp = 16, 451
    lens_np = np.sort(np.random.RandomState(0).randint(Tp // 3, Tp + 1, N))[::-1].copy()
    lens_np[0] = Tp
    _repeat_sweeps_bit_identically(kind, D, N, 512, Tp, lens_np, 10)


@pytest.mark.parametrize("kind,D,N,H,lens", [
    ("lstm", 2, 5, 1280, [66, 60, 51, 39, 30]),        # groups of 2, 2 and 1 clips; the last group's only clip ends at step 39 of 66
    ("gru", 2, 9, 1024, [90, 90, 80, 33, 31, 30, 12, 11, 10]),
    ("lstm", 1, 64, 1280, None),                       # config 5b's groups (11 clips), lengths spread over [T'/3, T']
])
def test_rnn_persist3_groups_of_short_clips_stay_in_lock_step(kind, D, N, H, lens):
    """The four-slot exchange relies on LOCK-STEP: a workgroup reads every peer's publish of the step before, so none can run more
    than a step ahead and re-arm a slot somebody still needs.  A variant of these kernels gathered only the rows of clips inside
    their sequences; a workgroup of a group whose clips had all ended then stopped waiting, ran ahead, and its peers timed out on
    a slot it had passed (caught once in a whole suite run, round 4).  Groups whose clips end long before T', thirty launches:
    no time-out, bit-identical results."""
    if lens is None:
        Tp = 151
        lens_np = np.sort(np.random.RandomState(3).randint(Tp // 3, Tp + 1, N))[::-1].copy()
        lens_np[0] = Tp
    else:
        lens_np, Tp = np.array(lens), max(lens)
    _repeat_sweeps_bit_identically(kind, D, N, H, Tp, lens_np, 30)


def test_rnn_persist3_long_ragged_sweep():
    """T' = 311 ragged-length steps through the round-4 kernels and the tuned H = 1024 kernels: the four-slot wrap, the re-arm and
    the bias accumulation over hundreds of steps at KERNEL level (the whole-model tests cover them only indirectly)."""
    for kind, D, N, H in (("lstm", 2, 44, 1280), ("gru", 2, 32, 1024), ("gru", 2, 12, 800)):
        o = ops()
        assert o.use_persistent(kind, torch.bfloat16, D, N, H)
        _rnn_sweep_case(torch.bfloat16, kind, D, N, H, 311, 0.5 / np.sqrt(H), tol_scale=3.0)
        o.check_persistent_kernels()


@pytest.mark.parametrize("dtype,kind,D,N,H,Tp", GENERAL_PERSISTENT_CASES)
def test_rnn_persistent_general_sweeps(dtype, kind, D, N, H, Tp):
    """The generalised persistent kernels (ds2_rnn_persist2_impl.h: H = 800 / 1024 / 1280, bf16 and fp32 storage, 1-4 m-tiles)
    against the oracle, and against the per-time-step kernels on the same inputs."""
    o = ops()
    assert o.use_persistent(kind, dtype, D, N, H), "persistent path not selected on this device"
    res_p = _rnn_sweep_case(dtype, kind, D, N, H, Tp, 1.0 / np.sqrt(H))
    o.check_persistent_kernels()
    o.PERSIST_ENABLED = False
    try:
        res_s = _rnn_sweep_case(dtype, kind, D, N, H, Tp, 1.0 / np.sqrt(H))
    finally:
        o.PERSIST_ENABLED = True
    bar = 2e-2 if dtype == torch.bfloat16 else 1e-5
    for a_, b_ in zip(res_p, res_s):   # same operands; only the fp32 summation order differs
        assert np.abs(a_ - b_).max() <= bar * max(1.0, np.abs(b_).max()), np.abs(a_ - b_).max()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind,D,N,H,Tp", [("gru", 2, 5, 32, 9), ("lstm", 2, 4, 48, 7), ("rnn", 1, 6, 32, 8), ("lstm", 1, 20, 64, 6),
                                           ("gru", 2, 8, 1024, 5)])
def test_rnn_bptt_with_an_initial_state(dtype, kind, D, N, H, Tp):
    """Round 6: backward through a forward that was given `hs` (reference model.py:224-230).  The launch-per-time-step BPTT takes
    h0 / c0 -- h_{t-1} / c_{t-1} of every clip's FIRST step (for the reverse direction that is step length - 1) -- and returns
    d loss / d h0, d c0 (one launch behind the sweep folds the last step's recurrent term in).  Against the oracle (held to torch's
    nn.GRU / nn.LSTM / nn.RNN with hx by tests/test_oracle_ops_vs_torch.py): gate gradients and the state gradients."""
    rs = np.random.RandomState(7 + D + N + H)
    G = O.GATES[kind]
    lens = np.sort(rs.randint(1, Tp + 1, size=N))[::-1].copy()
    lens[0] = Tp
    Whh = rs.uniform(-1, 1, (D, G * H, H)) / np.sqrt(H)
    bhh = rs.uniform(-0.2, 0.2, (D, G * H))
    GI = rs.standard_normal((Tp * N, D * G * H)) * 0.5
    h0 = rs.standard_normal((D, N, H)) * 0.5
    c0 = rs.standard_normal((D, N, H)) * 0.5
    dout = rs.standard_normal((Tp, N, H))
    GI_r, Whh_r, dout_r = rnd(GI, dtype), rnd(Whh, dtype), rnd(dout, dtype)
    o = ops()
    lens_d = torch.from_numpy(lens.astype(np.int32)).to(DEV)
    h0d, c0d = cu(h0), (cu(c0) if kind == "lstm" else None)
    hext, Sv, hn, cn = o.rnn_fwd(kind, cu(GI, dtype), cu(Whh, dtype), cu(bhh), lens_d, D, N, H, Tp, h0=h0d, c0=c0d)
    rg = o.rnn_bwd(kind, cu(dout, dtype), cu(Whh.transpose(0, 2, 1), dtype), hext, Sv, lens_d, D, N, H, Tp, h0=h0d, c0=c0d,
                   want_dstate=True)
    o.check_persistent_kernels()
    dgi = np64(rg.dGI).reshape(Tp, N, D, G * H)
    tol = TOL[dtype] * (1 if dtype == torch.float32 else 4)
    for d in range(D):
        out, hnr, cnr, cache = O.rnn_dir_fwd(kind, GI_r.reshape(Tp, N, D, G * H)[:, :, d], lens, np.eye(G * H), Whh_r[d], np.zeros(G * H),
                                             bhh[d], reverse=(d == 1), h0=h0[d], c0=c0[d] if kind == "lstm" else None)
        assert np.abs(np64(hext[d, 1:Tp + 1]) - out).max() < tol
        dx, _, _, _, _, dh0, dc0 = O.rnn_dir_bwd(cache, dout_r, np.eye(G * H), Whh_r[d], return_dstate=True)
        sc = max(np.abs(dx).max(), 1e-6)
        assert np.abs(dgi[:, :, d] - dx).max() / sc < tol * (1 if dtype == torch.float32 else 3), (kind, d)
        assert np.abs(np64(rg.dh0[d]) - dh0).max() / max(np.abs(dh0).max(), 1e-6) < tol * (1 if dtype == torch.float32 else 3), (kind, d, "dh0")
        if kind == "lstm":
            assert np.abs(np64(rg.dc0[d]) - dc0).max() / max(np.abs(dc0).max(), 1e-6) < tol * (1 if dtype == torch.float32 else 3), (kind, d, "dc0")


def test_rnn_persistent_general_initial_state():
    """h0/c0 carry (reference inference.py:86-96) through the general persistent forward kernels (fp32 and bf16)."""
    rs = np.random.RandomState(13)
    for dt_, kind, D, N, H, Tp in ((torch.float32, "lstm", 2, 1, 800, 6), (torch.bfloat16, "gru", 1, 3, 1280, 5)):
        G = O.GATES[kind]
        GI = rs.standard_normal((Tp * N, D * G * H))
        Whh, bhh = rs.uniform(-0.05, 0.05, (D, G * H, H)), rs.uniform(-0.2, 0.2, (D, G * H))
        h0, c0 = rs.standard_normal((D, N, H)), rs.standard_normal((D, N, H))
        lens = np.array([Tp] + [max(1, Tp - 2)] * (N - 1), dtype=np.int32)
        GI_r, Whh_r = rnd(GI, dt_), rnd(Whh, dt_)
        o = ops()
        assert o.use_persistent(kind, dt_, D, N, H)
        hext, Sv, hn, cn = o.rnn_fwd(kind, cu(GI, dt_), cu(Whh, dt_), cu(bhh), torch.from_numpy(lens).to(DEV), D, N, H, Tp,
                                     h0=cu(h0), c0=cu(c0) if kind == "lstm" else None)
        o.check_persistent_kernels()
        tol = 4e-2 if dt_ == torch.bfloat16 else 1e-4
        for d in range(D):
            out, hn_ref, cn_ref, _ = O.rnn_dir_fwd(kind, GI_r.reshape(Tp, N, D, G * H)[:, :, d], lens, np.eye(G * H), Whh_r[d],
                                                   np.zeros(G * H), bhh[d], reverse=(d == 1), h0=h0[d],
                                                   c0=c0[d] if kind == "lstm" else None)
            assert np.abs(np64(hext[d, 1:Tp + 1]) - out).max() < tol
            assert np.abs(np64(hn[d]) - hn_ref).max() < tol
            if kind == "lstm":
                assert np.abs(np64(cn[d]) - cn_ref).max() < tol * 1.5


def _rnn_sweep_case(dtype, kind, D, N, H, Tp, wscale, tol_scale=1.0, sort_lens=True, min_len=1):
    rs = np.random.RandomState(D * 100 + N + H + Tp)
    G = O.GATES[kind]
    I = 24
    lens = rs.randint(min_len, Tp + 1, size=N)
    if sort_lens:
        lens = np.sort(lens)[::-1].copy()
        lens[0] = Tp
    else:
        lens[N // 2] = Tp
    x = rs.standard_normal((Tp, N, I))
    for i, ln in enumerate(lens):
        x[ln:, i] = 0
    Wih = rs.uniform(-0.3, 0.3, (D, G * H, I))
    Whh = rs.uniform(-wscale, wscale, (D, G * H, H))
    bih, bhh = rs.uniform(-0.2, 0.2, (D, G * H)), rs.uniform(-0.2, 0.2, (D, G * H))
    Whh_r = rnd(Whh, dtype)
    o = ops()
    # the input projection is part of the GEMM tests; feed the sweeps the oracle's projection
    GI = np.stack([x.reshape(Tp * N, I) @ Wih[d].T + bih[d] for d in range(D)], axis=1).reshape(Tp * N, D * G * H)
    GI_r = rnd(GI, dtype)
    outs, caches, hns, cns = [], [], [], []
    for d in range(D):
        # oracle with W_ih = identity on the pre-computed (rounded) projection
        out, hn, cn, cache = O.rnn_dir_fwd(kind, GI_r.reshape(Tp, N, D, G * H)[:, :, d], lens, np.eye(G * H), Whh_r[d],
                                           np.zeros(G * H), bhh[d], reverse=(d == 1))
        outs.append(out), caches.append(cache), hns.append(hn), cns.append(cn)
    lens_d = torch.from_numpy(lens.astype(np.int32)).to(DEV)
    hext, Sv, hn_d, cn_d = o.rnn_fwd(kind, cu(GI, dtype), cu(Whh, dtype), cu(bhh), lens_d, D, N, H, Tp)
    got = np64(hext[:, 1:Tp + 1])
    tol = TOL[dtype] * (1 if dtype == torch.float32 else 4) * tol_scale
    for d in range(D):
        assert np.abs(got[d] - outs[d]).max() < tol, (kind, d)
        assert np.abs(np64(hn_d[d]) - hns[d]).max() < tol
        if kind == "lstm":
            assert np.abs(np64(cn_d[d]) - cns[d]).max() < tol * 2
    assert np.all(np64(hext[:, 0]) == 0) and np.all(np64(hext[:, Tp + 1]) == 0)
    # ---- BPTT
    dout = rs.standard_normal((Tp, N, H))
    dout_r = rnd(dout, dtype)
    WhhT = cu(Whh.transpose(0, 2, 1), dtype)
    rg = o.rnn_bwd(kind, cu(dout, dtype), WhhT, hext, Sv, lens_d, D, N, H, Tp)
    dgi = np64(rg.dGI).reshape(Tp, N, D, G * H)
    for d in range(D):
        dx, dwi, dwh, dbi, dbh = O.rnn_dir_bwd(caches[d], dout_r, np.eye(G * H), Whh_r[d])
        # with W_ih = I the oracle's dx IS d loss / d GI
        scale = max(np.abs(dx).max(), 1e-6)
        assert np.abs(dgi[:, :, d] - dx).max() / scale < tol * (1 if dtype == torch.float32 else 3), (kind, d)
        # bias gradients: from the sweep's per-sample sums (persistent kernels) or the column sums of the stored planes
        if rg.bacc is not None:
            b = np64(rg.bacc[d]).sum(0)
            assert relerr(b[:G * H], dbi) < tol * 4, (kind, d, "bias_ih")
            got_bh = np.concatenate([b[:2 * H], b[3 * H:4 * H]]) if kind == "gru" else b[:G * H]
            assert relerr(got_bh, dbh) < tol * 4, (kind, d, "bias_hh")
            if kind == "gru":     # the stored dQ plane sums to the same thing
                assert relerr(np64(rg.dQ[d]).reshape(Tp * N, H).sum(0), dbh[2 * H:]) < tol * 4
        elif kind == "gru":
            assert relerr(np64(rg.dGH[d]).reshape(Tp * N, G * H).sum(0), dbh) < tol * 4
    return got, np64(hn_d), dgi


def test_rnn_persistent_initial_state():
    """h0/c0 carry (reference inference.py:86-96) through the persistent forward kernel, batch 1 and batch 3."""
    rs = np.random.RandomState(12)
    for kind, D, N, H, Tp in (("lstm", 2, 1, 1024, 6), ("gru", 1, 3, 1024, 5)):
        G = O.GATES[kind]
        GI = rs.standard_normal((Tp * N, D * G * H))
        Whh, bhh = rs.uniform(-0.05, 0.05, (D, G * H, H)), rs.uniform(-0.2, 0.2, (D, G * H))
        h0, c0 = rs.standard_normal((D, N, H)), rs.standard_normal((D, N, H))
        lens = np.array([Tp] + [max(1, Tp - 2)] * (N - 1), dtype=np.int32)
        dt_ = torch.bfloat16
        GI_r, Whh_r = rnd(GI, dt_), rnd(Whh, dt_)
        o = ops()
        assert o.use_persistent(kind, dt_, D, N, H)
        hext, Sv, hn, cn = o.rnn_fwd(kind, cu(GI, dt_), cu(Whh, dt_), cu(bhh), torch.from_numpy(lens).to(DEV), D, N, H, Tp,
                                     h0=cu(h0), c0=cu(c0) if kind == "lstm" else None)
        o.check_persistent_kernels()
        for d in range(D):
            # the kernel feeds the bf16-rounded state to the MFMA and keeps the fp32 state for the elementwise carry
            out, hn_ref, cn_ref, _ = O.rnn_dir_fwd(kind, GI_r.reshape(Tp, N, D, G * H)[:, :, d], lens, np.eye(G * H), Whh_r[d],
                                                   np.zeros(G * H), bhh[d], reverse=(d == 1), h0=h0[d],
                                                   c0=c0[d] if kind == "lstm" else None)
            assert np.abs(np64(hext[d, 1:Tp + 1]) - out).max() < 4e-2
            assert np.abs(np64(hn[d]) - hn_ref).max() < 4e-2
            if kind == "lstm":
                assert np.abs(np64(cn[d]) - cn_ref).max() < 6e-2


@pytest.mark.parametrize("dtype", DTYPES)
def test_rnn_initial_state(dtype):
    """hidden-state carry of reference inference.py:86-96: h0/c0 given, batch 1."""
    rs = np.random.RandomState(11)
    D, N, H, Tp, kind = 2, 1, 32, 6, "lstm"
    G = 4
    GI = rs.standard_normal((Tp * N, D * G * H))
    Whh, bhh = rs.uniform(-0.3, 0.3, (D, G * H, H)), rs.uniform(-0.2, 0.2, (D, G * H))
    h0, c0 = rs.standard_normal((D, N, H)), rs.standard_normal((D, N, H))
    lens = np.array([Tp])
    o = ops()
    hext, _, hn, cn = o.rnn_fwd(kind, cu(GI, dtype), cu(Whh, dtype), cu(bhh), torch.from_numpy(lens.astype(np.int32)).to(DEV),
                                D, N, H, Tp, h0=cu(h0), c0=cu(c0))
    for d in range(D):
        out, h, c, _ = O.rnn_dir_fwd(kind, rnd(GI, dtype).reshape(Tp, N, D, G * H)[:, :, d], lens, np.eye(G * H),
                                     rnd(Whh, dtype)[d], np.zeros(G * H), bhh[d], reverse=(d == 1), h0=h0[d], c0=c0[d])
        tol = TOL[dtype] * (1 if dtype == torch.float32 else 4)
        assert np.abs(np64(hext[d, 1:Tp + 1]) - out).max() < tol
        assert np.abs(np64(hn[d]) - h).max() < tol and np.abs(np64(cn[d]) - c).max() < tol * 2


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ctx", [20, 7])
@pytest.mark.parametrize("shape", [(33, 3, 48), (257, 2, 24), (100, 1, 8), (101, 2, 8), (7, 2, 16)])
def test_lookahead(dtype, ctx, shape):
    """ctx = 20 (the reference default) takes the sliding-window kernels: shapes cover one segment, several segments with a ragged
    tail, exactly one full segment, one frame into the next, and a sequence shorter than the context."""
    rs = np.random.RandomState(12)
    Tp, N, H = shape
    x = rs.standard_normal((Tp, N, H)) * 3
    w = rs.uniform(-0.6, 0.9, (H, 1, ctx))
    xr = rnd(x, dtype)
    pre = O.lookahead_fwd(xr, w)
    y = O.hardtanh_fwd(pre)
    o = ops()
    xd, wd = cu(x.reshape(Tp * N, H), dtype), cu(w.reshape(H, ctx))
    yd, pred = o.lookahead_fwd(xd, wd, Tp, N, H)
    assert np.abs(np64(yd).reshape(Tp, N, H) - y).max() < TOL[dtype] * 30
    dy = rs.standard_normal((Tp, N, H))
    g = O.hardtanh_bwd(rnd(pre, dtype), rnd(dy, dtype))
    dx, dw = O.lookahead_bwd(xr, w, g)
    dxd, dwd = o.lookahead_bwd(xd, wd, pred, cu(dy.reshape(Tp * N, H), dtype), Tp, N, H)
    tol = TOL[dtype] * (1 if dtype == torch.float32 else 6)
    assert relerr(np64(dxd).reshape(Tp, N, H), dx) < tol
    assert relerr(np64(dwd), dw.reshape(H, ctx)) < tol


# ---------------------------------------------------------------------------------------------------------------
def _ctc_case(N, Tp, lens, tlens, seed, Cc=29, scale=2.0):
    rs = np.random.RandomState(seed)
    logits = rs.standard_normal((Tp, N, Cc)) * scale
    targets = rs.randint(1, Cc, size=int(np.sum(tlens)))
    # force some repeated labels (the s-2 skip rule)
    if len(targets) > 3:
        targets[1] = targets[0]
        targets[-1] = targets[-2]
    return logits, targets


@pytest.mark.parametrize("N,Tp,lens,tlens", [
    (3, 31, [31, 25, 19], [7, 6, 4]),
    (4, 40, [40, 39, 15, 11], [9, 9, 3, 21]),      # last sample infeasible (S > T'): zero_infinity
    (1, 12, [12], [12]),                            # S == T' (feasible only without repeats)
    (2, 300, [300, 150], [140, 1]),                 # more states than threads (2S+1 = 281 > 256)
    (2, 9, [9, 5], [0, 2]),                         # empty target
    (3, 520, [520, 401, 77], [255, 190, 64]),       # 8 states per lane of the one-wave recursion (2S+1 = 511), 6 and 3 for the others
    (2, 530, [530, 530], [256, 3]),                 # 2S+1 = 513: beyond the one-wave kernel -> the four-wave kernel
    (5, 70, [70, 64, 33, 2, 1], [30, 31, 16, 1, 1]),  # lengths around the 8-step staging chunks; one- and two-frame clips
    (4, 130, [130, 129, 121, 9], [63, 64, 57, 4]),    # pair tiles: 64 / 65 pairs = one wave exactly / the first pair of a second wave
    (3, 300, [300, 290, 250], [111, 112, 48]),        # pair tiles: 112 pairs = two waves exactly (64 + 48 owned), 113 = a third
    (2, 1700, [1700, 900], [783, 400]),               # pair tiles: 16 waves (the most), and 8
    (2, 1700, [1700, 9], [784, 2]),                   # one label beyond the pair tiles: the four-wave kernel's strided form
])
@pytest.mark.parametrize("recursion", [0, 2, 1])
def test_ctc_loss_and_grad(N, Tp, lens, tlens, recursion, monkeypatch):
    # 0: pair tiles (the default); 2: one wave per (sample, direction) up to 255 labels; 1: the four-wave kernel
    if max(tlens) > 600 and recursion == 2:
        pytest.skip("same kernel as recursion = 1 at this size")
    monkeypatch.setattr(ops(), "CTC_RECURSION", recursion)
    _ctc_check(N, Tp, lens, tlens)


@pytest.mark.parametrize("N,Tp,lens,tlens", [(6, 333, [333, 300, 251, 120, 64, 9], [150, 17, 99, 120, 1, 0]),
                                             (3, 600, [600, 411, 600], [255, 200, 3])])
def test_ctc_recursion_kernels_agree_bit_for_bit(N, Tp, lens, tlens):
    """Same formulas, same order: the pair-tile, the one-wave and the four-wave recursion give identical loss, per-sample nll and
    gradient."""
    from deepspeech.pytorch_amd import _lib
    logits, targets = _ctc_case(N, Tp, lens, tlens, seed=99)
    o = ops()
    lg = torch.zeros((Tp * N, 32), dtype=torch.float32, device=DEV)
    lg[:, :29] = cu(logits.reshape(Tp * N, 29))
    offs = np.concatenate([[0], np.cumsum(tlens)[:-1]]).astype(np.int32)
    outs = []
    for recursion in (0, 2, 1, 3):
        outs.append(o.ctc_loss_grad(lg, torch.from_numpy(targets.astype(np.int32)).to(DEV), torch.from_numpy(offs).to(DEV),
                                    torch.from_numpy(np.asarray(lens, np.int32)).to(DEV),
                                    torch.from_numpy(np.asarray(tlens, np.int32)).to(DEV), Tp, N, 29, 0, int(max(tlens)), recursion=recursion))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def _ctc_check(N, Tp, lens, tlens, Cc=29, blank=0):
    logits, targets = _ctc_case(N, Tp, lens, tlens, seed=Tp + N, Cc=Cc)
    if blank != 0:                      # labels avoid the blank index, wherever it sits
        targets = np.where(targets == blank, 0, targets)
    lp = O.log_softmax(logits)
    # (the vectorised oracle recursion for the long cases: it is held to the loop version in tests/test_oracle_vs_golden.py)
    loss_ref, nll_ref, dlp = (O.ctc_loss_and_grad_fast if Tp > 100 else O.ctc_loss_and_grad)(lp, targets, np.asarray(lens), np.asarray(tlens), blank=blank)
    dlogits_ref = dlp - np.exp(lp) * dlp.sum(-1, keepdims=True)
    o = ops()
    ld = (Cc + 31) // 32 * 32
    lg = torch.zeros((Tp * N, ld), dtype=torch.float32, device=DEV)
    lg[:, :Cc] = cu(logits.reshape(Tp * N, Cc))
    offs = np.concatenate([[0], np.cumsum(tlens)[:-1]]).astype(np.int32)
    loss, nll, dl = o.ctc_loss_grad(lg, torch.from_numpy(targets.astype(np.int32)).to(DEV), torch.from_numpy(offs).to(DEV),
                                    torch.from_numpy(np.asarray(lens, np.int32)).to(DEV),
                                    torch.from_numpy(np.asarray(tlens, np.int32)).to(DEV), Tp, N, Cc, blank, int(max(tlens)))
    assert np.allclose(np64(nll), nll_ref, rtol=2e-5, atol=1e-4), (np64(nll), nll_ref)
    assert abs(float(loss.item()) - loss_ref) <= 2e-5 * max(1.0, abs(loss_ref))
    got = np64(dl).reshape(Tp, N, ld)
    # fp32 log-space recursion: alpha/beta reach magnitude ~3*T', whose fp32 ulp (6e-8 * 3T') bounds the precision of
    # exp(alpha + beta - ll); torch's fp32 CTC has the same noise.  Stated bar: 2e-6 * T' absolute (4e-6 * T' for the > 200-label
    # cases, whose nearly forced alignments run the log-space values to ~5 T').
    # larger label sets: log-probabilities ~ -log C, so the log-space values (and their fp32 ulp) grow by log C / log 29
    scale_c = 1.0 if Cc == 29 else 1.25 * np.log(Cc) / np.log(29.0)
    assert np.abs(got[:, :, :Cc] - dlogits_ref).max() < max(2e-5, (4e-6 if max(tlens) > 200 else 2e-6) * Tp) * scale_c
    assert np.all(got[:, :, Cc:] == 0)


@pytest.mark.parametrize("Cc,blank", [(257, 0), (300, 299), (1000, 0), (5000, 17)])
@pytest.mark.parametrize("N,Tp,lens,tlens", [(3, 31, [31, 25, 19], [7, 6, 4]), (4, 40, [40, 39, 15, 11], [9, 9, 3, 21]),
                                             (2, 300, [300, 150], [140, 1]), (2, 9, [9, 5], [0, 2])])
@pytest.mark.parametrize("recursion", [0, 1])
def test_ctc_loss_and_grad_large_label_sets(Cc, blank, N, Tp, lens, tlens, recursion, monkeypatch):
    """More than 256 classes (the reference's labels file is unbounded, model.py:139,203; a Mandarin model has thousands): the class
    rows are no longer staged in LDS, every state reads the log-probability of its own label (pair tiles; recursion = 1:
    k_ctc_recursion_big)."""
    monkeypatch.setattr(ops(), "CTC_RECURSION", recursion)
    _ctc_check(N, Tp, lens, tlens, Cc=Cc, blank=blank)


def test_softmax_rows():
    rs = np.random.RandomState(13)
    x = rs.standard_normal((77, 29)) * 4
    e = np.exp(x - x.max(-1, keepdims=True))
    ref = e / e.sum(-1, keepdims=True)
    lg = torch.zeros((77, 32), dtype=torch.float32, device=DEV)
    lg[:, :29] = cu(x)
    assert np.abs(np64(ops().softmax_rows(lg, 29)) - ref).max() < 1e-6


def test_cpu_tensor_is_refused():
    from deepspeech.pytorch_amd import _lib
    with pytest.raises(_lib.Ds2HipError):
        ops().add2(torch.zeros(8), torch.zeros(8))


@pytest.mark.parametrize("dtype", DTYPES)
def test_small_weight_layouts_equal_the_torch_expressions(dtype):
    """ds2_small_weight_layouts (one launch) against the tensor expressions that define the kernel layouts of the conv / head
    weights: conv1 tap-major, conv2 tap-major, the flipped row-parity sub-kernels of the conv2 data gradient, the zero-padded head
    weight and its transpose."""
    rs = np.random.RandomState(9)
    H, Cc = 96, 29
    w1 = torch.from_numpy(rs.standard_normal((32, 1, 41, 11)).astype(np.float32)).to(DEV)
    w2 = torch.from_numpy(rs.standard_normal((32, 32, 21, 11)).astype(np.float32)).to(DEV)
    wf = torch.from_numpy(rs.standard_normal((Cc, H)).astype(np.float32)).to(DEV)
    w1k, w2t, w2d, wp, wpT = ops().small_weight_layouts(w1, w2, wf, dtype)
    assert torch.equal(w1k, w1.reshape(32, 451).t().contiguous())
    assert torch.equal(w2t, w2.permute(2, 3, 0, 1).contiguous().to(dtype))
    for q in (0, 1):
        assert torch.equal(w2d[q], w2[:, :, q::2, :].flip(2, 3).permute(2, 3, 1, 0).contiguous().to(dtype))
    ref = torch.cat([wf, torch.zeros(32 - Cc, H, device=DEV)], 0).to(dtype)
    assert torch.equal(wp, ref) and torch.equal(wpT, ref.t().contiguous())
