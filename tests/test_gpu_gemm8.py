"""The 256x256 phase-split bf16 GEMM kernels (csrc/ds2_gemm8.hip) against fp32 products of the same bf16 operands computed by torch
on the device: NT (input projections / dX, reference model.py:97-99) and grouped TN (weight gradients: both operands K-major).
Shapes cover edge tiles in M and N, 1 / 2 / odd / even numbers of K-tiles (prologue and drain paths of the DMA pipeline), bias,
both output types, several problems per launch and the two-buffer A operand of the GRU's hidden-side gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(shape, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1).to(DEV).to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K,f32,bias", [
    (256, 256, 64, True, False),          # one tile, one K-tile
    (300, 520, 128, False, True),         # edge tiles in M and N, two K-tiles
    (1000, 264, 192, True, True),         # three K-tiles
    (2048, 1344, 320, False, False),      # five K-tiles, N = the padded conv feature count
    (24032, 1024, 1024, False, True),     # cfg3's dX / input-projection row count
])
def test_gemm8_nt_matches_fp32_product(M, N, K, f32, bias):
    from deepspeech.pytorch_amd import ops
    A, B = _rand((M, K), 1), _rand((N, K), 2)
    bv = torch.linspace(-1, 1, N, device=DEV) if bias else None
    ref = A.float() @ B.float().t()
    if bias:
        ref = ref + bv
    for rep in range(2):                                                    # a second launch: no state carried between launches
        C = ops.gemm8_nt(A, B, bias=bv, out_dtype=torch.float32 if f32 else None)
        assert C.dtype == (torch.float32 if f32 else torch.bfloat16) and tuple(C.shape) == (M, N)
        err = (C.float() - ref).abs().max().item() / ref.abs().max().item()
        assert err < (2e-5 if f32 else 6e-3), (M, N, K, err)


def test_gemm8_nt_strided_operand_views():
    """A and B as column windows of wider matrices (the layer-0 input projection reads a [T'N][1344] window)."""
    from deepspeech.pytorch_amd import ops
    Aw, Bw = _rand((700, 512), 3), _rand((300, 384), 4)
    A, B = Aw[:, 128:384], Bw[:, 64:320]                                    # K = 256, lda = 512, ldb = 384
    C = ops.gemm8_nt(A, B, out_dtype=torch.float32, M=700, N=300, K=256, lda=512, ldb=384)
    ref = A.float() @ B.float().t()
    assert (C - ref).abs().max().item() / ref.abs().max().item() < 2e-5


@pytest.mark.parametrize("K", [64, 128, 192, 448, 40, 100, 333])
def test_gemm8_tn_grouped_matches_fp32_products(K):
    from deepspeech.pytorch_amd import ops
    # three products of different shapes in one launch; the second has its A operand in two buffers (rows >= 512 from At2)
    # (ragged K: the rows past K - 1 of the last K-tile are neither read nor used -- the operands end exactly at row K - 1, and
    # the neighbouring allocation is poisoned with NaN)
    At0, Bt0 = _rand((K, 520), 5), _rand((K, 264), 6)
    poison = torch.full((1 << 20,), float("nan"), device=DEV)
    At1, At1b, Bt1 = _rand((K, 768), 7), _rand((K, 256), 8), _rand((K, 256), 9)
    At2w, Bt2w = _rand((K, 1024), 10), _rand((K, 640), 11)
    probs = [dict(At=At0, Bt=Bt0, M=520, N=264, lda=520, ldb=264),
             dict(At=At1, At2=At1b, lda2=256, m_split=512, Bt=Bt1, M=768, N=256, lda=768, ldb=256),
             dict(At=At2w[:, 256:], Bt=Bt2w[:, 128:], M=768, N=512, lda=1024, ldb=640)]
    outs = ops.gemm8_tn_grouped(probs, K)
    refs = [At0.float().t() @ Bt0.float(),
            torch.cat([At1[:, :512], At1b], 1).float().t() @ Bt1.float(),
            At2w[:, 256:].float().t() @ Bt2w[:, 128:].float()]
    for o, r in zip(outs, refs):
        assert tuple(o.shape) == tuple(r.shape)
        err = (o - r).abs().max().item() / r.abs().max().item()
        assert err < 2e-5, (K, tuple(o.shape), err)


def test_gemm8_tn_weight_gradient_shape():
    """cfg3's per-layer weight gradients as ONE grouped launch: dW_ih [6144][1024] and the two dW_hh [3072][1024] with the GRU's
    [dr, dz | dQ] operand split, contraction over a zero-padded T'N = 2 x 1504 rows (a short sequence keeps the check cheap)."""
    from deepspeech.pytorch_amd import ops
    R, H = 3008, 1024
    dGI, X = _rand((R, 6 * H), 12), _rand((R, H), 13)
    dQ, Hp = _rand((2, R, H), 14), _rand((2, R, H), 15)
    probs = [dict(At=dGI, Bt=X, M=6 * H, N=H, lda=6 * H, ldb=H)]
    for d in range(2):
        # rows [0, 2H) of direction d's gate gradient come from dGI, rows [2H, 3H) from dQ
        probs.append(dict(At=dGI[:, d * 3 * H:], At2=dQ[d], lda2=H, m_split=2 * H, Bt=Hp[d], M=3 * H, N=H, lda=6 * H, ldb=H))
    outs = ops.gemm8_tn_grouped(probs, R)
    refs = [dGI.float().t() @ X.float()]
    for d in range(2):
        refs.append(torch.cat([dGI[:, d * 3 * H:d * 3 * H + 2 * H], dQ[d]], 1).float().t() @ Hp[d].float())
    for o, r in zip(outs, refs):
        assert (o - r).abs().max().item() / r.abs().max().item() < 2e-5


def test_gemm8_weight_gradients_and_dx_in_one_launch():
    """The mixed launch (TN weight-gradient problems + the NT dX product, ds2_gemm8_wgrad_dx) gives exactly what the two separate
    launches give -- the same tiles, the same order of summation."""
    from deepspeech.pytorch_amd import ops
    R, H = 3008 + 40, 1024                                   # ragged K for the TN problems
    dGI, X = _rand((R, 6 * H), 21), _rand((R, H + 64), 22)[:, :H]
    dQ, Hp = _rand((2, R, H), 23), _rand((2, R, H), 24)
    WihT = _rand((H, 6 * H), 25)

    def problems():
        probs = [dict(At=dGI, Bt=X, M=6 * H, N=H, lda=6 * H, ldb=X.stride(0))]
        for d in range(2):
            probs.append(dict(At=dGI[:, d * 3 * H:], At2=dQ[d], lda2=H, m_split=2 * H, Bt=Hp[d], M=3 * H, N=H, lda=6 * H, ldb=H))
        return probs
    outs_a = ops.gemm8_tn_grouped(problems(), R)
    dx_a = ops.gemm8_nt(dGI, WihT)
    outs_b, dx_b = ops.gemm8_tn_grouped(problems(), R, dx=(dGI, WihT))
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    assert torch.equal(dx_a, dx_b)
    ref = dGI.float() @ WihT.float().t()
    assert (dx_b.float() - ref).abs().max().item() / ref.abs().max().item() < 6e-3
    ref_w = dGI.float().t() @ X.float()
    assert (outs_b[0] - ref_w).abs().max().item() / ref_w.abs().max().item() < 2e-5


def test_gemm8_race_screen_bit_identical_under_uneven_load():
    """The kernels order their LDS hand-offs by counters and barriers, never by timing; the summation order is fixed.  So repeated
    launches must be BIT-identical whatever else the chip does: 25 rounds of an NT, a grouped TN and a mixed launch while a second
    stream streams HBM (uneven load moves the DMA landing times around), every output compared with the first round's bit for bit."""
    from deepspeech.pytorch_amd import ops
    A, B = _rand((2048 + 77, 576), 31), _rand((1024 + 40, 576), 32)
    R, H = 1000, 512
    dG, X, Hp = _rand((R, 3 * H), 33), _rand((R, H), 34), _rand((R, H), 35)
    W = _rand((H, 3 * H), 36)

    def problems():
        return [dict(At=dG, Bt=X, M=3 * H, N=H, lda=3 * H, ldb=H), dict(At=dG[:, H:], Bt=Hp, M=2 * H, N=H, lda=3 * H, ldb=H)]
    hog_src = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    hog_dst = torch.empty_like(hog_src)
    side = torch.cuda.Stream()
    first = None
    for rnd in range(25):
        if rnd % 2 == 1:
            with torch.cuda.stream(side):
                for _ in range(1 + rnd % 3):
                    hog_dst.copy_(hog_src, non_blocking=True)
        c_nt = ops.gemm8_nt(A, B, out_dtype=torch.float32)
        c_tn = ops.gemm8_tn_grouped(problems(), R)
        c_mx, dx = ops.gemm8_tn_grouped(problems(), R, dx=(dG, W))
        torch.cuda.synchronize()
        cur = [c_nt] + c_tn + c_mx + [dx]
        if first is None:
            first = [t.clone() for t in cur]
            ref = A.float() @ B.float().t()
            assert (c_nt - ref).abs().max().item() / ref.abs().max().item() < 2e-5
        else:
            for a, b in zip(first, cur):
                assert torch.equal(a, b), rnd


def _frame_list(Tp, N, seed):
    """Row list of a padded [T' x N] sequence matrix with sorted-or-not ragged lengths: (lens int32 device, rows int32 device)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    lens = rs.randint(max(1, Tp // 3), Tp + 1, N).astype(np.int32)
    lens[rs.randint(N)] = Tp
    rows = np.flatnonzero(np.arange(Tp)[:, None] < lens[None, :]).astype(np.int32)
    return torch.from_numpy(lens).to(DEV), torch.from_numpy(rows).to(DEV)


@pytest.mark.parametrize("Tp,N,K,Nc,f32,bias", [
    (37, 8, 64, 264, True, True),          # 296 physical rows, ~200 listed: edge tile in M
    (151, 32, 320, 1344, False, True),     # five K-tiles, bf16 epilogue through LDS
    (301, 64, 1280, 512, False, False),    # cfg5-shaped rows: 19264 physical
])
def test_gemm8_nt_row_list_visits_exactly_the_listed_rows(Tp, N, K, Nc, f32, bias):
    """ds2_gemm8_nt_rows: listed rows equal the full product's rows BIT for bit (same tile arithmetic, only the row -> address map
    differs); unlisted rows keep what the output buffer held (here: NaN poison), and ds2_zero_pad_rows zeroes exactly those."""
    from deepspeech.pytorch_amd import ops, _lib
    R = Tp * N
    lens, rows = _frame_list(Tp, N, 41)
    A, B = _rand((R, K), 42), _rand((Nc, K), 43)
    bv = torch.linspace(-1, 1, Nc, device=DEV) if bias else None
    full = ops.gemm8_nt(A, B, bias=bv, out_dtype=torch.float32 if f32 else None)
    out = torch.full_like(full, float("nan"))
    _lib.call("ds2_gemm8_nt_rows", ops.P(A), ops.P(B), ops.P(out), ops.PF(bv), R, Nc, K, K, K, Nc, 1 if f32 else 0, ops.P(rows), rows.numel(),
              ops.S())
    listed = torch.zeros(R, dtype=torch.bool, device=DEV)
    listed[rows.long()] = True
    assert torch.equal(out[listed], full[listed])
    assert torch.isnan(out[~listed]).all()
    ops.zero_pad_rows(out, lens, Tp, N)
    assert torch.equal(out[listed], full[listed]) and (out[~listed] == 0).all()
    # the wrapper: same thing in one call
    out2 = ops.gemm_nt(A, B, bias=bv, out_dtype=torch.float32 if f32 else None, rows=rows, zero_pad=(lens, Tp, N))
    assert torch.equal(out2[listed], full[listed]) and (out2[~listed] == 0).all()


@pytest.mark.parametrize("Tp,N", [(37, 8), (151, 32), (203, 64)])
def test_gemm8_tn_row_list_contracts_over_the_listed_rows(Tp, N):
    """Grouped TN products over a row list = the products over the gathered rows (unlisted rows are poisoned with NaN: they must not
    be read into any sum); the mixed launch's dX has the listed rows of the full product and zeros elsewhere."""
    from deepspeech.pytorch_amd import ops
    R, H = Tp * N, 512
    lens, rows = _frame_list(Tp, N, 51)
    listed = torch.zeros(R, dtype=torch.bool, device=DEV)
    listed[rows.long()] = True
    dG, X, Hp, dQ = _rand((R, 3 * H), 52), _rand((R, H + 64), 53)[:, :H], _rand((R, H), 54), _rand((R, H), 55)
    W = _rand((H, 3 * H), 56)
    dGp, Xp, Hpp, dQp = dG.clone(), X.clone(), Hp.clone(), dQ.clone()
    for t_ in (dGp, Xp, Hpp, dQp):
        t_[~listed] = float("nan")

    def problems(g, x, hp, dq):
        return [dict(At=g, Bt=x, M=3 * H, N=H, lda=3 * H, ldb=x.stride(0)),
                dict(At=g, At2=dq, lda2=H, m_split=2 * H, Bt=hp, M=3 * H, N=H, lda=3 * H, ldb=H)]
    outs = ops.gemm8_tn_grouped(problems(dGp, Xp, Hpp, dQp), R, rows=rows)
    li = rows.long()
    refs = [dG[li].float().t() @ X[li].float(), torch.cat([dG[li][:, :2 * H], dQ[li]], 1).float().t() @ Hp[li].float()]
    for o, r in zip(outs, refs):
        assert torch.isfinite(o).all()
        assert (o - r).abs().max().item() / r.abs().max().item() < 2e-5
    # equal, bit for bit, to the same launch over physically gathered operands (same K-tiles, same order of summation)
    outs_g = ops.gemm8_tn_grouped(problems(dG[li].contiguous(), X[li].contiguous(), Hp[li].contiguous(), dQ[li].contiguous()), li.numel())
    for a, b in zip(outs, outs_g):
        assert torch.equal(a, b)
    outs_m, dx = ops.gemm8_tn_grouped(problems(dGp, Xp, Hpp, dQp), R, dx=(dGp, W), rows=rows, zero_pad=(lens, Tp, N))
    for a, b in zip(outs, outs_m):
        assert torch.equal(a, b)
    full = ops.gemm8_nt(dG, W)
    assert torch.equal(dx[listed], full[listed]) and (dx[~listed] == 0).all()
