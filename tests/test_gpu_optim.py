"""GPU parity of the optimizer kernels (csrc/ds2_optim.hip) against torch.optim on the CPU -- the reference's own optimizers
(reference model.py:273-297) in their single-tensor fp32 arithmetic -- and against torch.nn.utils.clip_grad_norm_ (Lightning's
gradient_clip_val, reference configs/an4.yaml:12)."""
import numpy as np
import pytest
import torch

from fixtures import Fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"
ULP = 1.2e-7


def _tensors(seed):
    rs = np.random.RandomState(seed)
    shapes = [(7,), (33, 5), (16384,), (16385,), (3, 41, 11), (50000,), (96, 40), (1,)]
    return [rs.standard_normal(s).astype(np.float32) for s in shapes]


def _close(a, b, ulps):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= ulps * ULP * max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("clip", [None, 0.75])
def test_fused_adamw_matches_torch_single_tensor(clip):
    from deepspeech.pytorch_amd.optim import FusedAdamW
    ws = _tensors(1)
    ref_p = [torch.nn.Parameter(torch.from_numpy(w.copy())) for w in ws]
    own_p = [torch.nn.Parameter(torch.from_numpy(w.copy()).to(DEV)) for w in ws]
    kw = dict(lr=1.5e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    ref = torch.optim.AdamW(ref_p, foreach=False, **kw)
    own = FusedAdamW(own_p, clip_grad_norm=clip, **kw)
    for step in range(3):
        gs = _tensors(10 + step)
        for p, q, g in zip(ref_p, own_p, gs):
            p.grad = torch.from_numpy(g.copy())
            q.grad = torch.from_numpy(g.copy()).to(DEV)
        if clip is not None:
            total = torch.nn.utils.clip_grad_norm_(ref_p, clip)
        ref.step()
        own.step()
        if clip is not None:
            assert abs(float(own.last_grad_norm[own_p[0].device]) - float(total)) <= 1e-5 * float(total)
        for i, (p, q) in enumerate(zip(ref_p, own_p)):
            assert _close(q.detach().cpu().numpy(), p.detach().numpy(), 8 * (step + 1)), (step, i)
            assert _close(own.state[q]["exp_avg"].cpu().numpy(), ref.state[p]["exp_avg"].numpy(), 8 * (step + 1))
            assert _close(own.state[q]["exp_avg_sq"].cpu().numpy(), ref.state[p]["exp_avg_sq"].numpy(), 8 * (step + 1))
    # same state_dict layout as torch.optim.AdamW (Lightning checkpoints it)
    assert set(own.state_dict()["state"][0]) == set(ref.state_dict()["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(own.state_dict()["state"][0]["step"]) == 3.0


def test_fused_sgd_nesterov_matches_torch():
    from deepspeech.pytorch_amd.optim import FusedSGD
    ws = _tensors(2)
    ref_p = [torch.nn.Parameter(torch.from_numpy(w.copy())) for w in ws]
    own_p = [torch.nn.Parameter(torch.from_numpy(w.copy()).to(DEV)) for w in ws]
    kw = dict(lr=3e-2, momentum=0.9, nesterov=True, weight_decay=1e-3)
    ref = torch.optim.SGD(ref_p, foreach=False, **kw)
    own = FusedSGD(own_p, clip_grad_norm=2.0, **kw)
    for step in range(3):
        gs = _tensors(20 + step)
        for p, q, g in zip(ref_p, own_p, gs):
            p.grad = torch.from_numpy(g.copy())
            q.grad = torch.from_numpy(g.copy()).to(DEV)
        torch.nn.utils.clip_grad_norm_(ref_p, 2.0)
        ref.step()
        own.step()
        for i, (p, q) in enumerate(zip(ref_p, own_p)):
            assert _close(q.detach().cpu().numpy(), p.detach().numpy(), 8 * (step + 1)), (step, i)
            assert _close(own.state[q]["momentum_buffer"].cpu().numpy(), ref.state[p]["momentum_buffer"].numpy(), 8 * (step + 1))


def test_opt_matrix_layouts_equal_the_cast_kernels():
    """ds2_opt_matrix: updated fp32 matrix + bf16 copy + bf16 transpose (with the rnns.0 column permutation and zero padding) are
    bit-identical to updating with ds2_opt_multi and re-laying out with ds2_cast_transpose_bf16."""
    import ctypes as C
    from deepspeech.pytorch_amd import ops
    from deepspeech.pytorch_amd._lib import call
    rs = np.random.RandomState(4)
    hp = (C.c_float * 7)(1 - 1e-3, 0.1, 0.999, 0.001, 0.0316, 1e-8, -0.015)
    for R, Cc, perm, Cout in ((96, 1312, (32, 41), 1344), (192, 64, None, 64), (48, 200, None, 200)):
        w, g = rs.standard_normal((R, Cc)).astype(np.float32), rs.standard_normal((R, Cc)).astype(np.float32)
        a = [torch.from_numpy(x.copy()).to(DEV) for x in (w, g, np.zeros_like(w), np.zeros_like(w))]
        b = [t.clone() for t in a]
        dst = torch.full((R, Cout), 7.0, dtype=torch.bfloat16, device=DEV)
        dstT = torch.full((Cout, R), 7.0, dtype=torch.bfloat16, device=DEV)
        pc, pf = perm if perm else (0, 0)
        call("ds2_opt_matrix", 0, ops.P(a[0]), ops.P(a[1]), ops.P(a[2]), ops.P(a[3]), R, Cc, pc, pf, Cout, ops.P(dst), Cout, ops.P(dstT), R,
             hp, 0, ops.P(None), ops.S())
        ptr = lambda t: (C.c_void_p * 1)(t.data_ptr())
        call("ds2_opt_multi", 0, 1, ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(b[3]), (C.c_long * 1)(R * Cc), hp, 0, ops.P(None), ops.S())
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        d2 = torch.empty_like(dst)
        d2T = torch.empty_like(dstT)
        ops.cast_transpose_bf16(b[0], d2, Cout, d2T, R, perm=perm, cout=Cout)
        assert torch.equal(dst, d2) and torch.equal(dstT, d2T)


@pytest.mark.parametrize("mode", [0, 1])
def test_opt_matrices_equals_one_call_per_matrix(mode):
    """ds2_opt_matrices (every un-permuted matrix of a group in one launch per 36, the permuted ones one by one) leaves the same
    bits as ds2_opt_matrix matrix by matrix: parameters, both optimizer states, the bf16 copy and the bf16 transpose.  40 matrices of
    mixed shapes (one launch is flushed in the middle), a column-permuted one and an odd-width one (C % 4 != 0) among them, some
    without a copy, some without a transpose."""
    import ctypes as C
    from deepspeech.pytorch_amd import ops
    from deepspeech.pytorch_amd._lib import call
    rs = np.random.RandomState(40 + mode)
    hp = (C.c_float * 7)(1 - 1e-3, 0.1, 0.999, 0.001, 0.0316, 1e-8, -0.015) if mode == 0 else (C.c_float * 7)(1e-3, 0.9, 0, 0, 1, 0, -0.02)
    shapes = [(96, 1312, (32, 41), 1344), (48, 202, None, 202)] + [(16 * int(rs.randint(1, 9)), 4 * int(rs.randint(1, 70)), None, None) for _ in range(38)]
    rs.shuffle(shapes)
    sets = []
    for k, (R, Cc, perm, Cout) in enumerate(shapes):
        Cout = Cout or Cc
        w, g = rs.standard_normal((R, Cc)).astype(np.float32), rs.standard_normal((R, Cc)).astype(np.float32)
        m0, v0 = rs.standard_normal((R, Cc)).astype(np.float32) * 0.1, rs.random_sample((R, Cc)).astype(np.float32) * 0.01
        a = [torch.from_numpy(x.copy()).to(DEV) for x in (w, g, m0, v0)]
        b = [t.clone() for t in a]
        lay = [None if k % 5 == 3 else torch.full((R, Cout), 7.0, dtype=torch.bfloat16, device=DEV),
               None if k % 7 == 2 else torch.full((Cout, R), 7.0, dtype=torch.bfloat16, device=DEV)]
        lay_b = [None if t is None else t.clone() for t in lay]
        sets.append((R, Cc, perm if perm else (0, 0), Cout, a, b, lay, lay_b))
    for R, Cc, (pc, pf), Cout, a, b, lay, lay_b in sets:
        call("ds2_opt_matrix", mode, ops.P(b[0]), ops.P(b[1]), ops.P(b[2]), ops.P(b[3]), R, Cc, pc, pf, Cout, ops.P(lay_b[0]), Cout,
             ops.P(lay_b[1]), R, hp, 0, ops.P(None), ops.S())
    parr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() if t is not None else 0 for t in ts])
    iarr = lambda vs: (C.c_int * len(vs))(*vs)
    larr = lambda vs: (C.c_long * len(vs))(*vs)
    call("ds2_opt_matrices", mode, len(sets), parr([s_[4][0] for s_ in sets]), parr([s_[4][1] for s_ in sets]), parr([s_[4][2] for s_ in sets]),
         parr([s_[4][3] for s_ in sets]), iarr([s_[0] for s_ in sets]), iarr([s_[1] for s_ in sets]), iarr([s_[2][0] for s_ in sets]),
         iarr([s_[2][1] for s_ in sets]), iarr([s_[3] for s_ in sets]), parr([s_[6][0] for s_ in sets]), larr([s_[3] for s_ in sets]),
         parr([s_[6][1] for s_ in sets]), larr([s_[0] for s_ in sets]), hp, 0, ops.P(None), ops.S())
    for k, (R, Cc, _, Cout, a, b, lay, lay_b) in enumerate(sets):
        for x, y in zip(a[:3] + ([a[3]] if mode == 0 else []), b[:3] + ([b[3]] if mode == 0 else [])):
            assert torch.equal(x, y), (k, R, Cc)
        for x, y in zip(lay, lay_b):
            assert x is None or torch.equal(x, y), (k, R, Cc)


def test_model_trains_identically_with_the_fused_optimizer():
    """Three training steps of the drop-in class (bf16 mode) with configure_optimizers()'s FusedAdamW (clip inside) against
    clip_grad_norm_ + torch.optim.AdamW: same loss trajectory; the bf16 weight layouts the optimizer leaves in the model's cache
    are the ones the cast kernels would produce from the updated weights."""
    from test_gpu_model import build
    from deepspeech.pytorch_amd import ops
    from deepspeech.pytorch_amd.optim import FusedAdamW
    fx = Fixture("gru_bi_mid")
    inputs, targets, pct, tsz = fx.batch()
    traj = {}
    for fused in (False, True):
        m = build(fx, "bf16").train()
        m.optim_cfg.learning_rate = 1e-2
        if fused:
            opt = m.configure_optimizers()[0][0]
            assert isinstance(opt, FusedAdamW)
            opt.clip_grad_norm = 50.0
        else:
            opt = torch.optim.AdamW(m.parameters(), lr=1e-2, betas=tuple(m.optim_cfg.betas), eps=m.optim_cfg.eps,
                                    weight_decay=m.optim_cfg.weight_decay)
        ls = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = m.training_step((torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()),
                                    torch.from_numpy(tsz)), 0)
            loss.backward()
            if not fused:
                torch.nn.utils.clip_grad_norm_(list(m.parameters()), 50.0)
            opt.step()
            ls.append(float(loss.item()))
        traj[fused] = ls
        if fused:
            c = m._cache
            for li, layer in enumerate(m.rnns):
                key = ("whhT", li, torch.bfloat16)
                assert key in c._store and c._store[key][0][0] == c.epoch + 1           # stamped for the next step
                got = c._store[key][1]
                want = torch.stack([getattr(layer.rnn, "weight_hh_l0" + s_).detach().t() for s_ in ("", "_reverse")], 0).to(torch.bfloat16)
                assert torch.equal(got, want.contiguous()), li
    assert abs(traj[True][0] - traj[False][0]) <= 1e-6 * abs(traj[False][0])
    assert abs(traj[False][2] - traj[False][0]) > 1e-2 * abs(traj[False][0])          # the loss moves
    for a, b in zip(traj[True], traj[False]):
        assert abs(a - b) <= 2e-3 * abs(b), (traj[True], traj[False])
