"""The oracle's building blocks against the ops the reference delegates to (torch on CPU, float64), one by one, at the call
sites' hyper-parameters: Conv2d (model.py:158,161), BatchNorm train / eval / running update (:159,162,86,196), Hardtanh(0, 20)
(:160,163,192), packed nn.GRU / nn.LSTM / nn.RNN(tanh) in both directions with gradients (:87-88,97-99), the Lookahead depthwise
conv (:115-130), log_softmax + CTCLoss(sum, zero_infinity) (:203,246-248).  The golden vectors pin the assembled step against the
reference class; these localise a disagreement to one restated op."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ds2_oracle as O

torch.set_default_dtype(torch.float32)
T64 = lambda a: torch.from_numpy(np.asarray(a, np.float64))  # noqa: E731


@pytest.mark.parametrize("geom", [((1, 8, 41, 11), (2, 2), (20, 5)), ((8, 8, 21, 11), (2, 1), (10, 5))])
def test_conv2d_forward_and_backward(geom):
    (cin, cout, kf, kt), stride, pad = geom
    rs = np.random.RandomState(0)
    x, w, b = rs.standard_normal((2, cin, 33, 17)), rs.standard_normal((cout, cin, kf, kt)) * 0.1, rs.standard_normal(cout)
    xt, wt, bt = T64(x).requires_grad_(), T64(w).requires_grad_(), T64(b).requires_grad_()
    yt = F.conv2d(xt, wt, bt, stride=stride, padding=pad)
    y = O.conv2d_fwd(x, w, b, stride, pad)
    assert np.abs(y - yt.detach().numpy()).max() < 1e-11
    dy = rs.standard_normal(y.shape)
    yt.backward(T64(dy))
    dx, dw, db = O.conv2d_bwd(x, w, dy, stride, pad)
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-10
    assert np.abs(dw - wt.grad.numpy()).max() < 1e-10
    assert np.abs(db - bt.grad.numpy()).max() < 1e-10


def test_batchnorm_train_eval_and_running_statistics():
    rs = np.random.RandomState(1)
    x = rs.standard_normal((3, 4, 5, 7)) * 2 + 1
    x[1, :, :, 4:] = 0                                           # masked frames are part of the statistics (model.py:61-68)
    g, b = rs.uniform(0.5, 1.5, 4), rs.standard_normal(4)
    bn = torch.nn.BatchNorm2d(4).double()
    with torch.no_grad():
        bn.weight.copy_(T64(g))
        bn.bias.copy_(T64(b))
    xt = T64(x).requires_grad_()
    yt = bn(xt)
    y, cache = O.bn_train_fwd(x, g, b, (0, 2, 3))
    assert np.abs(y - yt.detach().numpy()).max() < 1e-12
    rm, rv = O.bn_running_update(np.zeros(4), np.ones(4), cache)
    assert np.abs(rm - bn.running_mean.numpy()).max() < 1e-12 and np.abs(rv - bn.running_var.numpy()).max() < 1e-12
    dy = rs.standard_normal(y.shape)
    yt.backward(T64(dy))
    dx, dg, dbeta = O.bn_train_bwd(dy, g, cache)
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-11
    assert np.abs(dg - bn.weight.grad.numpy()).max() < 1e-11 and np.abs(dbeta - bn.bias.grad.numpy()).max() < 1e-11
    bn.eval()
    ye = O.bn_eval_fwd(x, g, b, rm, rv, 1)
    assert np.abs(ye - bn(T64(x)).detach().numpy()).max() < 1e-12
    # SequenceWise BatchNorm1d over T*N rows (model.py:18-33,86)
    rows = rs.standard_normal((11, 6))
    bn1 = torch.nn.BatchNorm1d(6).double()
    y1, c1 = O.bn_train_fwd(rows, np.ones(6), np.zeros(6), (0,))
    assert np.abs(y1 - bn1(T64(rows)).detach().numpy()).max() < 1e-12
    rm1, rv1 = O.bn_running_update(np.zeros(6), np.ones(6), c1)
    assert np.abs(rv1 - bn1.running_var.numpy()).max() < 1e-12


def test_hardtanh_value_and_strict_gradient_at_the_clamp_boundaries():
    x = np.array([-1.0, 0.0, 1e-12, 5.0, 20.0 - 1e-9, 20.0, 25.0])
    xt = T64(x).requires_grad_()
    yt = F.hardtanh(xt, 0.0, 20.0)
    assert np.array_equal(O.hardtanh_fwd(x), yt.detach().numpy())
    yt.backward(torch.ones_like(yt))
    assert np.array_equal(O.hardtanh_bwd(x, np.ones_like(x)), xt.grad.numpy())


@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
@pytest.mark.parametrize("bidirectional", [False, True])
def test_packed_recurrent_layer_forward_state_and_gradients(kind, bidirectional):
    rs = np.random.RandomState(2)
    T, N, I, H = 7, 4, 5, 6
    lens = np.array([7, 5, 5, 2])
    x = rs.standard_normal((T, N, I))
    for i, ln in enumerate(lens):
        x[ln:, i] = 0
    cls = {"gru": torch.nn.GRU, "lstm": torch.nn.LSTM, "rnn": torch.nn.RNN}[kind]
    m = cls(I, H, bidirectional=bidirectional, bias=True).double()
    xt = T64(x).requires_grad_()
    packed = torch.nn.utils.rnn.pack_padded_sequence(xt, torch.from_numpy(lens))
    out_p, hn = m(packed)
    out_t, _ = torch.nn.utils.rnn.pad_packed_sequence(out_p, total_length=T)
    dirs = [""] + (["_reverse"] if bidirectional else [])
    outs, caches = [], []
    for d, suf in enumerate(dirs):
        p = {k: getattr(m, k + "_l0" + suf).detach().numpy() for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
        o, h, c, cache = O.rnn_dir_fwd(kind, x, lens, p["weight_ih"], p["weight_hh"], p["bias_ih"], p["bias_hh"], reverse=(d == 1))
        outs.append(o)
        caches.append((cache, p))
        want_h = (hn[0] if kind == "lstm" else hn)[d].detach().numpy()
        assert np.abs(h - want_h).max() < 1e-12
        if kind == "lstm":
            assert np.abs(c - hn[1][d].detach().numpy()).max() < 1e-12
        assert np.abs(o - out_t.detach().numpy()[:, :, d * H:(d + 1) * H]).max() < 1e-12
        assert np.all(o[lens[3]:, 3] == 0)                                   # pad_packed_sequence: zeros past a clip's length
    dsum = rs.standard_normal((T, N, H))                                      # gradient of the direction SUM (model.py:101)
    y = out_t[:, :, :H] + (out_t[:, :, H:] if bidirectional else 0)
    y.backward(T64(dsum))
    dx = 0
    for d, suf in enumerate(dirs):
        cache, p = caches[d]
        dxd, dwi, dwh, dbi, dbh = O.rnn_dir_bwd(cache, dsum, p["weight_ih"], p["weight_hh"])
        dx = dx + dxd
        for got, name in ((dwi, "weight_ih"), (dwh, "weight_hh"), (dbi, "bias_ih"), (dbh, "bias_hh")):
            want = getattr(m, name + "_l0" + suf).grad.numpy()
            assert np.abs(got - want).max() < 1e-10 * max(1.0, np.abs(want).max()), (name, suf)
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-10


@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
@pytest.mark.parametrize("bidirectional", [False, True])
def test_packed_recurrent_layer_with_an_initial_state_and_its_gradients(kind, bidirectional):
    """`hs` of reference model.py:224-230 = hx of torch's recurrent layers on a packed sequence: forward, every parameter gradient
    (the first steps' share of dW_hh comes from h0) and the gradients with respect to h0 / c0 themselves."""
    rs = np.random.RandomState(5)
    T, N, I, H = 6, 4, 5, 6
    lens = np.array([6, 6, 4, 1])
    D = 2 if bidirectional else 1
    x = rs.standard_normal((T, N, I))
    for i, ln in enumerate(lens):
        x[ln:, i] = 0
    cls = {"gru": torch.nn.GRU, "lstm": torch.nn.LSTM, "rnn": torch.nn.RNN}[kind]
    m = cls(I, H, bidirectional=bidirectional, bias=True).double()
    h0 = T64(rs.standard_normal((D, N, H))).requires_grad_()
    c0 = T64(rs.standard_normal((D, N, H))).requires_grad_()
    xt = T64(x).requires_grad_()
    packed = torch.nn.utils.rnn.pack_padded_sequence(xt, torch.from_numpy(lens))
    out_p, hn = m(packed, (h0, c0) if kind == "lstm" else h0)
    out_t, _ = torch.nn.utils.rnn.pad_packed_sequence(out_p, total_length=T)
    dsum = rs.standard_normal((T, N, H))
    y = out_t[:, :, :H] + (out_t[:, :, H:] if bidirectional else 0)
    y.backward(T64(dsum))
    dx = 0
    for d, suf in enumerate([""] + (["_reverse"] if bidirectional else [])):
        p = {k: getattr(m, k + "_l0" + suf).detach().numpy() for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
        o, h, c, cache = O.rnn_dir_fwd(kind, x, lens, p["weight_ih"], p["weight_hh"], p["bias_ih"], p["bias_hh"], reverse=(d == 1),
                                       h0=h0[d].detach().numpy(), c0=c0[d].detach().numpy() if kind == "lstm" else None)
        assert np.abs(o - out_t.detach().numpy()[:, :, d * H:(d + 1) * H]).max() < 1e-12
        assert np.abs(h - (hn[0] if kind == "lstm" else hn)[d].detach().numpy()).max() < 1e-12
        dxd, dwi, dwh, dbi, dbh, dh0, dc0 = O.rnn_dir_bwd(cache, dsum, p["weight_ih"], p["weight_hh"], return_dstate=True)
        dx = dx + dxd
        for got, name in ((dwi, "weight_ih"), (dwh, "weight_hh"), (dbi, "bias_ih"), (dbh, "bias_hh")):
            want = getattr(m, name + "_l0" + suf).grad.numpy()
            assert np.abs(got - want).max() < 1e-10 * max(1.0, np.abs(want).max()), (name, suf)
        assert np.abs(dh0 - h0.grad[d].numpy()).max() < 1e-10
        if kind == "lstm":
            assert np.abs(dc0 - c0.grad[d].numpy()).max() < 1e-10
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-10


@pytest.mark.parametrize("ctx", [20, 3])
def test_lookahead_is_the_right_padded_depthwise_conv(ctx):
    rs = np.random.RandomState(3)
    T, N, H = 9, 2, 4
    x, w = rs.standard_normal((T, N, H)), rs.standard_normal((H, 1, ctx))
    xt, wt = T64(x).requires_grad_(), T64(w).requires_grad_()
    z = F.pad(xt.transpose(0, 1).transpose(1, 2), (0, ctx - 1), value=0)      # model.py:125-128
    yt = F.conv1d(z, wt, groups=H).transpose(1, 2).transpose(0, 1)
    y = O.lookahead_fwd(x, w)
    assert np.abs(y - yt.detach().numpy()).max() < 1e-12
    dy = rs.standard_normal(y.shape)
    yt.backward(T64(dy))
    dx, dw = O.lookahead_bwd(x, w, dy)
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-11 and np.abs(dw - wt.grad.numpy()).max() < 1e-11


def test_log_softmax_and_ctc_sum_with_zero_infinity():
    rs = np.random.RandomState(4)
    T, N, C = 12, 4, 29
    logits = rs.standard_normal((T, N, C)) * 1.5
    il, tl = np.array([12, 10, 9, 4]), np.array([4, 5, 1, 6])          # the last clip is infeasible (6 labels in 4 frames)
    targets = rs.randint(1, C, size=int(tl.sum()))
    targets[4:6] = 7                                                    # a repeated label (blank in between is mandatory)
    lt = T64(logits).requires_grad_()
    lp_t = F.log_softmax(lt, -1)
    loss_t = F.ctc_loss(lp_t, torch.from_numpy(targets), torch.from_numpy(il), torch.from_numpy(tl), blank=0, reduction="sum",
                        zero_infinity=True)
    loss_t.backward()
    want = float(loss_t.detach())
    lp = O.log_softmax(logits)
    assert np.abs(lp - lp_t.detach().numpy()).max() < 1e-12
    for fn in (O.ctc_loss_and_grad, O.ctc_loss_and_grad_fast):
        loss, nll, dlp = fn(lp, targets, il, tl, blank=0)
        assert abs(loss - want) < 1e-9 * abs(want) and nll[3] == 0
        dlogits = dlp - np.exp(lp) * dlp.sum(-1, keepdims=True)
        assert np.abs(dlogits - lt.grad.numpy()).max() < 1e-9, fn.__name__
