"""CPU ORACLE for the DeepSpeech2 train-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the algorithm of
``/root/reference/deepspeech_pytorch/model.py`` (class ``DeepSpeech``: forward + CTC
training step) including a hand-derived backward pass.  It exists to check the HIP
kernels; it is imported ONLY by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product path (``deepspeech.pytorch_amd``)
never imports it and has no CPU fallback.

Where the arithmetic lives: the reference is 100 % Python and delegates every op to a
third-party dependency that is not vendored under /root/reference -- PyTorch
(``requirements.txt:14`` ``torch``, unpinned; the container has torch 2.10.0).  Each
function below restates torch's published semantics for the op named at the reference
call site it cites.

Parity pin: the reference's own tests hold no golden vectors for this path
(``tests/smoke_test.py:78-80`` only asserts a checkpoint file exists).  The oracle is
therefore pinned against OUTPUTS OF THE REFERENCE ITSELF, generated in the build
container by ``tests/golden/make_golden.py`` (which imports the reference model.py
unmodified and runs it on CPU fp32) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_vs_golden.py`` checks the oracle against every fixture.

All arrays are numpy; ``dtype`` selects float32 or float64 arithmetic.
"""
import math

import numpy as np

BN_EPS = 1e-5          # torch.nn.BatchNorm{1,2}d default eps (model.py:159,162,86,196)
BN_MOMENTUM = 0.1      # torch default momentum
HT_MIN, HT_MAX = 0.0, 20.0   # nn.Hardtanh(0, 20) (model.py:160,163,192)
N_FREQ = 161           # floor(16000*0.02/2)+1 (model.py:166)
CONV1 = dict(k=(41, 11), s=(2, 2), p=(20, 5))   # model.py:158
CONV2 = dict(k=(21, 11), s=(2, 1), p=(10, 5))   # model.py:161
GATES = {"gru": 3, "lstm": 4, "rnn": 1}


# ----------------------------------------------------------------------------------------------
# length arithmetic
# ----------------------------------------------------------------------------------------------
def input_sizes_from_percentages(input_percentages, t_max):
    """model.py:243  ``input_percentages.mul_(int(inputs.size(3))).int()`` -- float32 multiply, truncation."""
    p = np.asarray(input_percentages, dtype=np.float32) * np.float32(int(t_max))
    return p.astype(np.int32)  # C cast truncates toward zero like Tensor.int()


def seq_lens(input_lengths):
    """model.py:299-310 get_seq_lens: both Conv2d modules, time axis (dim 1 of padding/kernel/stride)."""
    seq = np.asarray(input_lengths, dtype=np.int64)
    for c in (CONV1, CONV2):
        seq = (seq + 2 * c["p"][1] - 1 * (c["k"][1] - 1) - 1) // c["s"][1] + 1
    return seq.astype(np.int32)


def rnn_input_size():
    """model.py:166-169."""
    n = int(math.floor((16000 * 0.02) / 2) + 1)
    n = int(math.floor(n + 2 * 20 - 41) / 2 + 1)
    n = int(math.floor(n + 2 * 10 - 21) / 2 + 1)
    return n * 32


# ----------------------------------------------------------------------------------------------
# conv stack (MaskConv, model.py:53-69 over the Sequential at model.py:157-164)
# ----------------------------------------------------------------------------------------------
def _out_size(n, k, s, p):
    return (n + 2 * p - (k - 1) - 1) // s + 1


def _padded(x, pad):
    n, c, f, t = x.shape
    xp = np.zeros((n, c, f + 2 * pad[0], t + 2 * pad[1]), dtype=x.dtype)
    xp[:, :, pad[0]:pad[0] + f, pad[1]:pad[1] + t] = x
    return xp


# numpy has no convolution, and a product per tap over gathered operands spends its time in the gathers (minutes on the full-size
# fixtures).  Evaluation order used here, per sample: unfold the KERNEL ROWS only -- row (c, a) of U holds xpad[c, sf*f + a, :] over
# the whole padded time axis -- so that all KT time taps are ONE matrix product  Z[(bb, o)][(f, t')] = W[(bb, o)][(c, a)] U[(c, a)][(f, t')],
# and y[o, f, t] = sum_bb Z[bb, o, f, st*t + bb]: the time offset of a tap is applied to its product, not to its operand.
def _unfold_rows(xpi, kf, sf, fo):
    """[C][F_padded][T_padded] -> [(C*KF)][FO*T_padded]."""
    c, _, tp = xpi.shape
    win = np.lib.stride_tricks.sliding_window_view(xpi, kf, axis=1)[:, ::sf][:, :fo]      # [C][FO][T_padded][KF]
    return np.ascontiguousarray(win.transpose(0, 3, 1, 2)).reshape(c * kf, fo * tp)


def _taps_as_rows(w):
    """w (O,C,KF,KT) -> [(KT*O)][(C*KF)]: block bb = time tap bb of every output channel."""
    o, c, kf, kt = w.shape
    return np.ascontiguousarray(w.transpose(3, 0, 1, 2)).reshape(kt * o, c * kf)


def conv2d_fwd(x, w, b, stride, pad):
    """torch.nn.Conv2d cross-correlation (model.py:158,161). x (N,C,F,T), w (O,C,KF,KT), b (O):
    y[n,o,f,t] = b[o] + sum_{c,a,bb} w[o,c,a,bb] * xpad[n,c,sf*f+a,st*t+bb]."""
    n, c, f, t = x.shape
    o, _, kf, kt = w.shape
    (sf, st), tp = stride, t + 2 * pad[1]
    fo, to = _out_size(f, kf, sf, pad[0]), _out_size(t, kt, st, pad[1])
    xp = _padded(x, pad)
    wr = _taps_as_rows(w)
    y = np.zeros((n, o, fo, to), dtype=x.dtype)
    for i in range(n):
        z = (wr @ _unfold_rows(xp[i], kf, sf, fo)).reshape(kt, o, fo, tp)
        for bb in range(kt):
            y[i] += z[bb, :, :, bb:bb + st * to:st]
    return y + b[None, :, None, None]


def conv2d_bwd(x, w, dy, stride, pad, need_dx=True):
    n, c, f, t = x.shape
    o, _, kf, kt = w.shape
    (sf, st), tp = stride, t + 2 * pad[1]
    fo, to = dy.shape[2], dy.shape[3]
    xp = _padded(x, pad)
    wr = _taps_as_rows(w)
    dwr = np.zeros_like(wr)
    dxp = np.zeros_like(xp) if need_dx else None
    for i in range(n):
        dsh = np.zeros((kt, o, fo, tp), dtype=dy.dtype)                # dy[i] placed where time tap bb reads: position st*t + bb
        for bb in range(kt):
            dsh[bb, :, :, bb:bb + st * to:st] = dy[i]
        dsh = dsh.reshape(kt * o, fo * tp)
        dwr += dsh @ _unfold_rows(xp[i], kf, sf, fo).T
        if need_dx:
            du = (wr.T @ dsh).reshape(c, kf, fo, tp)                   # gradient of the unfolded rows, folded back row by row
            for a in range(kf):
                dxp[i, :, a:a + sf * fo:sf, :] += du[:, a]
    dw = dwr.reshape(kt, o, c, kf).transpose(1, 2, 3, 0)
    db = dy.sum(axis=(0, 2, 3))
    dx = dxp[:, :, pad[0]:pad[0] + f, pad[1]:pad[1] + t] if need_dx else None
    return dx, np.ascontiguousarray(dw), db


def time_mask(shape, lens):
    """model.py:61-68: True (== zero it) where t >= length_i, on the last axis of (N,C,F,T)."""
    m = np.zeros(shape, dtype=bool)
    for i, ln in enumerate(lens):
        if shape[3] - int(ln) > 0:
            m[i, :, :, int(ln):] = True
    return m


def bn_train_fwd(x, gamma, beta, axes, eps=BN_EPS):
    """torch batch_norm in training mode: batch mean / BIASED variance over `axes` (stats include every
    element, also the zeros the mask wrote -- model.py:61-68 masks BEFORE the next module sees x)."""
    mean = x.mean(axis=axes, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=axes, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    shp = [1] * x.ndim
    ch = [d for d in range(x.ndim) if d not in axes][0]
    shp[ch] = -1
    y = xhat * gamma.reshape(shp) + beta.reshape(shp)
    cnt = x.size // x.shape[ch]
    return y, dict(xhat=xhat, rstd=rstd, mean=mean.reshape(-1), var=var.reshape(-1), cnt=cnt, shp=shp, axes=axes)


def bn_eval_fwd(x, gamma, beta, rmean, rvar, ch_axis, eps=BN_EPS):
    shp = [1] * x.ndim
    shp[ch_axis] = -1
    return (x - rmean.reshape(shp)) / np.sqrt(rvar.reshape(shp) + eps) * gamma.reshape(shp) + beta.reshape(shp)


def bn_running_update(rmean, rvar, cache, momentum=BN_MOMENTUM):
    """running_var uses the UNBIASED batch variance (torch semantics)."""
    cnt = cache["cnt"]
    unb = cache["var"] * (cnt / max(cnt - 1, 1))
    return (1 - momentum) * rmean + momentum * cache["mean"], (1 - momentum) * rvar + momentum * unb


def bn_train_bwd(dy, gamma, cache):
    axes, shp, xhat, rstd = cache["axes"], cache["shp"], cache["xhat"], cache["rstd"]
    dgamma = (dy * xhat).sum(axis=axes)
    dbeta = dy.sum(axis=axes)
    m = cache["cnt"]
    dx = gamma.reshape(shp) * rstd * (dy - dbeta.reshape(shp) / m - xhat * dgamma.reshape(shp) / m)
    return dx, dgamma, dbeta


def hardtanh_fwd(x):
    return np.clip(x, HT_MIN, HT_MAX)


def hardtanh_bwd(x_in, dy):
    """torch hardtanh_backward: grad passes where min < x < max (strict)."""
    return dy * ((x_in > HT_MIN) & (x_in < HT_MAX))


def conv_stack_fwd(P, x, out_lens, train=True):
    """MaskConv.forward (model.py:53-69): after EACH of the 6 modules, zero t >= out_len_i."""
    c = {}
    y = conv2d_fwd(x, P["conv.seq_module.0.weight"], P["conv.seq_module.0.bias"], CONV1["s"], CONV1["p"])
    m1 = time_mask(y.shape, out_lens)
    y[m1] = 0
    c["x"], c["m1"], c["y1"] = x, m1, y
    if train:
        z, c["bn1"] = bn_train_fwd(y, P["conv.seq_module.1.weight"], P["conv.seq_module.1.bias"], (0, 2, 3))
    else:
        z = bn_eval_fwd(y, P["conv.seq_module.1.weight"], P["conv.seq_module.1.bias"],
                        P["conv.seq_module.1.running_mean"], P["conv.seq_module.1.running_var"], 1)
    z[m1] = 0
    c["z1"] = z
    a = hardtanh_fwd(z)
    a[m1] = 0
    c["a1"] = a
    y = conv2d_fwd(a, P["conv.seq_module.3.weight"], P["conv.seq_module.3.bias"], CONV2["s"], CONV2["p"])
    m2 = time_mask(y.shape, out_lens)
    y[m2] = 0
    c["m2"], c["y2"] = m2, y
    if train:
        z, c["bn2"] = bn_train_fwd(y, P["conv.seq_module.4.weight"], P["conv.seq_module.4.bias"], (0, 2, 3))
    else:
        z = bn_eval_fwd(y, P["conv.seq_module.4.weight"], P["conv.seq_module.4.bias"],
                        P["conv.seq_module.4.running_mean"], P["conv.seq_module.4.running_var"], 1)
    z[m2] = 0
    c["z2"] = z
    a = hardtanh_fwd(z)
    a[m2] = 0
    return a, c


def conv_stack_bwd(P, c, da2):
    G = {}
    d = da2.copy()
    d[c["m2"]] = 0
    d = hardtanh_bwd(c["z2"], d)
    d[c["m2"]] = 0
    d, G["conv.seq_module.4.weight"], G["conv.seq_module.4.bias"] = bn_train_bwd(d, P["conv.seq_module.4.weight"], c["bn2"])
    d[c["m2"]] = 0
    d, G["conv.seq_module.3.weight"], G["conv.seq_module.3.bias"] = conv2d_bwd(
        c["a1"], P["conv.seq_module.3.weight"], d, CONV2["s"], CONV2["p"])
    d[c["m1"]] = 0
    d = hardtanh_bwd(c["z1"], d)
    d[c["m1"]] = 0
    d, G["conv.seq_module.1.weight"], G["conv.seq_module.1.bias"] = bn_train_bwd(d, P["conv.seq_module.1.weight"], c["bn1"])
    d[c["m1"]] = 0
    _, G["conv.seq_module.0.weight"], G["conv.seq_module.0.bias"] = conv2d_bwd(
        c["x"], P["conv.seq_module.0.weight"], d, CONV1["s"], CONV1["p"], need_dx=False)
    return G


# ----------------------------------------------------------------------------------------------
# recurrent layers (BatchRNN, model.py:80-102): packed-sequence semantics restated with masks
# ----------------------------------------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def rnn_dir_fwd(kind, x, lens, w_ih, w_hh, b_ih, b_hh, reverse, h0=None, c0=None):
    """One direction of nn.GRU / nn.LSTM / nn.RNN(tanh) on a packed sequence (model.py:97-99).

    pack_padded_sequence semantics: sample i takes part only at t < lens[i]; the reverse direction therefore
    starts at each sample's OWN last frame; pad_packed_sequence returns zeros at t >= lens[i]; h_n is the
    state after each sample's last valid step.  torch gate order: GRU r,z,n ; LSTM i,f,g,o.
    """
    T, N, _ = x.shape
    H = w_hh.shape[1]
    dt = x.dtype
    h = np.zeros((N, H), dt) if h0 is None else h0.astype(dt).copy()
    cst = np.zeros((N, H), dt) if c0 is None else c0.astype(dt).copy()
    gi_all = x.reshape(T * N, -1) @ w_ih.T + b_ih
    gi_all = gi_all.reshape(T, N, -1)
    out = np.zeros((T, N, H), dt)
    steps = []
    order = range(T - 1, -1, -1) if reverse else range(T)
    lens = np.asarray(lens)
    for t in order:
        act = (t < lens)[:, None]
        gi = gi_all[t]
        gh = h @ w_hh.T + b_hh
        st = dict(t=t, act=act, hprev=h.copy())
        if kind == "gru":
            r = _sigmoid(gi[:, :H] + gh[:, :H])
            z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            hn = gh[:, 2 * H:]
            n = np.tanh(gi[:, 2 * H:] + r * hn)
            hnew = (1 - z) * n + z * h
            st.update(r=r, z=z, n=n, hn=hn)
        elif kind == "lstm":
            s = gi + gh
            ig, fg = _sigmoid(s[:, :H]), _sigmoid(s[:, H:2 * H])
            gg, og = np.tanh(s[:, 2 * H:3 * H]), _sigmoid(s[:, 3 * H:])
            cnew = fg * cst + ig * gg
            tc = np.tanh(cnew)
            hnew = og * tc
            st.update(i=ig, f=fg, g=gg, o=og, cprev=cst.copy(), tc=tc)
            cst = np.where(act, cnew, cst)
        else:  # rnn tanh
            hnew = np.tanh(gi + gh)
            st.update(hnew=hnew)
        h = np.where(act, hnew, h)
        out[t] = np.where(act, hnew, 0)
        steps.append(st)
    return out, h, cst, dict(kind=kind, steps=steps, x=x, H=H)


def rnn_dir_bwd(cache, dout, w_ih, w_hh, return_dstate=False):
    """BPTT for one direction. Returns dx, dw_ih, dw_hh, db_ih, db_hh (+ dh0, dc0 with return_dstate: the gradients with respect to
    the initial state rnn_dir_fwd was given -- `hs` of reference model.py:224-230, i.e. hx of torch's nn.GRU / nn.LSTM / nn.RNN)."""
    kind, steps, x, H = cache["kind"], cache["steps"], cache["x"], cache["H"]
    T, N, I = x.shape
    dt = x.dtype
    G = GATES[kind]
    dgi_all = np.zeros((T, N, G * H), dt)
    dw_hh = np.zeros_like(w_hh)
    db_hh = np.zeros(G * H, dt)
    dh = np.zeros((N, H), dt)
    dc = np.zeros((N, H), dt)
    for st in reversed(steps):
        t, act, hprev = st["t"], st["act"], st["hprev"]
        dhn = np.where(act, dout[t] + dh, 0)   # grad wrt hnew for active samples
        dh_pass = np.where(act, 0, dh)          # inactive: state passes through unchanged
        if kind == "gru":
            r, z, n, hn = st["r"], st["z"], st["n"], st["hn"]
            dn = dhn * (1 - z) * (1 - n * n)
            dz = dhn * (hprev - n) * z * (1 - z)
            dr = dn * hn * r * (1 - r)
            dgi = np.concatenate([dr, dz, dn], axis=1)
            dgh = np.concatenate([dr, dz, dn * r], axis=1)
            dh = dhn * z + dgh @ w_hh + dh_pass
        elif kind == "lstm":
            ig, fg, gg, og, cprev, tc = st["i"], st["f"], st["g"], st["o"], st["cprev"], st["tc"]
            dcn = np.where(act, dc + dhn * og * (1 - tc * tc), 0)
            dc_pass = np.where(act, 0, dc)
            di = dcn * gg * ig * (1 - ig)
            df = dcn * cprev * fg * (1 - fg)
            dg = dcn * ig * (1 - gg * gg)
            do = dhn * tc * og * (1 - og)
            dgi = np.concatenate([di, df, dg, do], axis=1)
            dgh = dgi
            dh = dgh @ w_hh + dh_pass
            dc = dcn * fg + dc_pass
        else:
            hnew = st["hnew"]
            dgi = dhn * (1 - hnew * hnew)
            dgh = dgi
            dh = dgh @ w_hh + dh_pass
        dgi_all[t] = dgi
        dw_hh += dgh.T @ hprev
        db_hh += dgh.sum(0)
    flat = dgi_all.reshape(T * N, -1)
    dw_ih = flat.T @ x.reshape(T * N, I)
    db_ih = flat.sum(0)
    dx = (flat @ w_ih).reshape(T, N, I)
    if return_dstate:
        return dx, dw_ih, dw_hh, db_ih, db_hh, dh, dc
    return dx, dw_ih, dw_hh, db_ih, db_hh


def batch_rnn_fwd(P, prefix, kind, x, lens, bidirectional, batch_norm, train=True, h0=None):
    """BatchRNN.forward (model.py:94-102): [SequenceWise BN over T*N rows incl. zero pad rows] -> packed RNN ->
    pad -> sum of directions."""
    c = dict(kind=kind, bidirectional=bidirectional, batch_norm=batch_norm)
    T, N, _ = x.shape
    if batch_norm:
        bnp = prefix + ".batch_norm.module."
        flat = x.reshape(T * N, -1)
        if train:
            flat, c["bn"] = bn_train_fwd(flat, P[bnp + "weight"], P[bnp + "bias"], (0,))
        else:
            flat = bn_eval_fwd(flat, P[bnp + "weight"], P[bnp + "bias"], P[bnp + "running_mean"], P[bnp + "running_var"], 1)
        x = flat.reshape(T, N, -1)
    rp = prefix + ".rnn."
    dirs = [""] + (["_reverse"] if bidirectional else [])
    outs, hs, cs, c["dirs"] = [], [], [], []
    for d, suf in enumerate(dirs):
        hh0 = cc0 = None
        if h0 is not None:
            if kind == "lstm":
                hh0, cc0 = h0[0][d], h0[1][d]
            else:
                hh0 = h0[d]
        o, h, cst, dc = rnn_dir_fwd(kind, x, lens, P[rp + "weight_ih_l0" + suf], P[rp + "weight_hh_l0" + suf],
                                    P[rp + "bias_ih_l0" + suf], P[rp + "bias_hh_l0" + suf], reverse=(d == 1),
                                    h0=hh0, c0=cc0)
        outs.append(o)
        hs.append(h)
        cs.append(cst)
        c["dirs"].append(dc)
    out = outs[0] + outs[1] if bidirectional else outs[0]   # model.py:101 sum of directions
    hn = (np.stack(hs), np.stack(cs)) if kind == "lstm" else np.stack(hs)
    return out, hn, c


def batch_rnn_bwd(P, prefix, c, dout):
    G = {}
    rp = prefix + ".rnn."
    dirs = [""] + (["_reverse"] if c["bidirectional"] else [])
    dx = None
    for d, suf in enumerate(dirs):
        dxd, dwi, dwh, dbi, dbh = rnn_dir_bwd(c["dirs"][d], dout, P[rp + "weight_ih_l0" + suf], P[rp + "weight_hh_l0" + suf])
        G[rp + "weight_ih_l0" + suf], G[rp + "weight_hh_l0" + suf] = dwi, dwh
        G[rp + "bias_ih_l0" + suf], G[rp + "bias_hh_l0" + suf] = dbi, dbh
        dx = dxd if dx is None else dx + dxd
    if c["batch_norm"]:
        bnp = prefix + ".batch_norm.module."
        T, N, I = dx.shape
        flat, G[bnp + "weight"], G[bnp + "bias"] = bn_train_bwd(dx.reshape(T * N, I), P[bnp + "weight"], c["bn"])
        dx = flat.reshape(T, N, I)
    return dx, G


# ----------------------------------------------------------------------------------------------
# lookahead (model.py:105-135) -- uni-directional models only (model.py:189-193, 232-233)
# ----------------------------------------------------------------------------------------------
def lookahead_fwd(x, w):
    """x (T,N,H); w = conv.weight (H,1,ctx). Right-pad ctx-1 zeros, depthwise conv over time."""
    T, N, H = x.shape
    ctx = w.shape[2]
    xp = np.concatenate([x, np.zeros((ctx - 1, N, H), x.dtype)], axis=0)
    y = np.zeros_like(x)
    for k in range(ctx):
        y += xp[k:k + T] * w[:, 0, k][None, None, :]
    return y


def lookahead_bwd(x, w, dy):
    T, N, H = x.shape
    ctx = w.shape[2]
    xp = np.concatenate([x, np.zeros((ctx - 1, N, H), x.dtype)], axis=0)
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for k in range(ctx):
        dw[:, 0, k] = (xp[k:k + T] * dy).sum(axis=(0, 1))
        dxp[k:k + T] += dy * w[:, 0, k][None, None, :]
    return dxp[:T], dw


# ----------------------------------------------------------------------------------------------
# CTC (model.py:203,248: CTCLoss(blank=0, reduction='sum', zero_infinity=True)) + log_softmax (model.py:246)
# ----------------------------------------------------------------------------------------------
def log_softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = x - m
    return e - np.log(np.exp(e).sum(axis=-1, keepdims=True))


def _lse(*vals):
    m = max(vals)
    if m == -np.inf:
        return -np.inf
    return m + math.log(sum(math.exp(v - m) for v in vals))


def ctc_loss_and_grad(log_probs, targets, input_lengths, target_lengths, blank=0, zero_infinity=True):
    """torch.nn.functional.ctc_loss (native _ctc_loss, log-space alpha/beta recursion over the extended label
    sequence l' = blank,l1,blank,...,blank of 2S+1 states), reduction='sum'.
    log_probs (T,N,C). targets 1-D concatenated. Returns (sum loss, per-sample nll, d(sum loss)/d(log_probs))."""
    T, N, C = log_probs.shape
    lp_all = log_probs.astype(np.float64)
    grad = np.zeros((T, N, C), np.float64)
    nll = np.zeros(N, np.float64)
    off = 0
    NEG = -np.inf
    for i in range(N):
        S = int(target_lengths[i])
        Ti = int(input_lengths[i])
        tg = [int(v) for v in targets[off:off + S]]
        off += S
        ext = [blank] * (2 * S + 1)
        ext[1::2] = tg
        L = len(ext)
        lp = lp_all[:, i, :]
        la = np.full((Ti, L), NEG)
        la[0, 0] = lp[0, blank]
        if L > 1:
            la[0, 1] = lp[0, ext[1]]
        for t in range(1, Ti):
            for s in range(L):
                v = [la[t - 1, s]]
                if s >= 1:
                    v.append(la[t - 1, s - 1])
                if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                    v.append(la[t - 1, s - 2])
                la[t, s] = _lse(*v) + lp[t, ext[s]]
        ll = _lse(la[Ti - 1, L - 1], la[Ti - 1, L - 2]) if L > 1 else la[Ti - 1, L - 1]
        nll[i] = -ll
        if ll == NEG:
            if zero_infinity:
                nll[i] = 0.0   # loss and grad -> 0 for infeasible samples
                continue
            else:
                grad[:, i, :] = np.nan
                continue
        lb = np.full((Ti, L), NEG)
        lb[Ti - 1, L - 1] = lp[Ti - 1, ext[L - 1]]
        if L > 1:
            lb[Ti - 1, L - 2] = lp[Ti - 1, ext[L - 2]]
        for t in range(Ti - 2, -1, -1):
            for s in range(L):
                v = [lb[t + 1, s]]
                if s + 1 < L:
                    v.append(lb[t + 1, s + 1])
                if s + 2 < L and ext[s + 2] != blank and ext[s + 2] != ext[s]:
                    v.append(lb[t + 1, s + 2])
                lb[t, s] = _lse(*v) + lp[t, ext[s]]
        ab = la + lb
        for t in range(Ti):
            for cidx in set(ext):
                vals = [ab[t, s] for s in range(L) if ext[s] == cidx]
                lcab = _lse(*vals)
                if lcab != NEG:
                    grad[t, i, cidx] = -math.exp(lcab - lp[t, cidx] + nll[i])
    return float(nll.sum()), nll, grad.astype(log_probs.dtype)


def ctc_loss_and_grad_fast(log_probs, targets, input_lengths, target_lengths, blank=0):
    """Vectorised (over states) version of ctc_loss_and_grad with zero_infinity=True; same results, used for the
    larger parity sizes and the cpu_baseline timing.  Pure-python loop over t only."""
    T, N, C = log_probs.shape
    grad = np.zeros((T, N, C), np.float64)
    nll = np.zeros(N, np.float64)
    off = 0
    NEG = -np.inf
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for i in range(N):
            S = int(target_lengths[i])
            Ti = int(input_lengths[i])
            tg = np.asarray(targets[off:off + S], dtype=np.int64)
            off += S
            L = 2 * S + 1
            ext = np.zeros(L, np.int64) + blank
            ext[1::2] = tg
            skip = np.zeros(L, bool)
            skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
            lp = log_probs[:Ti, i, :].astype(np.float64)[:, ext]      # (Ti, L)

            def lse3(a, b, c):
                m = np.maximum(np.maximum(a, b), c)
                ms = np.where(np.isfinite(m), m, 0.0)
                return np.where(np.isfinite(m), ms + np.log(np.exp(a - ms) + np.exp(b - ms) + np.exp(c - ms)), NEG)

            la = np.full((Ti, L), NEG)
            la[0, 0] = lp[0, 0]
            if L > 1:
                la[0, 1] = lp[0, 1]
            for t in range(1, Ti):
                p = la[t - 1]
                p1 = np.concatenate([[NEG], p[:-1]])
                p2 = np.where(skip, np.concatenate([[NEG, NEG], p[:-2]]), NEG)
                la[t] = lse3(p, p1, p2) + lp[t]
            ll = np.logaddexp(la[Ti - 1, L - 1], la[Ti - 1, L - 2]) if L > 1 else la[Ti - 1, L - 1]
            if not np.isfinite(ll):
                continue
            nll[i] = -ll
            lb = np.full((Ti, L), NEG)
            lb[Ti - 1, L - 1] = lp[Ti - 1, L - 1]
            if L > 1:
                lb[Ti - 1, L - 2] = lp[Ti - 1, L - 2]
            skipn = np.zeros(L, bool)
            skipn[:-2] = skip[2:]
            for t in range(Ti - 2, -1, -1):
                p = lb[t + 1]
                p1 = np.concatenate([p[1:], [NEG]])
                p2 = np.where(skipn, np.concatenate([p[2:], [NEG, NEG]]), NEG)
                lb[t] = lse3(p, p1, p2) + lp[t]
            ab = la + lb                                             # (Ti, L)
            m = ab.max(axis=1, keepdims=True)
            e = np.exp(ab - m)
            acc = np.zeros((Ti, C))
            np.add.at(acc, (np.arange(Ti)[:, None].repeat(L, 1), ext[None, :].repeat(Ti, 0)), e)
            lpc = log_probs[:Ti, i, :].astype(np.float64)
            g = -np.exp(np.log(acc) + m + nll[i] - lpc)
            g[acc == 0] = 0
            grad[:Ti, i, :] = g
    return float(nll.sum()), nll, grad.astype(log_probs.dtype)


# ----------------------------------------------------------------------------------------------
# whole model (DeepSpeech.forward model.py:214-239, training_step model.py:241-249)
# ----------------------------------------------------------------------------------------------
def model_forward(P, cfg, x, lengths, train=True, hs=None, keep_cache=True):
    """cfg: dict(rnn_type in {'gru','lstm','rnn'}, hidden_size, hidden_layers, bidirectional, lookahead_context).
    Returns (out (N,T',C), output_lengths int32, new_hs, cache)."""
    kind, L, bi = cfg["rnn_type"], cfg["hidden_layers"], cfg["bidirectional"]
    out_lens = seq_lens(lengths)                       # model.py:215-216
    a2, cc = conv_stack_fwd(P, x, out_lens, train)     # model.py:217
    n, ch, f, t = a2.shape
    xr = a2.reshape(n, ch * f, t).transpose(2, 0, 1).copy()   # model.py:219-221: feature index = c*41+f, (T,N,H)
    cache = dict(conv=cc, a2_shape=a2.shape, rnn=[], out_lens=out_lens)
    new_hs = []
    for l in range(L):
        h0 = None if hs is None else hs[l]
        xr, hn, rc = batch_rnn_fwd(P, "rnns.%d" % l, kind, xr, out_lens, bi, batch_norm=(l > 0), train=train, h0=h0)
        cache["rnn"].append(rc)
        new_hs.append(hn)
    if not bi:                                          # model.py:232-233
        cache["la_in"] = xr
        y = lookahead_fwd(xr, P["lookahead.0.conv.weight"])
        cache["la_pre"] = y
        xr = hardtanh_fwd(y)
    T, N, H = xr.shape
    flat = xr.reshape(T * N, H)                         # model.py:235 SequenceWise(BN -> Linear no bias)
    if train:
        flat, cache["fc_bn"] = bn_train_fwd(flat, P["fc.0.module.0.weight"], P["fc.0.module.0.bias"], (0,))
    else:
        flat = bn_eval_fwd(flat, P["fc.0.module.0.weight"], P["fc.0.module.0.bias"],
                           P["fc.0.module.0.running_mean"], P["fc.0.module.0.running_var"], 1)
    cache["fc_in"] = flat
    logits = (flat @ P["fc.0.module.1.weight"].T).reshape(T, N, -1)
    out = logits.transpose(1, 0, 2)                     # model.py:236
    if not train:                                       # model.py:238, 72-77
        e = np.exp(out - out.max(-1, keepdims=True))
        out = e / e.sum(-1, keepdims=True)
    cache["logits_tnc"] = logits
    return out, out_lens, new_hs, (cache if keep_cache else None)


def model_backward(P, cfg, cache, dlogits_tnc):
    """Backward from d(loss)/d(logits (T,N,C)) to every parameter. Returns dict keyed by state_dict names."""
    G = {}
    T, N, C = dlogits_tnc.shape
    dflat = dlogits_tnc.reshape(T * N, C)
    G["fc.0.module.1.weight"] = dflat.T @ cache["fc_in"]
    d = dflat @ P["fc.0.module.1.weight"]
    d, G["fc.0.module.0.weight"], G["fc.0.module.0.bias"] = bn_train_bwd(d, P["fc.0.module.0.weight"], cache["fc_bn"])
    d = d.reshape(T, N, -1)
    if not cfg["bidirectional"]:
        d = hardtanh_bwd(cache["la_pre"], d)
        d, G["lookahead.0.conv.weight"] = lookahead_bwd(cache["la_in"], P["lookahead.0.conv.weight"], d)
    for l in range(cfg["hidden_layers"] - 1, -1, -1):
        d, g = batch_rnn_bwd(P, "rnns.%d" % l, cache["rnn"][l], d)
        G.update(g)
    n, ch, f, t = cache["a2_shape"]
    da2 = d.transpose(1, 2, 0).reshape(n, ch, f, t)
    G.update(conv_stack_bwd(P, cache["conv"], da2))
    return G


def running_stats_after_step(P, cfg, cache):
    """New values of every BatchNorm running_mean / running_var after one training forward."""
    R = {}
    for idx, key in ((1, "bn1"), (4, "bn2")):
        p = "conv.seq_module.%d." % idx
        R[p + "running_mean"], R[p + "running_var"] = bn_running_update(P[p + "running_mean"], P[p + "running_var"], cache["conv"][key])
    for l in range(1, cfg["hidden_layers"]):
        p = "rnns.%d.batch_norm.module." % l
        R[p + "running_mean"], R[p + "running_var"] = bn_running_update(P[p + "running_mean"], P[p + "running_var"], cache["rnn"][l]["bn"])
    p = "fc.0.module.0."
    R[p + "running_mean"], R[p + "running_var"] = bn_running_update(P[p + "running_mean"], P[p + "running_var"], cache["fc_bn"])
    return R


def training_step(P, cfg, inputs, targets, input_percentages, target_sizes, fast_ctc=True):
    """DeepSpeech.training_step (model.py:241-249) + autograd backward. Returns dict(loss, logits (N,T',C),
    output_lengths, grads, running)."""
    input_sizes = input_sizes_from_percentages(input_percentages, inputs.shape[3])
    out, out_lens, _, cache = model_forward(P, cfg, inputs, input_sizes, train=True)
    logits_tnc = cache["logits_tnc"]
    lp = log_softmax(logits_tnc)
    fn = ctc_loss_and_grad_fast if fast_ctc else ctc_loss_and_grad
    loss, nll, dlp = fn(lp, targets, out_lens, target_sizes, blank=0)
    # log_softmax backward: dlogit = g - softmax * sum_c g
    dlogits = dlp - np.exp(lp) * dlp.sum(-1, keepdims=True)
    G = model_backward(P, cfg, cache, dlogits.astype(inputs.dtype))
    return dict(loss=loss, nll=nll, logits=out, output_lengths=out_lens, grads=G,
                running=running_stats_after_step(P, cfg, cache), log_probs=lp, dlogits=dlogits)


def greedy_decode(out, sizes, labels, blank=0):
    """GreedyDecoder.decode (decoder.py:164-181): argmax, collapse repeats, drop blanks."""
    res = []
    idx = out.argmax(-1)
    for i in range(out.shape[0]):
        s, prev = "", None
        for t in range(int(sizes[i])):
            k = int(idx[i, t])
            if k != blank and not (t != 0 and k == prev):
                s += labels[k]
            prev = k
        res.append(s)
    return res


# ==================================================================================================================
# spectrogram front-end (reference loader/data_loader.py:73-94)
# ==================================================================================================================
def spect_window(name="hamming", n=320):
    """Periodic window (scipy.signal.get_window(name, n, fftbins=True), what librosa.stft uses for a window NAME)."""
    k = np.arange(n, dtype=np.float64)
    if name == "hamming":
        return 0.54 - 0.46 * np.cos(2 * np.pi * k / n)
    if name == "hann":
        return 0.5 - 0.5 * np.cos(2 * np.pi * k / n)
    raise ValueError(name)


def log_spectrogram(y, sample_rate=16000, window_size=0.02, window_stride=0.01, window="hamming", normalize=True,
                    pad_mode="constant"):
    """compute_spectrogram (data_loader.py:73-94).  The STFT is librosa's (third-party, NOT vendored under /root/reference
    and unpinned in requirements.txt:4 -- "parity unpinned" for this function): restated from its published algorithm --
    centre padding of n_fft//2 samples on both sides (zeros in librosa >= 0.10, reflection before), frames of n_fft samples
    every hop, periodic window, rfft; then magnitude (librosa.magphase), log1p, and (x - mean) / std with torch's UNBIASED std
    over the whole utterance.  Returns float64 [n_fft/2 + 1][1 + len(y)//hop].  Cross-checked against scipy.signal.stft in
    tests/test_oracle_vs_golden.py."""
    y = np.asarray(y, dtype=np.float64)
    n_fft = int(sample_rate * window_size)
    hop = int(sample_rate * window_stride)
    yp = np.pad(y, n_fft // 2, mode="reflect" if pad_mode == "reflect" else "constant")
    T = 1 + len(y) // hop
    w = spect_window(window, n_fft)
    frames = np.stack([yp[t * hop:t * hop + n_fft] * w for t in range(T)], 1)          # [n_fft][T]
    spect = np.log1p(np.abs(np.fft.rfft(frames, axis=0)))
    if normalize:
        spect = (spect - spect.mean()) / spect.std(ddof=1)
    return spect
