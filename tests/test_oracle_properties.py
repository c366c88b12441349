"""Size-independent properties of the CPU oracle itself (no GPU, no reference needed): they pin the restatement from a
second side -- the golden vectors say "equal to the reference on these inputs", these say "behaves like the algorithm
the reference's call sites name" (model.py:53-69 masking, :94-102 packed sequences, :203/:248 CTC sum with
zero_infinity, :299-310 lengths).  The GPU suite checks the same properties on the HIP path (tests/test_gpu_properties.py)."""
import itertools

import numpy as np
import pytest

from fixtures import Fixture
from oracle import ds2_oracle as O


def _brute_force_nll(lp, target, T, blank=0):
    """-log sum over ALL alignments pi in C^T whose collapse (merge repeats, drop blanks) equals `target`."""
    C = lp.shape[1]
    total = -np.inf
    for pi in itertools.product(range(C), repeat=T):
        col = [k for k, _ in itertools.groupby(pi)]
        col = [k for k in col if k != blank]
        if col == list(target):
            total = np.logaddexp(total, sum(lp[t, pi[t]] for t in range(T)))
    return -total


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ctc_oracle_equals_brute_force_over_all_alignments(seed):
    rs = np.random.RandomState(seed)
    T, N, C = 5, 4, 3
    lp = O.log_softmax(rs.standard_normal((T, N, C)) * 2.0)
    tl = np.array([2, 1, 3, 2])
    il = np.array([5, 4, 5, 3])
    targets = np.array([1, 2, 2, 1, 1, 2, 1, 1])            # sample 2: "1 1 2" needs a blank between the repeats; sample 3: "1 1" in 3 frames
    for fn in (O.ctc_loss_and_grad, O.ctc_loss_and_grad_fast):
        loss, nll, grad = fn(lp, targets, il, tl, blank=0)
        off = 0
        for i in range(N):
            want = _brute_force_nll(lp[:il[i], i], targets[off:off + tl[i]], int(il[i]))
            off += tl[i]
            assert abs(nll[i] - want) < 1e-9, (fn.__name__, i, nll[i], want)
        assert abs(loss - nll.sum()) < 1e-9
        assert np.all(grad[il[1]:, 1] == 0) and np.all(grad[il[3]:, 3] == 0)      # frames past a clip's length carry no gradient


def test_ctc_gradient_is_the_derivative_of_the_loss():
    rs = np.random.RandomState(5)
    T, N, C = 6, 2, 4
    x = rs.standard_normal((T, N, C))
    targets, tl, il = np.array([1, 3, 2, 2]), np.array([2, 2]), np.array([6, 5])

    def f(z):
        return O.ctc_loss_and_grad(O.log_softmax(z), targets, il, tl)[0]

    lp = O.log_softmax(x)
    _, _, dlp = O.ctc_loss_and_grad(lp, targets, il, tl)
    dlogits = dlp - np.exp(lp) * dlp.sum(-1, keepdims=True)          # log_softmax backward, as in O.training_step
    eps = 1e-6
    for idx in [(0, 0, 0), (2, 1, 3), (5, 0, 2), (4, 1, 1), (5, 1, 0)]:
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps
        xm[idx] -= eps
        num = (f(xp) - f(xm)) / (2 * eps)
        assert abs(num - dlogits[idx]) < 1e-6 * max(1.0, abs(num)), (idx, num, dlogits[idx])


def test_ctc_infeasible_clip_contributes_zero_loss_and_zero_gradient():
    """zero_infinity=True (model.py:203): a target that does not fit its frames (S > T', or repeats without room for the blank)."""
    rs = np.random.RandomState(3)
    T, N, C = 4, 3, 5
    lp = O.log_softmax(rs.standard_normal((T, N, C)))
    targets = np.array([1, 2, 3, 4, 1,   2, 2, 2,   3, 1])          # 5 labels in 4 frames; "2 2 2" needs 5 frames; feasible
    tl, il = np.array([5, 3, 2]), np.array([4, 4, 4])
    for fn in (O.ctc_loss_and_grad, O.ctc_loss_and_grad_fast):
        loss, nll, grad = fn(lp, targets, il, tl)
        assert nll[0] == 0 and nll[1] == 0 and nll[2] > 0
        assert np.all(grad[:, 0] == 0) and np.all(grad[:, 1] == 0) and np.any(grad[:, 2] != 0)
        alone, _, galone = fn(lp[:, 2:3], targets[8:], il[2:], tl[2:])
        assert abs(loss - alone) < 1e-12 and np.allclose(grad[:, 2:3], galone, atol=1e-12)


@pytest.mark.parametrize("name", ["gru_bi_tiny", "lstm_uni_la", "rnn_bi_tiny"])
def test_eval_outputs_do_not_depend_on_batch_composition_or_padding(name):
    """Eval mode (running statistics): a clip's valid output frames are a function of that clip alone -- MaskConv zeroes the
    padding after every conv-stack module (model.py:53-69), pack_padded_sequence hides it from the recurrence (:94-102)."""
    fx = Fixture(name)
    P = {k: np.asarray(v, np.float64) for k, v in fx.params().items()}
    inputs, _, pct, _ = fx.batch()
    x = inputs.astype(np.float64)
    lengths = O.input_sizes_from_percentages(pct, x.shape[3])
    out, olens, _, _ = O.model_forward(P, fx.cfg, x, lengths, train=False, keep_cache=False)
    i = len(lengths) - 1                                              # the shortest clip
    ti = int(lengths[i])
    alone, ol1, _, _ = O.model_forward(P, fx.cfg, x[i:i + 1, :, :, :ti], lengths[i:i + 1], train=False, keep_cache=False)
    assert int(ol1[0]) == int(olens[i])
    assert np.abs(alone[0, :ol1[0]] - out[i, :olens[i]]).max() < 1e-9
    # more padding, same answer for every clip
    pad = np.concatenate([x, np.zeros(x.shape[:3] + (24,))], axis=3)
    out2, olens2, _, _ = O.model_forward(P, fx.cfg, pad, lengths, train=False, keep_cache=False)
    assert np.array_equal(olens2, olens)
    for n in range(len(lengths)):
        assert np.abs(out2[n, :olens[n]] - out[n, :olens[n]]).max() < 1e-9


def test_training_loss_is_invariant_under_a_permutation_of_equal_length_clips():
    """Train-mode BatchNorm statistics are sums over the batch: swapping two clips of equal length permutes the logits and
    leaves the summed CTC loss and every parameter gradient unchanged (up to the order of floating-point sums)."""
    fx = Fixture("gru_bi_tiny")
    P = {k: np.asarray(v, np.float64) for k, v in fx.params().items()}
    inputs, targets, pct, tsz = fx.batch()
    x = inputs.astype(np.float64)
    x[1] = 0
    x[1, :, :, :] = np.random.RandomState(7).standard_normal(x[1].shape)
    pct = pct.copy()
    pct[1] = pct[0]                                                  # clips 0 and 1 now have equal lengths (keeps the order sorted)
    a = O.training_step(P, fx.cfg, x, targets, pct, tsz, fast_ctc=True)
    perm = [1, 0] + list(range(2, x.shape[0]))
    offs = np.concatenate([[0], np.cumsum(tsz)])
    tperm = np.concatenate([targets[offs[j]:offs[j + 1]] for j in perm])
    b = O.training_step(P, fx.cfg, x[perm], tperm, pct[perm], tsz[perm], fast_ctc=True)
    assert abs(a["loss"] - b["loss"]) < 1e-9 * abs(a["loss"])
    assert np.abs(a["logits"][perm] - b["logits"]).max() < 1e-9
    for k in a["grads"]:
        assert np.abs(a["grads"][k] - b["grads"][k]).max() <= 1e-9 * max(1.0, np.abs(a["grads"][k]).max()), k


def test_output_lengths_follow_the_reference_formula_for_every_length():
    """get_seq_lens (model.py:299-310): conv1 (k 11, stride 2, pad 5) then conv2 (stride 1): T' = floor((T - 1) / 2) + 1."""
    t = np.arange(1, 4000)
    assert np.array_equal(O.seq_lens(t), (t - 1) // 2 + 1)
