"""Pins the numpy oracle (oracle/ds2_oracle.py) against outputs of the real reference model (golden fixtures)."""
import numpy as np
import pytest

from fixtures import Fixture, fixture_names
from oracle import ds2_oracle as O


@pytest.mark.parametrize("name", fixture_names())
def test_train_step_matches_reference(name):
    """float64 oracle vs (a) the reference code run in float64 -- tight, (b) the reference as shipped (fp32)."""
    fx = Fixture(name)
    P = {k: v.astype(np.float64) if v.dtype == np.float32 else v for k, v in fx.params().items()}
    inputs, targets, pct, tsz = fx.batch()
    r = O.training_step(P, fx.cfg, inputs.astype(np.float64), targets, pct, tsz, fast_ctc=True)
    assert np.array_equal(r["output_lengths"], fx.z["output_lengths"])
    ref64 = float(fx.z["loss64"])
    assert abs(r["loss"] - ref64) <= 1e-10 * abs(ref64), (r["loss"], ref64)
    assert np.abs(r["logits"] - fx.z["logits64"]).max() <= 2e-6 * max(1.0, np.abs(fx.z["logits64"]).max())
    for gname in fx.grad_names():
        fx.check_grad(gname, r["grads"][gname], rtol=2e-6)     # fixture stores the float64 grads as float32
    # (b) fp32 reference: its own rounding noise is the only difference
    ref32 = float(fx.z["loss"])
    assert abs(r["loss"] - ref32) <= 2e-6 * abs(ref32), (r["loss"], ref32)
    assert np.abs(r["logits"] - fx.z["logits"]).max() <= 5e-5 * max(1.0, np.abs(fx.z["logits"]).max())
    for k in fx.z.files:
        if k.startswith("running."):
            nm = k.split(".", 1)[1]
            assert np.allclose(r["running"][nm], fx.z[k], rtol=1e-4, atol=1e-6), nm


@pytest.mark.parametrize("name", ["gru_bi_tiny", "gru_bi_clamp_inf"])
def test_float32_oracle_close_to_float64(name):
    """The oracle run in float32 (what the cpu_baseline times) stays within the north-star tolerance."""
    fx = Fixture(name)
    inputs, targets, pct, tsz = fx.batch()
    r = O.training_step(fx.params(), fx.cfg, inputs, targets, pct, tsz)
    assert abs(r["loss"] - float(fx.z["loss64"])) <= 1e-4 * abs(float(fx.z["loss64"]))
    assert np.abs(r["logits"] - fx.z["logits64"]).max() <= 1e-3


@pytest.mark.parametrize("name", ["gru_bi_tiny", "lstm_uni_la"])
def test_slow_ctc_equals_fast_ctc(name):
    fx = Fixture(name)
    inputs, targets, pct, tsz = fx.batch()
    T, N, C = int(fx.z["output_lengths"].max()), len(fx.lengths), 29
    rs = np.random.RandomState(5)
    lp = O.log_softmax(rs.standard_normal((T, N, C)))
    a = O.ctc_loss_and_grad(lp, targets, fx.z["output_lengths"], tsz)
    b = O.ctc_loss_and_grad_fast(lp, targets, fx.z["output_lengths"], tsz)
    assert abs(a[0] - b[0]) < 1e-9 * abs(a[0])
    assert np.abs(a[2] - b[2]).max() < 1e-9


@pytest.mark.parametrize("name", fixture_names())
def test_eval_forward_and_transcripts(name):
    fx = Fixture(name)
    P = {k: v.astype(np.float64) if v.dtype == np.float32 else v for k, v in fx.params().items()}
    inputs, _, pct, _ = fx.batch()
    sizes = O.input_sizes_from_percentages(pct, inputs.shape[3])
    assert np.array_equal(sizes, fx.z["input_sizes"])
    out, out_lens, hs, _ = O.model_forward(P, fx.cfg, inputs.astype(np.float64), sizes, train=False, keep_cache=False)
    assert np.abs(out - fx.z["eval_probs"]).max() < 2e-5
    assert O.greedy_decode(out, out_lens, fx.labels) == fx.meta["transcripts"]
    # hidden-state carry (reference inference.py:86-96)
    t0 = int(fx.lengths[0])
    x1 = inputs[:1, :, :, :t0].astype(np.float64)
    _, _, hs1, _ = O.model_forward(P, fx.cfg, x1, np.array([t0]), train=False, keep_cache=False)
    out2, _, hs2, _ = O.model_forward(P, fx.cfg, x1, np.array([t0]), train=False, hs=hs1, keep_cache=False)
    assert np.abs(out2 - fx.z["carry_probs"]).max() < 2e-5
    hl = hs2[-1][0] if fx.cfg["rnn_type"] == "lstm" else hs2[-1]
    assert np.abs(hl - fx.z["carry_h_last"]).max() < 2e-5


def test_seq_lens_formula():
    # reference model.py:299-310; survey probe: linspace(201,101,8) -> [101,93,86,79,72,65,58,51]
    ln = np.array([201, 186, 172, 158, 143, 129, 115, 101])
    assert O.seq_lens(ln).tolist() == [101, 93, 86, 79, 72, 65, 58, 51]
    assert O.rnn_input_size() == 1312


@pytest.mark.parametrize("name", fixture_names())
def test_torch_port_matches_reference(name):
    """oracle/ds2_torch_port.py (the stock-torch restatement timed as cpu_baseline / stock-ROCm baseline) reproduces the
    real reference's fp32 loss, logits and every gradient."""
    import torch
    from oracle import ds2_torch_port as TP
    fx = Fixture(name)
    port = TP.Port(fx.cfg, fx.params(), "cpu")
    inputs, targets, pct, tsz = fx.batch()
    batch = (torch.from_numpy(inputs), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
    loss = port.training_loss(batch)
    loss.backward()
    ref32 = float(fx.z["loss"])
    lv = float(loss.detach())
    assert abs(lv - ref32) <= 2e-5 * abs(ref32), (lv, ref32)
    logits, ol = port.forward(batch[0], torch.from_numpy(fx.z["input_sizes"].copy()), train=True)
    assert np.array_equal(ol.numpy(), fx.z["output_lengths"])
    assert np.abs(logits.detach().transpose(0, 1).numpy() - fx.z["logits"]).max() <= 1e-4
    for gname in fx.grad_names():
        noise = float(fx.z["noise." + gname]) if "noise." + gname in fx.z.files else 0.0
        fx.check_grad(gname, port.P[gname].grad.numpy(), rtol=max(2e-3, 3 * min(noise, 1e-2)))


def test_hardtanh_boundary_flip_explains_the_gpu_deviation_of_rnn_bi_1024():
    """rnn_bi_1024 has ONE pre-Hardtanh value (second conv block) 5.5e-7 from the clamp boundary 0.  An fp32 implementation
    may land on either side; flipping that single decision in the oracle moves the conv1 weight gradient by 2.513e-3 of its
    scale (2.800e-2 absolute) -- exactly the deviation measured on the MI355X fp32 path -- while everything matches the
    reference when the decision is left alone.  The flip-aware comparison accepts the flipped result and still rejects a
    result that matches neither."""
    from fixtures import check_grads_or_flip_variant, hardtanh_flip_variants
    fx = Fixture("rnn_bi_1024")
    variants = hardtanh_flip_variants(fx)
    assert 1 <= len(variants) <= 7
    k = "conv.seq_module.0.weight"
    ref = fx.z["grad." + k].astype(np.float64)
    scale = np.abs(ref).max()
    devs = [np.abs(v[k].reshape(ref.shape) - ref).max() / scale for v in variants]
    assert any(abs(d - 2.513e-3) < 2e-5 for d in devs), devs
    flipped = {n: np.asarray(g, dtype=np.float32) for n, g in variants[int(np.argmin([abs(d - 2.513e-3) for d in devs]))].items()}
    with pytest.raises(AssertionError):
        for n, g in flipped.items():
            fx.check_grad(n, g, rtol=1e-3)
    assert check_grads_or_flip_variant(fx, flipped, lambda n: 1e-3) == "flip-variant"
    wrong = dict(flipped)
    wrong[k] = flipped[k] * 1.01
    with pytest.raises(AssertionError):
        check_grads_or_flip_variant(fx, wrong, lambda n: 1e-3)


def test_log_spectrogram_oracle_against_scipy_stft():
    """The front-end oracle (librosa.stft restated; librosa is absent and unpinned) against an independent published STFT:
    scipy.signal.stft with the same framing (zero boundary extension of n_fft/2, hop 160, periodic hamming) is the same
    transform up to scipy's 1/sum(window) scaling."""
    import scipy.signal as ss
    rs = np.random.RandomState(0)
    y = rs.standard_normal(16000 + 37)
    raw = O.log_spectrogram(y, normalize=False)
    assert raw.shape == (161, 1 + len(y) // 160)
    f, t, Z = ss.stft(y, fs=16000, window="hamming", nperseg=320, noverlap=160, nfft=320, boundary="zeros", padded=False)
    mag = np.abs(Z) * ss.get_window("hamming", 320).sum()
    n = min(mag.shape[1], raw.shape[1])
    assert n >= raw.shape[1] - 1
    assert np.abs(np.log1p(mag[:, :n]) - raw[:, :n]).max() < 1e-9
    z = O.log_spectrogram(y)
    assert abs(z.mean()) < 1e-12 and abs(z.std(ddof=1) - 1) < 1e-12


@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_log_spectrogram_oracle_against_torch_stft(pad_mode):
    """A second independent implementation that IS in the image: torch.stft(center=True, periodic hamming window, hop 160,
    n_fft = win_length = 320) is the transform librosa.stft computes (torch documents it as librosa-compatible; the reference's
    own requirements pull both) -- for both centre paddings (zeros: librosa >= 0.10; reflection: before)."""
    import torch
    rs = np.random.RandomState(3)
    for n in (16000 + 37, 801, 480, 24000):
        y = rs.standard_normal(n) * 0.3 + 0.01
        Z = torch.stft(torch.from_numpy(y), n_fft=320, hop_length=160, win_length=320, window=torch.hamming_window(320, periodic=True, dtype=torch.float64),
                       center=True, pad_mode=pad_mode, return_complex=True)
        ref = torch.log1p(Z.abs()).numpy()
        raw = O.log_spectrogram(y, normalize=False, pad_mode=pad_mode)
        assert raw.shape == ref.shape == (161, 1 + n // 160)
        assert np.abs(raw - ref).max() < 1e-9
        # the reference's normalisation (data_loader.py:88-92: torch mean / UNBIASED std over the whole utterance)
        t = torch.from_numpy(ref)
        refn = ((t - t.mean()) / t.std()).numpy()
        assert np.abs(O.log_spectrogram(y, pad_mode=pad_mode) - refn).max() < 1e-9
