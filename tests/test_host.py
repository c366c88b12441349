"""CPU-side checks (run with -m "not gpu"): the C-ABI library builds, loads and exports every symbol that
include/ds2hip.h declares; the drop-in class has the reference's constructor surface, state_dict keys/shapes and
length arithmetic; the product path refuses CPU tensors (no fallback)."""
import os
import re

import numpy as np
import pytest
import torch

from fixtures import Fixture, fixture_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from deepspeech.pytorch_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from deepspeech.pytorch_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ds2hip.h")).read()
    declared = set(re.findall(r"\b(ds2_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("ds2_stream_t")
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libds2hip.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.ds2_version() >= 100
    assert b"alignment" in lib.ds2_error_string(1003)
    # pure size queries are host-only and safe without a GPU
    assert lib.ds2_rnn_gates(0) == 3 and lib.ds2_rnn_gates(1) == 4 and lib.ds2_rnn_gates(2) == 1
    assert lib.ds2_rnn_state_bytes(2, 32, 1024) == 2 * 32 * 1024 * 24
    assert lib.ds2_norm_partials(10) == 2 and lib.ds2_norm_partials(10 ** 7) == 1024
    # log-prob rows + alpha / beta rows of 2 S + 1 states and one pad float (an even stride: pair stores) + the beta shift + ll
    assert lib.ds2_ctc_ws_floats(751, 32, 29, 180) == 32 * 751 * 32 + 2 * 32 * 751 * 362 + 2 + 32
    # conv2 forward (bf16 storage): fp32 partial sums of the even kernel rows, [N][41][T'][32]; none for fp32 storage
    assert lib.ds2_conv2_fwd_ws_bytes(1, 32, 161, 751) == 32 * 41 * 751 * 32 * 4 and lib.ds2_conv2_fwd_ws_bytes(0, 32, 161, 751) == 0 and lib.ds2_conv2_fwd_ws_bytes(1, 32, 81, 751) == 0


def build_model(fx):
    from deepspeech.pytorch_amd import configs
    from deepspeech.pytorch_amd.model import DeepSpeech
    c = fx.cfg
    rt = getattr(configs.RNNType, c["rnn_type"])
    if c["bidirectional"]:
        mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=c["hidden_size"], hidden_layers=c["hidden_layers"])
    else:
        mc = configs.UniDirectionalConfig(rnn_type=rt, hidden_size=c["hidden_size"], hidden_layers=c["hidden_layers"],
                                          lookahead_context=c["lookahead_context"])
    return DeepSpeech(labels=fx.labels, model_cfg=mc, precision=32, optim_cfg=configs.AdamConfig(),
                      spect_cfg=configs.SpectConfig(sample_rate=fx.sample_rate))


@pytest.mark.parametrize("name", fixture_names())
def test_state_dict_matches_reference_keys_and_shapes(name):
    fx = Fixture(name)
    m = build_model(fx)
    sd = m.state_dict()
    ref_shapes = {k: tuple(v) for k, v in fx.meta["shapes"].items()}
    assert list(sd.keys()) == list(ref_shapes.keys()) or set(sd.keys()) == set(ref_shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == ref_shapes[k], k
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in fx.params().items()}, strict=True)
    assert m.bidirectional == fx.cfg["bidirectional"]
    assert (m.lookahead is None) == fx.cfg["bidirectional"]


def test_seq_lens_and_optimizers():
    fx = Fixture("gru_bi_tiny")
    m = build_model(fx)
    ln = torch.tensor([201, 186, 172, 158, 143, 129, 115, 101])
    assert m.get_seq_lens(ln).tolist() == [101, 93, 86, 79, 72, 65, 58, 51]
    (opt,), (sched,) = m.configure_optimizers()
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults["lr"] == 1.5e-4 and opt.defaults["weight_decay"] == 1e-5
    from deepspeech.pytorch_amd import configs
    m.optim_cfg = configs.SGDConfig()
    (opt,), _ = m.configure_optimizers()
    assert isinstance(opt, torch.optim.SGD) and opt.defaults["nesterov"] and opt.defaults["momentum"] == 0.9


def test_same_seed_same_init_as_reference_layout():
    """parameter creation order and init distributions follow torch's modules: same seed -> same tensors twice."""
    fx = Fixture("lstm_bi_tiny")
    torch.manual_seed(0)
    a = build_model(fx).state_dict()
    torch.manual_seed(0)
    b = build_model(fx).state_dict()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    h = fx.cfg["hidden_size"]
    w = a["rnns.0.rnn.weight_hh_l0"]
    assert float(w.abs().max()) <= 1.0 / np.sqrt(h) + 1e-7


def test_no_cpu_fallback(lib):
    from deepspeech.pytorch_amd._lib import Ds2HipError
    fx = Fixture("gru_bi_tiny")
    m = build_model(fx)
    inputs, _, pct, _ = fx.batch()
    with pytest.raises(Ds2HipError):
        m(torch.from_numpy(inputs), torch.from_numpy(fx.z["input_sizes"].copy()))


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "deepspeech", "pytorch_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("CPU oracle", "").replace("the oracle", ""), fn


# ---- validation_step boundary (reference model.py:202-212, 251-271) ------------------------------------------------------
def test_model_carries_the_reference_evaluation_attributes():
    fx = Fixture("gru_bi_tiny")
    m = build_model(fx)
    for attr in ("inference_softmax", "criterion", "evaluation_decoder", "wer", "cer"):
        assert hasattr(m, attr), attr
    assert m.criterion.blank == 0 and m.criterion.reduction == "sum" and m.criterion.zero_infinity is True
    # none of them adds state: reference checkpoints still load strictly (test_state_dict_matches_reference_keys_and_shapes)
    assert not any(k.startswith(("criterion", "inference_softmax", "wer", "cer")) for k in m.state_dict())


def test_greedy_decoder_string_building_equals_the_reference():
    """convert_to_strings / process_string (host side; the metrics use them for the TARGET strings) against the reference's
    GreedyDecoder on random label sequences, with and without repeat removal."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import ref_harness
    if not ref_harness.reference_available():
        pytest.skip("reference tree not present")
    ns = ref_harness.load_reference()
    from deepspeech.pytorch_amd import configs
    from deepspeech.pytorch_amd.decoder import GreedyDecoder
    ref, own = ns.GreedyDecoder(ns.labels), GreedyDecoder(configs.LABELS)
    assert configs.LABELS == ns.labels
    rs = np.random.RandomState(0)
    seqs = torch.from_numpy(rs.randint(0, 29, size=(6, 40)))
    sizes = torch.tensor([40, 33, 21, 7, 1, 0])
    for rr in (False, True):
        a, ao = ref.convert_to_strings(seqs, sizes, remove_repetitions=rr, return_offsets=True)
        b, bo = own.convert_to_strings(seqs, sizes, remove_repetitions=rr, return_offsets=True)
        assert a == b
        for x, y in zip(ao, bo):
            assert torch.equal(x[0], y[0])


def test_validation_step_drives_reference_decoder_and_metrics():
    """validation_step with the REFERENCE's GreedyDecoder / WordErrorRate / CharErrorRate objects (through the stubs of
    tests/golden/ref_harness.py) on the reference's own eval probabilities: same transcripts, WER / CER logged."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import ref_harness
    if not ref_harness.reference_available():
        pytest.skip("reference tree not present")
    ns = ref_harness.load_reference()
    from deepspeech_pytorch.validation import CharErrorRate, WordErrorRate
    fx = Fixture("gru_bi_mid")
    m = build_model(fx).eval()
    from deepspeech.pytorch_amd import model as own_model
    assert type(m.wer).__module__ == "deepspeech_pytorch.validation"      # the reference's classes are picked up when importable
    dec = ns.GreedyDecoder(ns.labels)
    m.attach_evaluation(dec, WordErrorRate(decoder=dec, target_decoder=dec), CharErrorRate(decoder=dec, target_decoder=dec))
    probs = torch.from_numpy(fx.z["eval_probs"])
    sizes = torch.from_numpy(fx.z["output_lengths"].copy())
    m.forward = lambda x, lengths, hs=None: (probs, sizes, None)          # the HIP forward is covered by the -m gpu tests
    logged = {}
    m.log = lambda k, v, **kw: logged.__setitem__(k, v)
    inputs, targets, pct, tsz = fx.batch()
    out = m.validation_step((torch.from_numpy(inputs), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz)), 0)
    assert out is None                                                     # model.py:251-271 returns nothing
    assert set(logged) == {"wer", "cer"} and all(np.isfinite(v) and v > 0 for v in logged.values())
    strings, _ = dec.decode(probs, sizes)
    assert [s[0] for s in strings] == fx.meta["transcripts"]


# ---- spectrogram front-end host side (window / DFT basis builders) ---------------------------------------------------------
@pytest.mark.parametrize("name", ["hamming", "hann", "blackman", "bartlett"])
def test_spectrogram_windows_equal_scipy(name):
    """SpectConfig.window names (reference enums.py:8-14) -> the periodic windows scipy.signal.get_window / librosa produce."""
    import scipy.signal as ss
    from deepspeech.pytorch_amd.spectrogram import window_values
    assert np.abs(window_values(name) - ss.get_window(name, 320, fftbins=True)).max() < 1e-12


def test_dft_basis_reproduces_rfft_of_windowed_frames():
    from deepspeech.pytorch_amd.spectrogram import dft_basis, window_values
    rs = np.random.RandomState(0)
    frames = rs.standard_normal((5, 320))
    B = dft_basis("hamming").astype(np.float64)               # [322][320]: cos rows then -sin rows
    out = frames @ B.T
    ref = np.fft.rfft(frames * window_values("hamming"), axis=1)
    assert np.abs(out[:, :161] - ref.real).max() < 1e-4 and np.abs(out[:, 161:] - ref.imag).max() < 1e-4


def test_bench_frame_arithmetic_equals_get_seq_lens():
    """bench.out_frames (valid output frames, incl. the float32 percentage round trip of model.py:243) == the class's
    get_seq_lens applied to training_step's input_sizes, for random length sets."""
    import bench
    fx = Fixture("gru_bi_tiny")
    m = build_model(fx)
    rs = np.random.RandomState(1)
    for _ in range(20):
        lengths = np.sort(rs.randint(41, 1502, size=rs.randint(1, 40)))[::-1].copy()
        tmax = int(lengths.max())
        pct = torch.from_numpy((lengths / float(tmax)).astype(np.float32))
        sizes = pct.mul_(tmax).int()
        assert m.get_seq_lens(sizes).tolist() == bench.out_frames(lengths).tolist()


def test_build_rejects_any_other_use_of_the_l2_touch_sink_register(tmp_path):
    """ds2_rnn_persist_impl.h lands its fire-and-forget scalar loads in one fixed SGPR whose write arrives asynchronously;
    build.py checks the device assembly of those sources: only the touches themselves may name the register."""
    from deepspeech.pytorch_amd import build
    obj = tmp_path / "k.o"
    asm = tmp_path / "k-hip-amdgcn-amd-amdhsa-gfx950.s"
    good = "\ts_load_dword s%d, s[42:43], 0x0\n\ts_add_u32 s8, s64, 0x800\n\tv_mov_b32_e32 v101, v3\n; NumSgprs: 108 s101\n" % build.L2_SINK
    asm.write_text(good)
    assert build._l2_sink_misuse(str(obj)) == []
    for bad in ("\ts_mov_b32 s%d, s3\n" % build.L2_SINK, "\ts_load_dwordx4 s[%d:%d], s[0:1], 0x0\n" % (build.L2_SINK - 1, build.L2_SINK + 2),
                "\ts_load_dword s%d, s[42:43], 0x10\n" % build.L2_SINK, "\ts_add_u32 s3, s%d, s4\n" % build.L2_SINK):
        asm.write_text(good + bad)
        assert len(build._l2_sink_misuse(str(obj))) == 1, bad
    asm.unlink()
    assert build._l2_sink_misuse(str(obj))          # no assembly to check = not verified = rejected


def test_frame_rows_lists_the_frames_pack_padded_sequence_keeps():
    """model._frame_rows: rows t*N + n with t < length[n], in storage order -- the same frames, in the same (time-major) order, that
    torch's pack_padded_sequence keeps (reference model.py:96); no list when the batch carries (almost) no padding or the switch
    is off."""
    import numpy as np
    import torch
    from deepspeech.pytorch_amd import model as M
    lens = torch.tensor([7, 5, 5, 2], dtype=torch.int32)
    Tp, N = 7, 4
    lens_dev, rows = M._frame_rows(lens, Tp, N, "cpu")
    assert lens_dev.tolist() == [7, 5, 5, 2] and rows.dtype == torch.int32
    x = torch.arange(Tp * N, dtype=torch.float32).reshape(Tp, N, 1)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens.long(), enforce_sorted=True)
    assert rows.tolist() == packed.data.reshape(-1).int().tolist()           # the packed data IS the listed rows, in order
    assert rows.numel() == int(lens.sum())
    # unsorted lengths: still the valid frames in storage order
    lens2 = torch.tensor([2, 7, 5, 5], dtype=torch.int32)
    _, rows2 = M._frame_rows(lens2, Tp, N, "cpu")
    want = [t * N + n for t in range(Tp) for n in range(N) if t < int(lens2[n])]
    assert rows2.tolist() == want
    # (almost) no padding -> no list; switch off -> no list
    _, none = M._frame_rows(torch.tensor([7, 7, 7, 7], dtype=torch.int32), Tp, N, "cpu")
    assert none is None
    old = M.ROW_LISTS
    try:
        M.ROW_LISTS = False
        assert M._frame_rows(lens, Tp, N, "cpu")[1] is None
    finally:
        M.ROW_LISTS = old


def test_persistent_sweep_start_up_budget_policy(monkeypatch):
    """ops.persist_startup_ms: 300 ms for a single process (fail fast, name the cause), the process group's time-out under data
    parallelism (a sweep behind an RCCL collective that waits for a late peer must WAIT, as a stock kernel would queue;
    loader/data_loader.py:320-360 hands ranks unequal batches), explicit overrides; persist_options restores what it changed."""
    from deepspeech.pytorch_amd import ops
    monkeypatch.delenv("DS2_PERSIST_STARTUP_MS", raising=False)
    assert ops.data_parallel_ranks() == 1
    assert ops.persist_startup_ms() == ops.STARTUP_MS_SINGLE == 300
    with ops.persist_options(startup_ms=1234, variant=32, spin_limit=7):
        assert ops.persist_startup_ms() == 1234
        o = ops._persist_opts()
        assert (o.variant, o.spin_limit, o.startup_ms) == (32, 7, 1234)
        with pytest.raises(ZeroDivisionError):
            with ops.persist_options(variant=1):
                1 / 0
        assert ops._OPTS["variant"] == 32                    # restored although the body raised
    o = ops._persist_opts()
    assert (o.variant, o.spin_limit, o.startup_ms) == (0, 0, 300)
    monkeypatch.setattr(ops, "data_parallel_ranks", lambda: 8)
    assert ops.STARTUP_MS_DP_MIN <= ops.persist_startup_ms() <= ops.STARTUP_MS_DP_MAX     # no group here: torch's 10-minute default
    assert ops.persist_startup_ms() == 600_000
    monkeypatch.setenv("DS2_PERSIST_STARTUP_MS", "45000")
    assert ops.persist_startup_ms() == 45000
