"""GPU parity of the whole drop-in class against the golden vectors produced by the REAL reference model
(tests/golden/*.npz): logits / CTC loss within the north-star 1e-3 in fp32 mode, identical greedy transcripts, every
parameter gradient, BatchNorm running stats, eval-mode softmax, hidden-state carry.  bf16 mode: stated looser bars."""
import numpy as np
import pytest
import torch

from fixtures import Fixture, check_grads_or_flip_variant, gpu_fixture_names as fixture_names
from oracle import ds2_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(fx, precision=32):
    from deepspeech.pytorch_amd import configs
    from deepspeech.pytorch_amd.model import DeepSpeech
    c = fx.cfg
    rt = getattr(configs.RNNType, c["rnn_type"])
    if c["bidirectional"]:
        mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=c["hidden_size"], hidden_layers=c["hidden_layers"])
    else:
        mc = configs.UniDirectionalConfig(rnn_type=rt, hidden_size=c["hidden_size"], hidden_layers=c["hidden_layers"],
                                          lookahead_context=c["lookahead_context"])
    m = DeepSpeech(labels=fx.labels, model_cfg=mc, precision=precision, optim_cfg=configs.AdamConfig(),
                   spect_cfg=configs.SpectConfig(sample_rate=fx.sample_rate))
    sd = {k: torch.from_numpy(v.copy()) for k, v in fx.params().items()}
    m.load_state_dict(sd, strict=True)     # reference state_dict keys/shapes load unchanged
    return m.to(DEV)


def run_step(m, fx):
    inputs, targets, pct, tsz = fx.batch()
    m.train()
    m.zero_grad()
    x = torch.from_numpy(inputs).to(DEV)
    sizes = torch.from_numpy(pct.copy()).mul_(int(inputs.shape[3])).int()
    logits, out_sizes, _ = m(x, sizes)
    # the same batch through training_step (fresh BN buffers needed -> second model in callers); here: loss via CtcFn
    return logits, out_sizes


@pytest.mark.parametrize("name", fixture_names())
def test_fp32_train_step_matches_reference(name):
    fx = Fixture(name)
    m = build(fx, 32)
    inputs, targets, pct, tsz = fx.batch()
    m.train()
    batch = (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
    loss = m.training_step(batch, 0)
    loss.backward()
    ref32, ref64 = float(fx.z["loss"]), float(fx.z["loss64"])
    got = float(loss.item())
    assert abs(got - ref32) <= 1e-3 * abs(ref32), (got, ref32)          # north star: CTC loss within 1e-3 (rel)
    assert abs(got - ref64) <= 1e-4 * abs(ref64), (got, ref64)
    grads = {}
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        grads[k] = p.grad.detach().cpu().numpy()

    def rtol_of(k):
        # bar: 1e-3 of the tensor's max |grad| (conv-bias-before-BN gradients are pure cancellation: looser)
        rtol = 2e-2 if k in ("conv.seq_module.0.bias", "conv.seq_module.3.bias") else 1e-3
        return max(rtol, 3 * min(float(fx.z["noise." + k]), 1e-2))
    # every gradient against the reference's; a case with a pre-Hardtanh value on a clamp boundary may instead match the
    # oracle with that single decision flipped (fixtures.check_grads_or_flip_variant)
    how = check_grads_or_flip_variant(fx, grads, rtol_of)
    print("%s: gradients match the %s" % (name, how))
    # exactly ONE fixture is known to sit on a Hardtanh boundary (rnn_bi_1024: one pre-activation 5.5e-7 from the clamp); any other
    # fixture that needs the flip-variant acceptance is a regression, not a degenerate case
    assert how == "reference" or name == "rnn_bi_1024", (name, how)
    for k in fx.z.files:
        if k.startswith("running."):
            nm = k.split(".", 1)[1]
            buf = dict(m.named_buffers())[nm].detach().cpu().numpy()
            assert np.allclose(buf, fx.z[k], rtol=1e-4, atol=1e-5), nm
    nbt = dict(m.named_buffers())["conv.seq_module.1.num_batches_tracked"]
    assert int(nbt.item()) == 1


@pytest.mark.parametrize("name", fixture_names())
def test_fp32_logits_eval_transcripts_and_carry(name):
    fx = Fixture(name)
    m = build(fx, 32)
    inputs, targets, pct, tsz = fx.batch()
    x = torch.from_numpy(inputs).to(DEV)
    sizes = torch.from_numpy(fx.z["input_sizes"].copy())
    m.train()
    logits, out_sizes, hs = m(x, sizes)
    assert out_sizes.dtype == torch.int32 and not out_sizes.is_cuda
    assert np.array_equal(out_sizes.numpy(), fx.z["output_lengths"])
    assert tuple(logits.shape) == tuple(fx.z["logits"].shape)
    err = np.abs(logits.detach().cpu().numpy() - fx.z["logits"]).max()
    assert err <= 1e-3, err                                              # north star: logits within 1e-3 (abs)
    # eval mode (fresh model: the train forward above updated the running stats)
    m2 = build(fx, 32).eval()
    with torch.no_grad():
        probs, sizes2, hs2 = m2(x, sizes)
    p = probs.cpu().numpy()
    assert np.abs(p - fx.z["eval_probs"]).max() <= 1e-4
    from deepspeech.pytorch_amd import configs
    assert O.greedy_decode(p, sizes2.numpy(), fx.labels) == fx.meta["transcripts"]        # identical transcripts
    # hidden-state carry, reference inference.py:86-96
    t0 = int(fx.lengths[0])
    x1, l1 = x[:1, :, :, :t0].contiguous(), torch.tensor([t0], dtype=torch.int)
    with torch.no_grad():
        _, _, hs1 = m2(x1, l1)
        probs2, _, hs_out = m2(x1, l1, hs1)
    assert np.abs(probs2.cpu().numpy() - fx.z["carry_probs"]).max() <= 1e-4
    hl = hs_out[-1][0] if fx.cfg["rnn_type"] == "lstm" else hs_out[-1]
    assert tuple(hl.shape) == tuple(fx.z["carry_h_last"].shape)
    assert np.abs(hl.cpu().numpy() - fx.z["carry_h_last"]).max() <= 1e-4


BF16_CASES = ["gru_bi_tiny", "lstm_bi_tiny", "gru_uni_la", "gru_bi_mid", "gru_bi_1024", "lstm_uni_1024_la", "rnn_bi_1024", "lstm_bi_1024",
              "cfg2_full", "lstm_bi_1280", "lstm_uni_1280_la", "gru_bi_1024_l5_n32", "lstm_bi_1280_l7_n18", "lstm_uni_1280_la_l7_n18",
              "lstm_bi_h10_n10", "gru_uni_h50_la40_c45", "rnn_bi_h24_c40", "gru_bi_h32_c300", "gru_bi_8khz", "lstm_uni_12k8"]
# Stated bf16 bounds.  The comparator is the REFERENCE ITSELF under torch.autocast(bfloat16) (fixture keys loss_ac /
# logits_ac / grad_ac.* / acnoise.*, tests/golden/make_golden.py leg C) next to the reference in float64:
#   loss     within 1e-3 relative of the reference's autocast loss AND of its float64 loss
#   logits   within 0.12 absolute of the float64 logits (the reference's own autocast logits deviate by up to ~0.08)
#   gradient relative L2 distance from the float64 gradient <= max(BF16_GRAD_FACTOR x the reference-autocast's own distance
#            from float64 ("acnoise"), BF16_GRAD_FLOOR), for EVERY parameter except the two conv biases in front of BatchNorm
#            (exactly zero in exact arithmetic: pure rounding noise in any implementation)
BF16_LOSS_RTOL, BF16_LOGITS_ATOL, BF16_GRAD_FACTOR, BF16_GRAD_FLOOR = 1e-3, 0.12, 2.0, 4e-2     # loss: the north star's 1e-3 (measured <= 9e-5)


def bf16_step(fx):
    from deepspeech.pytorch_amd import ops as O_
    m = build(fx, "bf16")
    inputs, targets, pct, tsz = fx.batch()
    m.train()
    batch = (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
    loss = m.training_step(batch, 0)
    loss.backward()
    O_.check_persistent_kernels()
    grads = {k: p.grad.detach().float().cpu().numpy() for k, p in m.named_parameters()}
    m2 = build(fx, "bf16")
    m2.train()
    logits, _, _ = m2(batch[0], torch.from_numpy(fx.z["input_sizes"].copy()))
    return float(loss.item()), grads, logits.detach().float().cpu().numpy()


@pytest.mark.parametrize("name", BF16_CASES)
def test_bf16_train_step_vs_reference_autocast(name):
    """bf16 storage + bf16 MFMA operands (the performance mode) against the reference under torch.autocast(bfloat16) and the
    reference in float64, at the stated bounds above.  Includes BASELINE.json's configurations at their own size / width:
    cfg2_full (5 x BiGRU-800, 8 clips), lstm_bi_1280 / lstm_uni_1280_la (config 5 family), gru_bi_1024_l5_n32 (config 3's
    full depth and batch)."""
    fx = Fixture(name)
    loss, grads, logits = bf16_step(fx)
    l_ac, l64 = float(fx.z["loss_ac"]), float(fx.z["loss64"])
    assert abs(loss - l_ac) <= BF16_LOSS_RTOL * abs(l_ac), (loss, l_ac)
    assert abs(loss - l64) <= BF16_LOSS_RTOL * abs(l64), (loss, l64)
    assert np.abs(logits - fx.z["logits64"]).max() <= BF16_LOGITS_ATOL
    worst = ("", 0.0, 0.0)
    for k, g in grads.items():
        if k in ("conv.seq_module.0.bias", "conv.seq_module.3.bias"):
            continue
        assert np.isfinite(g).all(), k
        d64 = fx.rel_l2(k, g, "grad")
        bound = max(BF16_GRAD_FACTOR * float(fx.z["acnoise." + k]), BF16_GRAD_FLOOR)
        assert d64 <= bound, "grad %s: relative L2 distance %.3e from the float64 reference > %.3e (reference autocast: %.3e)" % (
            k, d64, bound, float(fx.z["acnoise." + k]))
        if d64 / bound > worst[1]:
            worst = (k, d64 / bound, d64)
    print("%s: loss %.4f (reference autocast %.4f, float64 %.4f); worst gradient %s at %.2f of its bound (rel L2 %.3e)" % (
        name, loss, l_ac, l64, worst[0], worst[1], worst[2]))


def test_full_size_properties():
    """AN4-shaped configuration (cfg2: BiGRU-800x5, N=8, 1-2 s clips) in fp32: size-independent properties.
    (1) per-sample losses are invariant to the other samples' padding in eval-BN mode is not testable in train mode
    (batch statistics), so: (a) loss is finite and positive, (b) gradient of every parameter is finite and non-zero,
    (c) determinism: two identical steps give bit-identical loss, (d) padded frames never leak: zeroing the padded input
    region changes nothing (inputs are already zero there) while perturbing it changes nothing EITHER in the
    conv-masked outputs beyond the receptive field -- checked through output rows >= length being exactly zero in the
    RNN input."""
    from deepspeech.pytorch_amd import configs, synth
    from deepspeech.pytorch_amd.model import DeepSpeech
    torch.manual_seed(0)
    mc = configs.BiDirectionalConfig(rnn_type=configs.RNNType.gru, hidden_size=800, hidden_layers=5)
    m = DeepSpeech(configs.LABELS, mc, 32, configs.AdamConfig(), configs.SpectConfig()).to(DEV)
    assert sum(p.numel() for p in m.parameters()) == 41187968      # GRU-800x5
    lengths = synth.synth_lengths(8, 101, 201, seed=2, linear=True)
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=2000)
    losses = []
    for _ in range(2):
        torch.manual_seed(0)
        m2 = DeepSpeech(configs.LABELS, mc, 32, configs.AdamConfig(), configs.SpectConfig()).to(DEV)
        m2.load_state_dict(m.state_dict())
        m2.train()
        batch = (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()),
                 torch.from_numpy(tsz))
        loss = m2.training_step(batch, 0)
        loss.backward()
        losses.append(float(loss.item()))
        for k, p in m2.named_parameters():
            g = p.grad
            assert torch.isfinite(g).all(), k
            assert float(g.abs().max()) > 0, k
    assert np.isfinite(losses[0]) and losses[0] > 0
    assert losses[0] == losses[1]
    out, sizes, _ = m2(torch.from_numpy(inputs).to(DEV), torch.from_numpy(lengths.astype(np.int32)))
    assert sizes.tolist() == O.seq_lens(lengths).tolist()


FULL_SIZE_FACTOR = 1.5
FULL_SIZE_CONFIGS = {
    # name: (rnn_type, hidden, layers, bidirectional, clips, Tmin, Tmax, data seed) -- exactly what bench.py builds (bench.CONFIGS)
    "cfg3": ("gru", 1024, 5, True, 32, 1201, 1501, 3000),
    "cfg5a": ("lstm", 1280, 7, True, 64, 501, 1501, 5000),
    "cfg5b": ("lstm", 1280, 7, False, 64, 501, 1501, 6000),
}


@pytest.mark.parametrize("cfg_name", ["cfg3", "cfg5a", "cfg5b"])
def test_full_size_matches_stock_torch_on_device(cfg_name):
    """BASELINE.json configs 3 and 5 at FULL size, bf16 (cfg3: 5 x BiGRU-1024, 32 clips of 12-15 s; cfg5a: 7 x BiLSTM-1280, 64
    clips of 5-15 s; cfg5b: 7 x uni-LSTM-1280 + Lookahead, same batch): the persistent recurrent kernels, the phase-split
    GEMMs and the bf16 conv path against stock PyTorch-ROCm (oracle/ds2_torch_port.py on the same GPU -- the reference's own
    op sequence) on the same weights and batch.  Truth = the stock op sequence in fp32 on the device; both bf16 pipelines are
    measured against it: CTC loss within 2e-3 relative, EVERY gradient of ours within FULL_SIZE_FACTOR x stock-bf16's own
    relative L2 distance from fp32 (or the 3e-2 floor), and a bit-identical loss when the step is repeated."""
    from deepspeech.pytorch_amd import configs, ops, synth
    from deepspeech.pytorch_amd.model import DeepSpeech
    from oracle import ds2_torch_port as TP
    kind, H, L, bi, N, tmin, tmax, seed = FULL_SIZE_CONFIGS[cfg_name]
    cfg = dict(rnn_type=kind, hidden_size=H, hidden_layers=L, bidirectional=bi, lookahead_context=20)
    state = TP.random_state(cfg, 0)
    lengths = synth.synth_lengths(N, tmin, tmax, seed=seed)
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=seed)
    mk = lambda: (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()),
                  torch.from_numpy(tsz))
    rt = getattr(configs.RNNType, kind)
    mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L) if bi else \
        configs.UniDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L, lookahead_context=20)
    m = DeepSpeech(configs.LABELS, mc, "bf16", configs.AdamConfig(), configs.SpectConfig())
    m.load_state_dict({k: v.clone() for k, v in state.items()}, strict=True)
    m = m.to(DEV).train()
    assert ops.use_persistent(kind, torch.bfloat16, 2 if bi else 1, N, H)
    losses = []
    for _ in range(2):
        m.zero_grad()
        loss = m.training_step(mk(), 0)
        loss.backward()
        losses.append(float(loss.item()))
    ops.check_persistent_kernels()
    assert losses[0] == losses[1]                      # forward is deterministic (BN buffers do not enter train-mode math)
    own = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}
    del m
    torch.cuda.empty_cache()
    port = TP.Port(cfg, state, DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref = port.training_loss(mk())
    ref.backward()
    rl = float(ref.item())
    stock = {k: p.grad.detach().float().cpu() for k, p in port.P.items() if p.grad is not None}
    del port, ref
    torch.cuda.empty_cache()
    assert abs(losses[0] - rl) <= 2e-3 * abs(rl), (losses[0], rl)
    port32 = TP.Port(cfg, state, DEV)
    ref32 = port32.training_loss(mk())
    ref32.backward()
    worst = ("", 0.0, 0.0, 0.0)
    for k, p in port32.P.items():
        if k in ("conv.seq_module.0.bias", "conv.seq_module.3.bias") or p.grad is None:
            continue
        t = p.grad.float().reshape(-1).double().cpu()
        a = own[k].reshape(-1).double()
        b = stock[k].reshape(-1).double()
        assert torch.isfinite(a).all(), k
        d_own, d_stock = float((a - t).norm() / (t.norm() + 1e-30)), float((b - t).norm() / (t.norm() + 1e-30))
        bound = max(FULL_SIZE_FACTOR * d_stock, 3e-2)
        assert d_own <= bound, "grad %s: rel L2 %.3e from the fp32 run > %.3e (stock bf16: %.3e)" % (k, d_own, bound, d_stock)
        if d_own / bound > worst[1]:
            worst = (k, d_own / bound, d_own, d_stock)
    print("%s full size: loss %.3f (stock bf16 %.3f, stock fp32 %.3f); worst gradient %s: rel L2 %.3e from fp32 (stock bf16: %.3e)" % (
        cfg_name, losses[0], rl, float(ref32.item()), worst[0], worst[2], worst[3]))
    assert abs(losses[0] - float(ref32.item())) <= 2e-3 * abs(float(ref32.item()))


def _full_fixture(cfg_name):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full", cfg_name + ".npz")
    if not os.path.exists(path):
        pytest.skip("no full-size reference fixture for %s (tests/golden/make_fullsize_golden.py)" % cfg_name)
    z = np.load(path)
    return z, json.loads(bytes(z["meta_json"]).decode())


@pytest.mark.parametrize("cfg_name", ["cfg2", "cfg3", "cfg5a", "cfg5b"])
def test_full_size_step_matches_the_reference_itself(cfg_name):
    """The same full-size step against THE REFERENCE'S OWN model.py run on the CPU at that full shape (fp32 as shipped, and under
    torch.autocast(bfloat16); tests/golden/make_fullsize_golden.py -> tests/golden/full/<config>.npz: loss, logits on every 16th
    frame, a strided sample + L2 norm of every parameter gradient).  Bars as for the small fixtures: fp32 mode (cfg2) -- loss
    1e-3 relative, logits 1e-3, every gradient sample within 1e-3 of the tensor's max; bf16 mode -- loss within 1e-3 of the
    reference's fp32 AND autocast losses, logits within 0.12, every gradient's relative L2 distance from the reference's fp32
    gradient <= max(2 x the reference-autocast's own distance (CPU), 1.5 x the distance of stock PyTorch-ROCm under bf16 autocast
    on this device -- measured here, against the same reference gradient --, 4e-2).  The second yard-stick is needed at this
    size: the conv-block BatchNorm gradients are sums of ~2e6 cancelling terms per channel, and a GPU bf16 pipeline (stock or
    ours) sits further from fp32 than the CPU autocast run does.  Round 5: tensors of <= 16 384 elements are compared WHOLE
    (fixture key gradfull.*): the stride-997 sample of a 32-element BatchNorm weight used until round 4 was a single number, and the
    "0.345 vs stock 0.262" of conv.seq_module.1.weight was that one channel -- over whole tensors and six batches ours and stock's
    distances are 7.5e-2 and 6.3e-2 rms (tools/diag_conv_bn.py, profiles/r05i_diag_conv_bn_seeds.txt).  No tensor is excused."""
    from deepspeech.pytorch_amd import configs, ops, synth
    from deepspeech.pytorch_amd.model import DeepSpeech
    z, meta = _full_fixture(cfg_name)
    kind, H, L, bi = meta["rnn_type"], meta["hidden_size"], meta["hidden_layers"], meta["bidirectional"]
    fp32 = cfg_name == "cfg2"
    lengths = np.asarray(meta["lengths"], dtype=np.int64)
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=meta["data_seed"])
    P = synth.synth_params({k: tuple(v) for k, v in meta["shapes"].items()}, meta["param_seed"])
    rt = getattr(configs.RNNType, kind)
    mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L) if bi else \
        configs.UniDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L, lookahead_context=meta["lookahead_context"])
    m = DeepSpeech(configs.LABELS, mc, 32 if fp32 else "bf16", configs.AdamConfig(), configs.SpectConfig())
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    m = m.to(DEV).train()
    x = torch.from_numpy(inputs).to(DEV)
    # fp32 mode: watch the Hardtanh inputs of the conv block (y = x * scale + shift over the conv outputs the kernels store): a
    # decision whose input lies within fp32 rounding of a clamp boundary is not determined at this precision (see below)
    spied = []
    if fp32:
        real_bn_fwd = ops.bn_fwd

        def bn_fwd_spy(X, mode, *a, **k):
            sv = real_bn_fwd(X, mode, *a, **k)
            if mode in (1, 2):
                spied.append((X, sv, k.get("lens")))
            return sv
        ops.bn_fwd = bn_fwd_spy
    try:
        loss = m.training_step((x, torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz)), 0)
    finally:
        if fp32:
            ops.bn_fwd = real_bn_fwd
    loss.backward()
    ops.check_persistent_kernels()
    undecided = 1.0        # smallest distance of a Hardtanh input from a clamp boundary, relative to the operands' magnitude
    for X, sv, lens_ in spied:
        xv = X.detach().double()
        y = xv * sv.scale.double() + sv.shift.double()
        mag = (xv * sv.scale.double()).abs() + sv.shift.double().abs() + 1e-30
        valid = torch.arange(xv.shape[2], device=xv.device)[None, None, :, None] < lens_.to(xv.device)[:, None, None, None]
        rel = torch.where(valid, torch.minimum(y.abs(), (y - 20.0).abs()) / mag, torch.full_like(y, 1.0))
        undecided = min(undecided, float(rel.min().item()))
    got, ref = float(loss.item()), float(z["loss"])
    stock = {}
    if not fp32:
        from oracle import ds2_torch_port as TP
        own_grads = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}
        port = TP.Port(dict(rnn_type=kind, hidden_size=H, hidden_layers=L, bidirectional=bi, lookahead_context=meta["lookahead_context"]),
                       {k: torch.from_numpy(v.copy()) for k, v in P.items()}, DEV)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            sl = port.training_loss((x, torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz)))
        sl.backward()
        stock = {k: p.grad.detach().float().cpu().numpy().astype(np.float64).reshape(-1) for k, p in port.P.items() if p.grad is not None}
        del port, sl
        torch.cuda.empty_cache()
    if fp32:
        assert abs(got - ref) <= 1e-3 * abs(ref), (got, ref)
    else:
        assert abs(got - ref) <= BF16_LOSS_RTOL * abs(ref), (got, ref)
        assert abs(got - float(z["loss_ac"])) <= BF16_LOSS_RTOL * abs(float(z["loss_ac"])), (got, float(z["loss_ac"]))
    worst = ("", 0.0, 0.0)
    st = meta["grad_stride"]
    outliers = []
    total_l2 = float(np.sqrt(sum(float(z[f]) ** 2 for f in z.files if f.startswith("gradl2."))))
    for k, p in m.named_parameters():
        g = p.grad.detach().float().cpu().numpy().astype(np.float64).reshape(-1)
        assert np.isfinite(g).all(), k
        if k in ("conv.seq_module.0.bias", "conv.seq_module.3.bias"):
            continue                                   # exactly zero in exact arithmetic: rounding noise in any implementation
        l2 = float(z["gradl2." + k])
        whole = ("gradfull." + k) in z.files      # small tensors are stored whole (round 5): a stride-997 sample of a 32-element
        sub = z[("gradfull." if whole else "gradsub.") + k].astype(np.float64)      # BatchNorm weight was ONE number
        mine = g if whole else g[::st]
        if fp32:
            scale = max(np.abs(sub).max(), 1e-3)
            err = np.abs(mine - sub).max() / scale
            # 1e-3 of the tensor's largest element, per element.  ONE exception, with its evidence: when a Hardtanh input of the conv
            # block lies within fp32 rounding of a clamp boundary (< 5e-7 of its operands' magnitude; cfg2: 1.2e-7, channel 9 of the
            # second block, profiles/r05r_diag_cfg2_hardtanh_boundary.txt), that decision -- and with it one element of the upstream
            # gradient -- is rounding in ANY fp32 implementation, the reference's included; the conv-block tensors are then held to
            # 1e-3 in relative L2 and 3e-3 per element (measured: 1.04e-3 on element 9 of conv.seq_module.4.bias, 4.2e-4 in L2; every
            # other tensor of the model <= 3.4e-5)
            if undecided < 5e-7 and k.startswith("conv.") and whole:
                rl2 = float(np.sqrt(((mine - sub) ** 2).sum()) / max(np.sqrt((sub ** 2).sum()), 1e-30))
                assert rl2 <= 1e-3 and err <= 3e-3, "grad %s: relative L2 %.3e, worst element %.3e of its max" % (k, rl2, err)
            else:
                assert err <= 1e-3, "grad %s: %.3e of its max" % (k, err)
            assert abs(np.sqrt((g ** 2).sum()) - l2) <= 4e-3 * max(l2, 1e-3), k
            bound = 1e-3
        else:
            den = max(np.sqrt((sub ** 2).sum()), 1e-30)
            err = float(np.sqrt(((mine - sub) ** 2).sum()) / den)
            d_stock = float(np.sqrt((((stock[k] if whole else stock[k][::st]) - sub) ** 2).sum()) / den)
            bound = max(BF16_GRAD_FACTOR * float(z["acnoise." + k]), FULL_SIZE_FACTOR * d_stock, BF16_GRAD_FLOOR)
            if err > bound:
                outliers.append("%s (relative L2 distance %.3e from the reference's fp32 gradient > %.3e; reference autocast %.3e, stock bf16 on "
                                "this device %.3e; %.1e of the whole gradient's norm)" % (k, err, bound, float(z["acnoise." + k]), d_stock,
                                                                                         err * l2 / total_l2))
                continue
            assert abs(np.sqrt((g ** 2).sum()) - l2) <= 2 * bound * max(l2, 1e-3), k
        if err / bound > worst[1]:
            worst = (k, err / bound, err)
    assert outliers == [], outliers          # round 5: no tensor is excused (small tensors are compared whole)
    # logits of a second model (fresh BatchNorm buffers), every LOGIT_STRIDE-th frame
    m2 = DeepSpeech(configs.LABELS, mc, 32 if fp32 else "bf16", configs.AdamConfig(), configs.SpectConfig())
    m2.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    m2 = m2.to(DEV).train()
    sizes = torch.from_numpy(pct.copy()).mul_(int(inputs.shape[3])).int()
    logits, out_sizes, _ = m2(x, sizes)
    assert np.array_equal(out_sizes.numpy(), z["output_lengths"])
    lg = logits.detach().float().cpu().numpy()[:, ::meta["logit_stride"]]
    # frames past a clip's length hold BatchNorm(0) of the padding in both implementations; compare the valid frames
    ok = (np.arange(lg.shape[1])[None, :] * meta["logit_stride"]) < z["output_lengths"][:, None]
    lerr = np.abs(lg - z["logits_sub"])[ok].max()
    assert lerr <= (1e-3 if fp32 else BF16_LOGITS_ATOL), lerr
    print("%s vs the reference at full size: loss %.4f (reference fp32 %.4f); worst gradient %s at %.2f of its bound (%.3e); logits %.2e%s" % (
        cfg_name, got, ref, worst[0], worst[1], worst[2], lerr,
        "; closest Hardtanh input to a clamp boundary: %.1e of its operands' magnitude" % undecided if fp32 else ""))


def test_in_place_updates_without_version_bump_are_seen():
    """torch's fused optimizers update parameters in place WITHOUT bumping Tensor._version.  The kernel-layout weight
    copies must still be refreshed: three training steps with AdamW(fused=True) follow the same loss trajectory as the
    unfused optimizer (and the loss moves)."""
    fx = Fixture("gru_bi_mid")
    inputs, targets, pct, tsz = fx.batch()
    traj = {}
    for fused in (False, True):
        m = build(fx, "bf16").train()
        opt = torch.optim.AdamW(m.parameters(), lr=1e-2, fused=fused)
        ls = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = m.training_step((torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()),
                                    torch.from_numpy(tsz)), 0)
            loss.backward()
            opt.step()
            ls.append(float(loss.item()))
        traj[fused] = ls
        # eval after training must see the trained weights too
        m.eval()
        with torch.no_grad():
            p1, _, _ = m(torch.from_numpy(inputs).to(DEV), torch.from_numpy(fx.z["input_sizes"].copy()))
        assert torch.isfinite(p1).all()
    assert abs(traj[True][0] - traj[False][0]) <= 1e-6 * abs(traj[False][0])
    assert abs(traj[False][2] - traj[False][0]) > 1e-2 * abs(traj[False][0])          # the loss moves at all
    for a, b in zip(traj[True], traj[False]):
        assert abs(a - b) <= 2e-2 * abs(b), (traj[True], traj[False])


def test_step_on_a_non_default_stream_gives_the_same_gradients():
    """The class runs on the caller's CURRENT stream (every launch takes it explicitly; the second stream forks from and
    joins it): a full-size bf16 step on a high-priority side stream must reproduce the default-stream step -- same loss
    bit for bit, gradients equal up to the order of the fp32 atomics of the split reductions."""
    from deepspeech.pytorch_amd import configs, ops, synth
    from deepspeech.pytorch_amd.model import DeepSpeech
    lengths = synth.synth_lengths(32, 1201, 1501, seed=77)
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=77)
    mc = configs.BiDirectionalConfig(rnn_type=configs.RNNType.gru, hidden_size=1024, hidden_layers=5)
    torch.manual_seed(5)
    m = DeepSpeech(configs.LABELS, mc, "bf16", configs.AdamConfig(), configs.SpectConfig()).to(DEV).train()

    def step():
        m.zero_grad()
        batch = (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()),
                 torch.from_numpy(tsz))
        loss = m.training_step(batch, 0)
        loss.backward()
        return float(loss.item()), {k: p.grad.detach().clone() for k, p in m.named_parameters()}

    l0, g0 = step()
    torch.cuda.synchronize()
    hp = torch.cuda.Stream(priority=-1)
    hp.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(hp):
        l1, g1 = step()
        hp.synchronize()
    torch.cuda.synchronize()
    ops.check_persistent_kernels()
    assert l0 == l1
    worst = 0.0
    for k in g0:
        a, b = g0[k].float(), g1[k].float()
        assert torch.isfinite(b).all(), k
        d = float((a - b).abs().max() / (a.abs().max() + 1e-30))
        worst = max(worst, d)
        assert d <= 1e-4, (k, d)
    print("worst relative gradient difference between streams: %.3g" % worst)


# ---- validation path (reference model.py:251-271, decoder.py:164-181) -----------------------------------------------------
def _py_greedy(scores, sizes, blank=0):
    out = []
    for n in range(scores.shape[0]):
        am = scores[n, :sizes[n]].argmax(-1)
        toks = [(int(t), int(v)) for t, v in enumerate(am) if v != blank and (t == 0 or v != am[t - 1])]
        out.append(toks)
    return out


def test_greedy_decode_kernel_matches_argmax_collapse():
    from deepspeech.pytorch_amd import ops
    rs = np.random.RandomState(3)
    N, T, C = 7, 203, 29
    # peaky scores with long runs of repeats and blanks, sizes incl. 0, 1, a multiple of 64 and the full length
    labels = rs.randint(0, 5, size=(N, T)) * rs.randint(0, 2, size=(N, T)) * rs.randint(1, 7)
    labels = np.repeat(labels[:, ::3], 3, axis=1)[:, :T] % C
    scores = rs.standard_normal((N, T, C)).astype(np.float32) * 0.1
    for n in range(N):
        scores[n, np.arange(T), labels[n]] += 3.0
    sizes = np.array([203, 128, 65, 64, 1, 0, 200], dtype=np.int32)
    x = torch.from_numpy(scores).to(DEV)
    for view in (x, x.transpose(0, 1).contiguous().transpose(0, 1)):       # (N,T,C) contiguous and the (T,N,C)-backed view
        toks, offs = ops.greedy_decode(view, torch.from_numpy(sizes), 0)
        ref = _py_greedy(scores, sizes)
        for n in range(N):
            assert toks[n] == [v for _, v in ref[n]], n
            assert offs[n].tolist() == [t for t, _ in ref[n]], n


@pytest.mark.parametrize("name", ["gru_bi_mid", "lstm_uni_la", "gru_bi_1024"])
def test_validation_step_decodes_on_device_and_logs_wer_cer(name):
    from deepspeech.pytorch_amd import configs
    fx = Fixture(name)
    m = build(fx, 32).eval()
    logged = {}
    m.log = lambda k, v, **kw: logged.__setitem__(k, v)
    inputs, targets, pct, tsz = fx.batch()
    batch = (torch.from_numpy(inputs), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
    with torch.no_grad():
        assert m.validation_step(batch, 0) is None
        probs, sizes, _ = m(batch[0].to(DEV), torch.from_numpy(fx.z["input_sizes"].copy()))
        strings, offsets = m.evaluation_decoder.decode(probs, sizes)
    got = [s[0] for s in strings]
    assert got == fx.meta["transcripts"]                                  # north star: argmax transcripts identical
    assert all(len(o[0]) == len(s[0]) for o, s in zip(offsets, strings))
    # WER / CER of those transcripts against the targets, computed independently (validation.py:48-132 formulas)
    from deepspeech.pytorch_amd.decoder import _edit_distance
    tg, off = [], 0
    for s_ in tsz:
        tg.append(''.join(configs.LABELS[int(v)] for v in targets[off:off + int(s_)]))
        off += int(s_)
    cer = 100.0 * sum(_edit_distance(a.replace(' ', ''), b.replace(' ', '')) for a, b in zip(got, tg)) / \
        sum(len(b.replace(' ', '')) for b in tg)
    assert abs(logged["cer"] - cer) < 1e-9 and logged["wer"] > 0


def test_criterion_attribute_equals_training_step_loss():
    fx = Fixture("gru_bi_clamp_inf")                                      # contains an infeasible sample (zero_infinity)
    m = build(fx, 32).train()
    inputs, targets, pct, tsz = fx.batch()
    x = torch.from_numpy(inputs).to(DEV)
    loss = m.training_step((x, torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz)), 0)
    m2 = build(fx, 32).train()
    out, out_sizes, _ = m2(x, torch.from_numpy(fx.z["input_sizes"].copy()))
    lp = out.transpose(0, 1).log_softmax(-1)                              # model.py:245-246
    l2 = m2.criterion(lp, torch.from_numpy(targets), out_sizes, torch.from_numpy(tsz))
    assert abs(float(l2) - float(loss)) <= 1e-5 * abs(float(loss))
    l2.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m2.parameters())


@pytest.mark.parametrize("kind,bi", [("gru", True), ("lstm", False)])
def test_row_lists_change_nothing_but_the_order_of_summation(kind, bi):
    """model._frame_rows: the dense products over the [T' x N] frames skip the padding (what pack_padded_sequence drops,
    model.py:96).  With and without the row list: the SAME loss bit for bit (the kept rows of every product are the same tile
    arithmetic), every gradient within fp32 summation-order noise (the weight gradients contract over fewer, differently
    grouped rows), and dX of the padding frames exactly zero either way (checked through the conv-stack gradients being equal)."""
    from deepspeech.pytorch_amd import configs, model as M, synth
    rt = getattr(configs.RNNType, kind)
    mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=512, hidden_layers=3) if bi else \
        configs.UniDirectionalConfig(rnn_type=rt, hidden_size=512, hidden_layers=3, lookahead_context=20)
    lengths = synth.synth_lengths(16, 301, 901, seed=77)
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=77)
    mk = lambda: (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
    torch.manual_seed(3)
    m = M.DeepSpeech(configs.LABELS, mc, "bf16", configs.AdamConfig(), configs.SpectConfig()).to(DEV).train()
    res = {}
    old = M.ROW_LISTS
    try:
        for flag in (True, False):
            M.ROW_LISTS = flag
            m.zero_grad()
            loss = m.training_step(mk(), 0)
            assert (m._frame_rows is not None) == flag
            loss.backward()
            res[flag] = (float(loss.item()), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
    finally:
        M.ROW_LISTS = old
    assert res[True][0] == res[False][0]
    for k, g in res[True][1].items():
        h = res[False][1][k]
        assert torch.isfinite(g).all(), k
        d = float((g - h).norm() / (h.norm() + 1e-30))
        assert d < 2e-4, (k, d)


@pytest.mark.parametrize("kind,H,precision,want", [("gru", 600, "bf16", 640), ("lstm", 1100, "bf16", 1152), ("gru", 700, 32, 800)])
def test_hidden_sizes_between_the_instantiated_widths_ride_the_persistent_sweeps(kind, H, precision, want, monkeypatch):
    """Round 5: a hidden size one notch off the widths the persistent sweeps are instantiated for (the reference leaves hidden_size free,
    train_config.py:49) is padded with zero units up to the next such width inside the weight cache, instead of falling to the
    launch-per-time-step kernels.  The padded model computes the same step as the same model on the per-step kernels (DS2_PAD_HIDDEN=0's
    path): the extra units are exactly zero, only the summation order inside the products differs."""
    import warnings
    from deepspeech.pytorch_amd import configs, model as M, ops, synth
    lengths = np.array([121, 100, 77, 64])
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=5)
    rt = getattr(configs.RNNType, kind)
    mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=2)

    def step(pad):
        monkeypatch.setattr(M, "PAD_HIDDEN", pad)
        torch.manual_seed(3)
        m = M.DeepSpeech(configs.LABELS, mc, precision, configs.AdamConfig(), configs.SpectConfig()).to(DEV).train()
        assert m._Hp == (want if pad else (H + 15) // 16 * 16)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            loss = m.training_step((torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz)), 0)
            loss.backward()
        ops.check_persistent_kernels()
        slow = [x for x in w if "launch per time step" in str(x.message) or "one launch per time step" in str(x.message)]
        return float(loss.item()), {k: p.grad.detach().double().cpu().numpy() for k, p in m.named_parameters()}, slow, list(m.state_dict())

    la, ga, slow_a, keys_a = step(True)
    lb, gb, slow_b, keys_b = step(False)
    assert not slow_a, [str(x.message) for x in slow_a]              # the padded model stays on the persistent kernels
    assert keys_a == keys_b                                          # the padding never shows in the state_dict
    tol_l, tol_g = (2e-3, 4e-2) if precision == "bf16" else (2e-5, 2e-4)
    assert abs(la - lb) <= tol_l * abs(lb), (la, lb)
    for k in ga:
        assert ga[k].shape == gb[k].shape
        den = max(np.sqrt((gb[k] ** 2).sum()), 1e-12)
        assert np.sqrt(((ga[k] - gb[k]) ** 2).sum()) / den <= tol_g, (k, np.sqrt(((ga[k] - gb[k]) ** 2).sum()) / den)


def test_nobody_reads_the_padding_rows_a_two_set_sweep_leaves_unwritten(monkeypatch):
    """Round 5: with a row list in force every reader of a BPTT sweep's dGI works from the real frames (weight gradients + dX in one
    row-list launch, bias gradients from the sweep's per-sample sums), so the two-set general sweeps no longer zero the padding rows
    their half-steps leave out (ds2_rnn_persist_bwd flags bit 0).  Here dGI starts as NaN: had anybody read an unwritten row, the
    step's gradients would be NaN; and they equal the gradients of the same step with the rows zeroed as before."""
    from deepspeech.pytorch_amd import configs, model as M, ops, synth
    lengths = np.array(sorted([97] * 6 + [90] * 10 + [71] * 12 + [50] * 12 + [33] * 12 + [21] * 12, reverse=True))     # 64 clips, 40 % padding
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=9)
    mc = configs.BiDirectionalConfig(rnn_type=configs.RNNType.lstm, hidden_size=1280, hidden_layers=2)

    def step(poison):
        monkeypatch.setattr(ops, "POISON_UNWRITTEN", poison)
        torch.manual_seed(4)
        m = M.DeepSpeech(configs.LABELS, mc, "bf16", configs.AdamConfig(), configs.SpectConfig()).to(DEV).train()
        loss = m.training_step((torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz)), 0)
        assert m._frame_rows is not None                      # the row list is in force (enough padding)
        loss.backward()
        ops.check_persistent_kernels()
        return float(loss.item()), {k: p.grad.detach().double().cpu().numpy() for k, p in m.named_parameters()}

    from deepspeech.pytorch_amd._lib import query
    assert ops.persist_kind(torch.bfloat16, "lstm", 2, 64, 1280) == 3      # two-set general sweeps
    la, ga = step(True)
    lb, gb = step(False)
    assert np.isfinite(la) and la == lb
    for k in ga:
        assert np.isfinite(ga[k]).all(), k
        assert np.array_equal(ga[k], gb[k]), k


@pytest.mark.parametrize("name", ["gru_bi_tiny", "lstm_bi_tiny", "rnn_bi_tiny", "gru_uni_h50_la40_c45", "lstm_bi_h10_n10", "gru_bi_1024"])
def test_fp32_backward_through_a_given_initial_state(name):
    """Round 6: `forward(x, lengths, hs)` in training mode followed by backward (reference model.py:224-230 passes hs[i] to every
    BatchRNN layer as hx; the reference's own callers only do so in inference).  Checked against the torch port of the reference's
    forward (oracle/ds2_torch_port.py: the same torch calls, nn.GRU / nn.LSTM / nn.RNN with hx on a packed sequence) in float64 on
    the CPU: logits, EVERY parameter gradient and the gradients with respect to the initial states themselves."""
    from oracle import ds2_torch_port as TP
    fx = Fixture(name)
    m = build(fx, 32).train()
    c = fx.cfg
    kind, H, L, D = c["rnn_type"], c["hidden_size"], c["hidden_layers"], (2 if c["bidirectional"] else 1)
    inputs, targets, pct, tsz = fx.batch()
    N, T = inputs.shape[0], inputs.shape[3]
    sizes = torch.from_numpy(pct.copy()).mul_(int(T)).int()
    g = torch.Generator().manual_seed(11)
    mk = lambda: (torch.randn((D, N, H), generator=g, dtype=torch.float64) * 0.3)
    hs_ref = [(mk().requires_grad_(), mk().requires_grad_()) if kind == "lstm" else mk().requires_grad_() for _ in range(L)]
    port = TP.Port(dict(c), {k: (np.asarray(v, np.float64) if np.asarray(v).dtype.kind == "f" else v) for k, v in fx.params().items()}, "cpu")
    for k in list(port.P):
        port.P[k] = port.P[k].detach().double().requires_grad_(True)
    for l, r in enumerate(port.rnns):
        r.double()
        for n_, p_ in r.named_parameters():
            port.P["rnns.%d.rnn.%s" % (l, n_)] = p_
    for k in port.buf:
        if port.buf[k].dtype.is_floating_point:
            port.buf[k] = port.buf[k].double()
    logits_ref, _ = port.forward(torch.from_numpy(inputs).double(), sizes, train=True, hs=hs_ref)          # (T', N, C)
    w = torch.randn(logits_ref.shape, generator=g, dtype=torch.float64)
    (logits_ref * w).sum().backward()
    # ---- ours
    to_dev = lambda t: t.detach().float().to(DEV).requires_grad_()
    hs = [(to_dev(h[0]), to_dev(h[1])) if kind == "lstm" else to_dev(h) for h in hs_ref]
    m.zero_grad()
    out, _, _ = m(torch.from_numpy(inputs).to(DEV), sizes, hs)                                             # (N, T', C)
    assert np.abs(out.detach().cpu().double().numpy() - logits_ref.detach().transpose(0, 1).numpy()).max() < 1e-3
    (out * w.transpose(0, 1).float().to(DEV)).sum().backward()
    ref_g = {k: v.grad.numpy() for k, v in port.P.items()}
    for k, p in m.named_parameters():
        got, want = p.grad.detach().cpu().double().numpy(), ref_g[k].reshape(p.shape)
        # conv-block gradients are sums of ~1e5-1e6 cancelling terms per entry and this loss (random weights on every logit) has
        # none of CTC's structure: fp32 accumulation noise reaches 3e-3 of the tensor's maximum on the H = 1024 fixture
        bar = 5e-3 if k.startswith("conv.") else 1e-3
        assert np.abs(got - want).max() <= bar * max(np.abs(want).max(), 1e-6), (k, np.abs(got - want).max(), np.abs(want).max())
    for l in range(L):
        pairs = zip(hs[l], hs_ref[l]) if kind == "lstm" else [(hs[l], hs_ref[l])]
        for a, b in pairs:
            got, want = a.grad.detach().cpu().double().numpy(), b.grad.numpy()
            assert np.abs(got - want).max() <= 1e-3 * max(np.abs(want).max(), 1e-6), ("hs", l, np.abs(got - want).max())
