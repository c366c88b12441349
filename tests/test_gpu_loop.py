"""The drop-in class inside the reference's TRAINING LOOP contract (reference training.py:13-47 -> Lightning's automatic
optimization; tests/lightning_stub.py restates the loop statement by statement): fit a few steps with gradient_clip_val 400 + AdamW +
ExponentialLR, checkpoint in Lightning's .ckpt layout (state_dict + hyper_parameters + optimizer / scheduler state), restore through
``DeepSpeech.load_from_checkpoint`` (utils.py:31) and resume; and the reference's SHIPPED precision, ``precision: 16``
(configs/an4.yaml:11, librispeech.yaml:12) = autocast + ``GradScaler``: the backward of the HIP path under an upstream gradient of
65 536 instead of 1, and overflow detection by ``GradScaler.unscale_`` on the gradients the HIP path produced."""
import os

import numpy as np
import pytest
import torch

from fixtures import Fixture
from lightning_stub import MiniTrainer

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(fx, precision=32, optim="adam"):
    from deepspeech.pytorch_amd import configs
    from deepspeech.pytorch_amd.model import DeepSpeech
    c = fx.cfg
    rt = getattr(configs.RNNType, c["rnn_type"])
    if c["bidirectional"]:
        mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=c["hidden_size"], hidden_layers=c["hidden_layers"])
    else:
        mc = configs.UniDirectionalConfig(rnn_type=rt, hidden_size=c["hidden_size"], hidden_layers=c["hidden_layers"],
                                          lookahead_context=c["lookahead_context"])
    oc = configs.AdamConfig() if optim == "adam" else configs.SGDConfig()
    m = DeepSpeech(labels=fx.labels, model_cfg=mc, precision=precision, optim_cfg=oc, spect_cfg=configs.SpectConfig(sample_rate=fx.sample_rate))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in fx.params().items()}, strict=True)
    return m.to(DEV)


class _Data:
    """Stands in for DeepSpeechDataModule.train_dataloader(): CPU 4-tuples as the reference's collate function returns them."""

    def __init__(self, fx, n_batches):
        self.fx, self.n = fx, n_batches

    def train_dataloader(self):
        inputs, targets, pct, tsz = self.fx.batch()
        for i in range(self.n):
            rs = np.random.RandomState(100 + i)
            x = inputs + 0.05 * rs.standard_normal(inputs.shape).astype(np.float32) * (inputs != 0)
            yield (torch.from_numpy(x), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))


@pytest.mark.parametrize("name,precision", [("gru_bi_tiny", 32), ("lstm_uni_la", 32), ("gru_bi_1024", "bf16"), ("lstm_bi_1280", 32)])
def test_fit_checkpoint_resume(name, precision, tmp_path):
    """3 steps of fit (clip 400, AdamW, ExponentialLR) -> .ckpt -> load_from_checkpoint + resume -> the 4th step of the restored run
    equals the 4th step of the run that never stopped (loss and every parameter afterwards)."""
    fx = Fixture(name)
    m = build(fx, precision)
    tr = MiniTrainer(max_epochs=1, precision=32, gradient_clip_val=400.0)
    tr.fit(m, _Data(fx, 3))
    assert len(tr.losses) == 3 and all(np.isfinite(tr.losses)), tr.losses
    assert tr.losses[2] < tr.losses[0], tr.losses                      # AdamW at lr 1.5e-4 on near-identical batches: it learns
    path = os.path.join(tmp_path, "last.ckpt")
    tr.save_checkpoint(path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck["hyper_parameters"]) == {"labels", "model_cfg", "precision", "optim_cfg", "spect_cfg"}       # model.py:139-147
    assert list(ck["state_dict"]) == list(m.state_dict())

    from deepspeech.pytorch_amd.model import DeepSpeech
    m2 = DeepSpeech.load_from_checkpoint(path).to(DEV)                  # utils.py:31
    tr2 = MiniTrainer(max_epochs=2, precision=32, gradient_clip_val=400.0)
    tr2.resume(m2, path)
    assert tr2.current_epoch == 1 and tr2.global_step == 3
    assert abs(tr2.optimizer.param_groups[0]["lr"] - tr.optimizer.param_groups[0]["lr"]) < 1e-12        # ExponentialLR state restored
    batch = tuple(t.to(DEV) for t in next(iter(_Data(fx, 1).train_dataloader())))
    m.train(), m2.train()
    la = tr.train_batch(tuple(t.clone() for t in batch), 0)
    lb = tr2.train_batch(tuple(t.clone() for t in batch), 0)
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la)), (float(la), float(lb))
    for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        d = (p.detach() - q.detach()).abs().max().item()
        # Both runs start the 4th step from identical parameters, so they can differ only through what is not run-to-run
        # deterministic INSIDE a step: the split-K atomics of the small-shape weight-gradient GEMMs (model._wgrad_splitk: the
        # recurrent weight matrices and the head's Linear).  Their noise reaches the parameters through AdamW's normalised update
        # m / sqrt(v): where a gradient element is ~0 the ratio is noise, bounded by the learning rate (1.5e-4).  Every other
        # parameter (convolutions, BatchNorm, biases, lookahead) is held to the tight bound.
        splitk_fed = any(t in k for t in ("weight_ih", "weight_hh", "fc.0.module.1.weight"))
        floor = 0.3 * 1.5e-4 if splitk_fed else 1e-7
        assert d <= max(1e-4 * p.detach().abs().max().item(), floor), (k, d)
    from deepspeech.pytorch_amd import ops
    ops.check_persistent_kernels()


def _grads_of_step(m, batch, scale):
    m.train()
    m.zero_grad(set_to_none=True)
    loss = m.training_step(tuple(t.clone() for t in batch), 0)
    (loss * scale).backward()
    return float(loss.item()), {k: p.grad.detach().double().cpu().numpy() for k, p in m.named_parameters()}


@pytest.mark.parametrize("name,precision", [("gru_bi_mid", 32), ("lstm_uni_la", 32), ("gru_bi_1024", "bf16"), ("lstm_bi_1280", "bf16"),
                                            ("cfg2_full", 32)])
def test_gradscaler_scaled_backward(name, precision):
    """Upstream gradient 65 536 (GradScaler's initial scale) instead of 1 through _CtcFn / the whole HIP backward: after the division
    the gradients equal those of the unscaled step -- a power-of-two scale commutes with every fp32 / bf16 rounding short of
    overflow and underflow, so the bar is fp32 noise, not a bf16 bar."""
    fx = Fixture(name)
    inputs, targets, pct, tsz = fx.batch()
    batch = (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
    l1, g1 = _grads_of_step(build(fx, precision), batch, 1.0)
    l2, g2 = _grads_of_step(build(fx, precision), batch, 65536.0)
    assert l1 == l2
    for k in g1:
        a, b = g1[k], g2[k] / 65536.0
        err = np.abs(a - b).max() / max(np.abs(a).max(), 1e-30)
        assert err <= 2e-6, (k, err)


@pytest.mark.parametrize("name,precision", [("gru_bi_mid", 16), ("gru_bi_1024", 16)])
def test_precision16_loop_and_overflow_detection(name, precision):
    """The reference's shipped ``precision: 16`` loop (autocast + GradScaler): a normal step updates the parameters and leaves the
    scale alone; a step whose scaled gradients overflow fp32 (scale 2^127) produces inf / NaN gradients that ``unscale_`` SEES --
    the optimizer step is skipped, the parameters stay put and the scale is halved -- and nothing hangs or poisons the next step."""
    fx = Fixture(name)
    m = build(fx, precision)
    tr = MiniTrainer(max_epochs=1, precision=16, gradient_clip_val=400.0, init_scale=65536.0)
    tr.setup(m)
    m.train()
    inputs, targets, pct, tsz = fx.batch()
    batch = (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets).to(DEV), torch.from_numpy(pct.copy()).to(DEV),
             torch.from_numpy(tsz).to(DEV))                               # Lightning moves every tensor of the batch to the device
    p0 = {k: p.detach().clone() for k, p in m.named_parameters()}
    tr.train_batch(tuple(t.clone() for t in batch), 0)
    assert tr.skipped_steps == 0 and tr.scaler.get_scale() == 65536.0
    assert any((p.detach() != p0[k]).any().item() for k, p in m.named_parameters())
    # overflow: the scaled upstream gradient is 2^127
    tr.scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 127)
    p1 = {k: p.detach().clone() for k, p in m.named_parameters()}
    tr.train_batch(tuple(t.clone() for t in batch), 1)
    assert tr.skipped_steps == 1 and tr.scaler.get_scale() == 2.0 ** 126
    for k, p in m.named_parameters():
        assert torch.equal(p.detach(), p1[k]), k
    # and the loop goes on
    tr.scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    l3 = tr.train_batch(tuple(t.clone() for t in batch), 2)
    assert np.isfinite(float(l3)) and tr.skipped_steps == 1
    from deepspeech.pytorch_amd import ops
    ops.check_persistent_kernels()


@pytest.mark.parametrize("name,precision", [("gru_bi_1024", "bf16"), ("lstm_uni_la", 32), ("cfg2_full", 32), ("lstm_bi_1280", "bf16")])
def test_per_layer_nodes_equal_the_composite_node(name, precision):
    """The round-4 graph (conv stack -> one autograd node per BatchRNN layer) against the round-1..3 graph (ONE composite node,
    DS2_COMPOSITE_NODE=1): same kernels in the same order on the caller's stream -- identical loss and gradients."""
    from deepspeech.pytorch_amd import model as M
    fx = Fixture(name)
    inputs, targets, pct, tsz = fx.batch()
    batch = (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
    res = []
    old = M.COMPOSITE_NODE
    try:
        for comp in (False, True):
            M.COMPOSITE_NODE = comp             # False: one node per layer; True: the composite node (None = per-shape default)
            res.append(_grads_of_step(build(fx, precision), batch, 1.0))
    finally:
        M.COMPOSITE_NODE = old
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        a, b = res[0][1][k], res[1][1][k]
        assert np.abs(a - b).max() <= 1e-6 * max(np.abs(a).max(), 1e-30), k
