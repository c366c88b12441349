"""Stock-PyTorch restatement of the DeepSpeech2 train step.  TEST / MEASUREMENT INFRASTRUCTURE ONLY.

The reference (``/root/reference/deepspeech_pytorch/model.py``) is pure Python and hands every op of the hot path to
PyTorch.  This file restates that op sequence functionally, with the SAME torch calls the reference's modules end up
making (conv2d, batch_norm, hardtanh, masked_fill, pack_padded_sequence -> nn.GRU/LSTM/RNN -> pad_packed_sequence,
linear, log_softmax, ctc_loss, clip_grad_norm_, AdamW), so that

  * ``bench.py``'s ``cpu_baseline`` leg can time "what the reference does on the host cores" on the GPU box, where
    ``/root/reference`` does not exist (kind = "port"), and
  * ``bench.py --stock`` can time stock PyTorch-ROCm (MIOpen conv/BN/RNN + ATen CTC) on the MI355X: the denominator of
    the north star's ">= 3x over stock PyTorch-ROCm" target.

It is pinned against the golden vectors generated from the real reference (``tests/test_oracle_vs_golden.py``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s baseline legs may import it; the product path
(``deepspeech.pytorch_amd``) never does.

Reference lines restated: mask after every conv-stack module (model.py:53-69), conv stack (157-164), collapse/transpose
(219-221), BatchRNN = [BN over T*N rows] -> packed RNN -> direction sum (94-102), Lookahead (125-130) + Hardtanh
(189-193), head BN + bias-free Linear (195-201), training step incl. the float32 percentage round trip (241-249),
AdamW hyper-parameters (283-289), Lightning's ``gradient_clip_val: 400`` (configs/an4.yaml:12).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

RNN_CLASSES = {"gru": nn.GRU, "lstm": nn.LSTM, "rnn": nn.RNN}


def out_lengths(lengths):
    """model.py:299-310 for conv1 (k=11, s=2, p=5) and conv2 (k=11, s=1, p=5) on the time axis."""
    L = lengths.cpu().int()
    L = (L + 2 * 5 - 10 - 1) // 2 + 1
    L = (L + 2 * 5 - 10 - 1) // 1 + 1
    return L.int()


class Port:
    """Parameters live in ``self.P`` under the reference's state_dict names; RNN weights are owned by torch RNN modules
    (so that the fused MIOpen / native RNN kernels are the ones that run, exactly as in the reference)."""

    def __init__(self, cfg, state, device="cpu"):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        H, L = cfg["hidden_size"], cfg["hidden_layers"]
        self.bi = bool(cfg["bidirectional"])
        self.rnns = []
        self.P, self.buf = {}, {}
        for k, v in state.items():
            t = torch.as_tensor(v).to(self.device)
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                self.buf[k] = t.clone()
            elif not (k.startswith("rnns.") and ".rnn." in k):
                self.P[k] = t.clone().float().requires_grad_(True)
        for l in range(L):
            # layer 0: 32 channels x the frequency rows after the two convolutions (model.py:166-169: 1312 at 161 bins), from the state
            m = RNN_CLASSES[cfg["rnn_type"]](input_size=int(np.asarray(state["rnns.0.rnn.weight_ih_l0"]).shape[1]) if l == 0 else H,
                                             hidden_size=H, bidirectional=self.bi, bias=True)
            m = m.to(self.device)
            with torch.no_grad():
                for n, p in m.named_parameters():
                    p.copy_(torch.as_tensor(state["rnns.%d.rnn.%s" % (l, n)]))
            for n, p in m.named_parameters():
                self.P["rnns.%d.rnn.%s" % (l, n)] = p
            self.rnns.append(m)

    def parameters(self):
        return list(self.P.values())

    def _bn(self, x, prefix, train):
        return F.batch_norm(x, self.buf[prefix + "running_mean"], self.buf[prefix + "running_var"], self.P[prefix + "weight"],
                            self.P[prefix + "bias"], training=train, momentum=0.1, eps=1e-5)

    def forward(self, x, lengths, train=True, hs=None, return_hs=False):
        """hs / return_hs: the hidden-state carry of reference model.py:224-230 (inference.py:86-96); with return_hs the
        result is (softmax probabilities (N, T', C), lengths, new_hs) as the reference's eval-mode forward returns."""
        P = self.P
        ol = out_lengths(lengths)
        N = x.shape[0]
        y = x
        Tp = (x.shape[3] + 2 * 5 - 10 - 1) // 2 + 1
        keep = (torch.arange(Tp)[None, :] < ol[:, None]).to(x.device).view(N, 1, 1, Tp)
        stages = (
            lambda v: F.conv2d(v, P["conv.seq_module.0.weight"], P["conv.seq_module.0.bias"], stride=(2, 2), padding=(20, 5)),
            lambda v: self._bn(v, "conv.seq_module.1.", train),
            lambda v: F.hardtanh(v, 0.0, 20.0),
            lambda v: F.conv2d(v, P["conv.seq_module.3.weight"], P["conv.seq_module.3.bias"], stride=(2, 1), padding=(10, 5)),
            lambda v: self._bn(v, "conv.seq_module.4.", train),
            lambda v: F.hardtanh(v, 0.0, 20.0),
        )
        for st in stages:
            y = st(y)
            y = y.masked_fill(~keep, 0)
        y = y.reshape(N, y.shape[1] * y.shape[2], Tp).permute(2, 0, 1).contiguous()       # (T', N, 1312)
        new_hs = []
        for l, rnn in enumerate(self.rnns):
            if l > 0:
                T_, N_ = y.shape[0], y.shape[1]
                y = self._bn(y.reshape(T_ * N_, -1), "rnns.%d.batch_norm.module." % l, train).view(T_, N_, -1)
            pk = pack_padded_sequence(y, ol)
            o, h_l = rnn(pk, hs[l] if hs is not None else None)
            new_hs.append(h_l)
            y, _ = pad_packed_sequence(o, total_length=Tp)
            if self.bi:
                y = y.view(y.shape[0], y.shape[1], 2, -1).sum(2)
        if not self.bi:
            w = P["lookahead.0.conv.weight"]
            ctx = w.shape[2]
            z = F.pad(y.permute(1, 2, 0), (0, ctx - 1))
            z = F.conv1d(z, w, groups=w.shape[0])
            y = F.hardtanh(z.permute(2, 0, 1).contiguous(), 0.0, 20.0)
        T_, N_ = y.shape[0], y.shape[1]
        y = self._bn(y.reshape(T_ * N_, -1), "fc.0.module.0.", train)
        y = F.linear(y, P["fc.0.module.1.weight"]).view(T_, N_, -1)
        if return_hs:
            return y.transpose(0, 1).float().softmax(-1), ol, new_hs
        return y, ol                                                                         # logits (T', N, C)

    def training_loss(self, batch):
        inputs, targets, pct, tsz = batch
        sizes = (pct.clone().float() * int(inputs.shape[3])).int()                           # model.py:243
        logits, ol = self.forward(inputs, sizes, train=True)
        lp = logits.float().log_softmax(-1)
        return F.ctc_loss(lp, targets, ol, tsz, blank=0, reduction="sum", zero_infinity=True)

    def make_optimizer(self):
        return torch.optim.AdamW(self.parameters(), lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)

    def train_step(self, batch, opt, autocast_dtype=None):
        """zero_grad -> training_step -> backward -> clip_grad_norm_(400) -> AdamW.step: Lightning's per-batch work."""
        opt.zero_grad(set_to_none=True)
        if autocast_dtype is not None:
            with torch.autocast(self.device.type, dtype=autocast_dtype):
                loss = self.training_loss(batch)
        else:
            loss = self.training_loss(batch)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.parameters(), 400.0)
        opt.step()
        return loss


def random_state(cfg, seed=0, n_classes=29):
    """Reference-shaped random initial state (torch default inits under a fixed seed), as numpy-free torch tensors."""
    g = torch.Generator().manual_seed(seed)
    H, L, bi = cfg["hidden_size"], cfg["hidden_layers"], cfg["bidirectional"]
    G = {"gru": 3, "lstm": 4, "rnn": 1}[cfg["rnn_type"]]

    def U(shape, a):
        return (torch.rand(shape, generator=g) * 2 - 1) * a
    S = {}
    S["conv.seq_module.0.weight"] = U((32, 1, 41, 11), (1.0 / 451) ** 0.5)
    S["conv.seq_module.0.bias"] = U((32,), (1.0 / 451) ** 0.5)
    S["conv.seq_module.3.weight"] = U((32, 32, 21, 11), (1.0 / 7392) ** 0.5)
    S["conv.seq_module.3.bias"] = U((32,), (1.0 / 7392) ** 0.5)

    def bn(prefix, c):
        S[prefix + "weight"], S[prefix + "bias"] = torch.ones(c), torch.zeros(c)
        S[prefix + "running_mean"], S[prefix + "running_var"] = torch.zeros(c), torch.ones(c)
        S[prefix + "num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    bn("conv.seq_module.1.", 32)
    bn("conv.seq_module.4.", 32)
    a = 1.0 / H ** 0.5
    for l in range(L):
        I = 1312 if l == 0 else H
        if l > 0:
            bn("rnns.%d.batch_norm.module." % l, H)
        for suf in [""] + (["_reverse"] if bi else []):
            S["rnns.%d.rnn.weight_ih_l0%s" % (l, suf)] = U((G * H, I), a)
            S["rnns.%d.rnn.weight_hh_l0%s" % (l, suf)] = U((G * H, H), a)
            S["rnns.%d.rnn.bias_ih_l0%s" % (l, suf)] = U((G * H,), a)
            S["rnns.%d.rnn.bias_hh_l0%s" % (l, suf)] = U((G * H,), a)
    if not bi:
        ctx = cfg.get("lookahead_context", 20)
        S["lookahead.0.conv.weight"] = U((H, 1, ctx), (1.0 / ctx) ** 0.5)
    bn("fc.0.module.0.", H)
    S["fc.0.module.1.weight"] = U((n_classes, H), a)
    return S
