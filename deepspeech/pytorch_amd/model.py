"""Drop-in replacement for ``deepspeech_pytorch.model.DeepSpeech`` (reference model.py:138-310) whose forward /
training step run on the hand-written gfx950 kernels of libds2hip.so.

Boundary kept identical to the reference:
  * constructor ``DeepSpeech(labels, model_cfg, precision, optim_cfg, spect_cfg)`` (model.py:139-145)
  * ``forward(x (N,1,161,T), lengths, hs=None) -> (out (N,T',C), output_lengths int32 CPU, new_hs)`` (model.py:214-239);
    logits in train mode, softmax probabilities in eval mode (model.py:72-77)
  * ``training_step`` (sum-reduced CTC, model.py:241-249), ``validation_step``, ``configure_optimizers``, ``get_seq_lens``
  * parameter / buffer names and shapes == the reference ``state_dict`` (checkpoints load both ways, strict=True)
The module tree below exists only to own the parameters under the reference names; none of the torch layers is ever
called.  Kernel-friendly re-layouts of the weights (bf16 copies, transposes, tap-major conv kernels) live in a cache
keyed on the parameters' version counters, never in the state.

There is no CPU / torch-op fallback: CPU tensors or a missing libds2hip.so raise.
"""
import math
from typing import List

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import Ds2HipError

try:  # the Lightning / OmegaConf stack is optional here (absent from the build image); the reference needs it
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover - exercised in the build image
    pl = None

    class _Base(nn.Module):
        """Minimal stand-in for pl.LightningModule when Lightning is absent: what the reference's model and its callers use of it
        -- ``save_hyperparameters()`` (model.py:147: the constructor arguments, stored in ``hparams`` and written into a checkpoint
        as ``hyper_parameters``), ``log`` (model.py:270-271), ``device`` (model.py:254) and ``load_from_checkpoint`` (utils.py:31,
        search_lm_params.py:47-51) on Lightning's ``.ckpt`` layout ``{"state_dict": ..., "hyper_parameters": ...}``."""

        def save_hyperparameters(self, *a, **k):
            import inspect
            frame = inspect.currentframe().f_back
            try:
                names = [n for n in inspect.signature(type(self).__init__).parameters if n != "self"]
                self._hparams = {n: frame.f_locals[n] for n in names if n in frame.f_locals}
            finally:
                del frame

        @property
        def hparams(self):
            return getattr(self, "_hparams", {})

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **overrides):
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(overrides)
            model = cls(**hp)
            model.load_state_dict(ckpt["state_dict"], strict=strict)
            return model

try:
    from omegaconf import OmegaConf

    def _cfg_type_name(cfg):
        t = OmegaConf.get_type(cfg)
        return getattr(t, "__name__", str(t))
except Exception:  # pragma: no cover
    def _cfg_type_name(cfg):
        return type(cfg).__name__

from .configs import AdamConfig, BiDirectionalConfig, SGDConfig, SpectConfig, UniDirectionalConfig, rnn_kind  # noqa: E402

import os  # noqa: E402

# DS2_WGRAD_BESIDE_DX=1: the grouped weight-gradient launch of a layer runs beside its dX GEMM on the second stream (measured and
# rejected as the default: 28.0 vs 27.4 ms per cfg3 step, 143.9 vs 142.4 on cfg5a -- two tile streams through one dispatcher)
WGRAD_BESIDE_DX = os.environ.get("DS2_WGRAD_BESIDE_DX", "0") not in ("", "0")
# the layer's dX product rides in the same launch as its weight gradients (long tiles first, the short dX tiles fill the CUs the
# weight gradients leave idle); DS2_WGRAD_WITH_DX=0 launches them one after the other
WGRAD_WITH_DX = os.environ.get("DS2_WGRAD_WITH_DX", "1") not in ("", "0")

# DS2_COMPOSITE_NODE=1: conv front-end + RNN stack as ONE autograd node (the round-1..3 graph, _FrontFn): the fp32 / small-shape
# backward then keeps its weight-gradient GEMMs on the second stream until the end of the node.  =0: one node per layer always.
# Unset (default): one node per layer in the bf16 performance mode and under any data-parallel wrapper, the composite node otherwise.
COMPOSITE_NODE = {"": None, "0": False}.get(os.environ.get("DS2_COMPOSITE_NODE", ""), True)     # None = decide per shape (see _logits)
# Set by dist.wrap_data_parallel; _data_parallel_active() also sees a wrapper this module was not told about (Lightning's own
# `strategy: ddp` wraps the LightningModule itself): under any data-parallel wrapper every shape gets per-layer nodes, so that the
# reducer receives a layer's gradients when the layer's backward returns.
PER_LAYER_NODES_FOR_DDP = [False]


# DS2_PAD_HIDDEN=0: round the hidden size up to the 16-unit tile only (rounds 3-4); default: up to the nearest width a persistent
# recurrent kernel is instantiated for, when that costs at most 1.5x the units
PAD_HIDDEN = os.environ.get("DS2_PAD_HIDDEN", "1") != "0"


def _padded_hidden(H, kind, precision, bidirectional=None):
    """Internal width of the recurrent stack: `hidden_size` rounded up to the 16-unit MFMA tile -- and, from 200 units on, further up to
    the nearest width the persistent sweeps are instantiated for (bf16: every 128 from 384 to 1280 / 1536, plus 800; fp32: 800 and
    1024), if that is at most 1.5x as wide.  The extra units carry zero weights and biases and stay exactly 0 (GRU, LSTM and tanh
    cells, forward and backward); the padding lives in the weight cache and the activations, never in the state_dict.  Why: a width
    one notch off the instantiated set used to fall to the launch-per-time-step kernels at 5-8x the time per step (the reference
    leaves hidden_size free, train_config.py:49); zero units in a latency-bound sweep cost next to nothing."""
    base = (int(H) + 15) // 16 * 16
    if not PAD_HIDDEN or base < 200 or kind not in ("gru", "lstm", "rnn"):
        return base
    try:
        from ._lib import query
        # compute_dtype()'s rule; a model built for fp32 and run under autocast lands on 800 / 1024, which the bf16 kernels cover too
        dt_ = ops.dt(torch.bfloat16 if str(precision) in ("16", "bf16", "16-mixed", "bf16-mixed") else torch.float32)
        cell = ops.CELLS[kind]
        # the model's own direction count (both when the caller does not say): a width covered only for the other one would pad
        # the units by up to 1.5x and still run one launch per time step
        dirs = (2, 1) if bidirectional is None else ((2,) if bidirectional else (1,))
        for w in range(base, int(base * 1.5) + 1, 16):
            if any(query("ds2_rnn_persist_shape_covered", dt_, cell, D_, 1, w) for D_ in dirs):
                return w
    except Exception:  # noqa: BLE001  (library not built: the constructor must still work on a CPU-only box)
        pass
    return base


def _data_parallel_active():
    if PER_LAYER_NODES_FOR_DDP[0]:
        return True
    try:
        import torch.distributed as _dist
        return _dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1
    except Exception:  # noqa: BLE001
        return False


# fp32 mode: weight gradients as grouped TN products on row-stacked split operands (DS2_FP32_WGRAD_TN=0: round 4's transposes + NT)
FP32_WGRAD_TN = os.environ.get("DS2_FP32_WGRAD_TN", "1") != "0"

# Row lists (_frame_rows): the GEMMs over the [T' x N] frames skip the padding when less than this fraction of the frames is real
# (DS2_ROW_LISTS=0: never; the products then run over every row, padding included, as in rounds 1-3).
ROW_LISTS = os.environ.get("DS2_ROW_LISTS", "1") != "0"
ROW_LIST_MIN_PADDING = 0.97

N_FREQ_CONV2 = 41              # the 16 kHz / 20 ms defaults (161 bins); a model holds its own geometry: _F0/_F1/_F2/_rnn_in/_rnn_ld
RNN_INPUT = 32 * N_FREQ_CONV2  # 1312, model.py:166-169
RNN_INPUT_LD = 1344            # leading dimension of the conv-stack output: 1312 rounded up to the GEMM K-tile (64); pad = 0


def _round64(n):
    return (n + 63) // 64 * 64


# ==================================================================================================================
# parameter containers (reference names; never called)
# ==================================================================================================================
class SequenceWise(nn.Module):  # model.py:18-39 (container only)
    def __init__(self, module):
        super().__init__()
        self.module = module


class MaskConv(nn.Module):  # model.py:42-69 (container only)
    def __init__(self, seq_module):
        super().__init__()
        self.seq_module = seq_module


class _RNNParams(nn.Module):
    """Owns weight_ih_l0 / weight_hh_l0 / bias_ih_l0 / bias_hh_l0 [+ _reverse] exactly like torch.nn.RNNBase
    (same creation order and the same U(-1/sqrt(H), 1/sqrt(H)) init, so identical seeds give identical weights)."""

    def __init__(self, kind, input_size, hidden_size, bidirectional):
        super().__init__()
        g = ops.GATES[kind]
        for suf in [""] + (["_reverse"] if bidirectional else []):
            self.register_parameter("weight_ih_l0" + suf, nn.Parameter(torch.empty(g * hidden_size, input_size)))
            self.register_parameter("weight_hh_l0" + suf, nn.Parameter(torch.empty(g * hidden_size, hidden_size)))
            self.register_parameter("bias_ih_l0" + suf, nn.Parameter(torch.empty(g * hidden_size)))
            self.register_parameter("bias_hh_l0" + suf, nn.Parameter(torch.empty(g * hidden_size)))
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)


class BatchRNN(nn.Module):  # model.py:80-102 (container only)
    def __init__(self, input_size, hidden_size, kind, bidirectional=False, batch_norm=True):
        super().__init__()
        self.input_size, self.hidden_size, self.bidirectional, self.kind = input_size, hidden_size, bidirectional, kind
        self.batch_norm = SequenceWise(nn.BatchNorm1d(input_size)) if batch_norm else None
        self.rnn = _RNNParams(kind, input_size, hidden_size, bidirectional)
        self.num_directions = 2 if bidirectional else 1


class Lookahead(nn.Module):  # model.py:105-135 (container only)
    def __init__(self, n_features, context):
        super().__init__()
        assert context > 0
        self.context, self.n_features = context, n_features
        self.conv = nn.Conv1d(n_features, n_features, kernel_size=context, stride=1, groups=n_features, padding=0, bias=False)


# ==================================================================================================================
# weight cache: kernel layouts derived from the fp32 parameters, rebuilt when a parameter's version changes
# ==================================================================================================================
class _WeightCache:
    """Kernel-layout copies of parameters.  An entry is valid for the parameters' (version, storage) AND the cache epoch:
    the module advances the epoch at every training-mode forward (and at the first eval forward after training), because
    in-place updates that do not bump ``Tensor._version`` exist -- torch's fused optimizers (``AdamW(fused=True)``) are
    one -- and a stale bf16 copy would silently freeze training."""

    def __init__(self):
        self._store = {}
        self.epoch = 0

    def get(self, key, params, builder):
        ver = (self.epoch,) + tuple((p._version, p.data_ptr()) for p in params)
        hit = self._store.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        with torch.no_grad():
            val = builder()
        self._store[key] = (ver, val)
        return val

    def _ver(self, params):
        return (self.epoch,) + tuple((p._version, p.data_ptr()) for p in params)

    def valid(self, key, params):
        hit = self._store.get(key)
        return hit is not None and hit[0] == self._ver(params)

    def put(self, key, params, val, epoch=None):
        ver = self._ver(params)
        if epoch is not None:                       # stamped for a later epoch (the fused optimizer prepares the next step's layouts)
            ver = (epoch,) + ver[1:]
        self._store[key] = (ver, val)

    def clear(self):
        self._store.clear()


def _wgrad_splitk(M, N, K):
    """Split-K factor of a weight-gradient GEMM (long K = T'*N, few output tiles): enough 128x128 tiles x K-slices to
    fill the 256 CUs twice over, every slice keeping >= 32 K-tiles."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 128:         # fills enough of the chip: the DMA-staged kernel without atomics is faster (tools/bench_gemm.py)
        return 1
    return max(1, min(8, 768 // max(tiles, 1), K // (64 * 32)))


def _pad_gate_rows(w, G, H, Hp):
    """[G*H, ...] -> [G*Hp, ...]: every gate block padded with zero rows (hidden units H..Hp-1 of the internal, tile-aligned
    width: zero weights and biases keep them at exactly 0 in every cell type, so they never touch a real unit)."""
    if Hp == H:
        return w
    out = w.new_zeros((G, Hp) + tuple(w.shape[1:]))
    out[:, :H] = w.reshape((G, H) + tuple(w.shape[1:]))
    return out.reshape((G * Hp,) + tuple(w.shape[1:]))


def _pad_cols(w, Cp):
    if w.shape[-1] == Cp:
        return w
    out = w.new_zeros(tuple(w.shape[:-1]) + (Cp,))
    out[..., :w.shape[-1]] = w
    return out


def _unpad_gate_rows(g, G, H, Hp):
    if Hp == H:
        return g
    return g.reshape((G, Hp) + tuple(g.shape[1:]))[:, :H].reshape((G * H,) + tuple(g.shape[1:])).contiguous()


def _bn_seq_fwd(bn, w, b, X, Xh, R, Ct, Cp, training):
    """SequenceWise BatchNorm1d over [R][Cp] rows whose real features are the first Ct columns (model.py:28-33, 86, 196).  Cp > Ct
    only for hidden sizes that are not a multiple of the tile: the zero pad columns get gain 1 / shift 0 (they stay exactly 0 in
    forward and backward) and the running statistics are updated through padded copies."""
    if Ct == Cp:
        return ops.bn_fwd(X, 0, training, w.detach(), b.detach(), bn.running_mean, bn.running_var, bn.num_batches_tracked, R, Ct,
                          X.stride(0), Xh, Xh.stride(0), eps=bn.eps, momentum=bn.momentum)
    g, be = w.new_ones(Cp), w.new_zeros(Cp)
    rm, rv = w.new_zeros(Cp), w.new_ones(Cp)
    g[:Ct], be[:Ct], rm[:Ct], rv[:Ct] = w.detach(), b.detach(), bn.running_mean, bn.running_var
    sv = ops.bn_fwd(X, 0, training, g, be, rm, rv, bn.num_batches_tracked, R, Cp, X.stride(0), Xh, Xh.stride(0), eps=bn.eps,
                    momentum=bn.momentum)
    if training:
        bn.running_mean.copy_(rm[:Ct])
        bn.running_var.copy_(rv[:Ct])
    return sv


def _bn_seq_bwd(G, X, DX, sv, R, Ct, Cp):
    dg, db = ops.bn_bwd(G, X, DX, 0, sv, R, Cp, G.stride(0), X.stride(0), DX.stride(0))
    return (dg, db) if Ct == Cp else (dg[:Ct].contiguous(), db[:Ct].contiguous())


def _perm_cols_to_internal(w):
    """rnns.0 weight_ih columns: reference feature c*F2+f -> internal f*32+c (F2 = columns / 32: 41 at 161 bins), zero-padded to the
    GEMM K-tile (1312 -> 1344)."""
    g, rin = w.shape
    out = w.new_zeros((g, _round64(rin)))
    out[:, :rin] = w.reshape(g, 32, rin // 32).permute(0, 2, 1).reshape(g, rin)
    return out


def _perm_cols_to_reference(w, rin=RNN_INPUT):
    g = w.shape[0]
    return w[:, :rin].reshape(g, rin // 32, 32).permute(0, 2, 1).reshape(g, rin)


# ==================================================================================================================
# autograd stages
# ==================================================================================================================
class _ConvStackFn(torch.autograd.Function):
    """MaskConv over conv1-BN-Hardtanh-conv2-BN-Hardtanh (model.py:157-164, 53-69) + collapse/transpose (219-221).
    Output: X0 [(t*N+n)][1344]: 1312 features in the internal order f*32+c + 32 zero pad columns (GEMM K-tile)."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, g2, be2, mod, lens_dev, Tp, dtype, training):
        N, T = x.shape[0], x.shape[3]
        c = mod._cache
        bn1, bn2 = mod.conv.seq_module[1], mod.conv.seq_module[4]
        w1k = c.get("w1k", [w1], lambda: w1.detach().reshape(32, 451).t().contiguous())
        w2t = c.get(("w2t", dtype), [w2], lambda: w2.detach().permute(2, 3, 0, 1).contiguous().to(dtype))
        x = x.contiguous().float()
        F0, F1, F2, rin, rld = mod._F0, mod._F1, mod._F2, mod._rnn_in, mod._rnn_ld
        if x.shape[2] != F0:
            raise ValueError("this model was built for %d frequency bins (SpectConfig: sample_rate * window_size / 2 + 1), got input "
                             "with %d" % (F0, x.shape[2]))
        y1 = ops.conv1_fwd(x, w1k, b1.detach(), lens_dev, Tp, dtype)
        R1 = N * F1 * Tp
        a1 = torch.empty_like(y1)
        sv1 = ops.bn_fwd(y1, 1, training, g1.detach(), be1.detach(), bn1.running_mean, bn1.running_var,
                         bn1.num_batches_tracked, R1, 32, 32, a1, 32, F=F1, Tp=Tp, N=N, lens=lens_dev, eps=bn1.eps,
                         momentum=bn1.momentum)
        y2 = ops.conv2_fwd(a1, w2t, b2.detach(), lens_dev, F0)
        R2 = N * F2 * Tp
        x0 = torch.empty((Tp * N, rld), dtype=dtype, device=x.device)
        if rld > rin:
            x0[:, rin:].zero_()
        sv2 = ops.bn_fwd(y2, 2, training, g2.detach(), be2.detach(), bn2.running_mean, bn2.running_var,
                         bn2.num_batches_tracked, R2, 32, 32, x0, rld, F=F2, Tp=Tp, N=N, lens=lens_dev, eps=bn2.eps,
                         momentum=bn2.momentum)
        ctx.mod, ctx.dims, ctx.sv = mod, (N, T, Tp, dtype), (sv1, sv2)
        ctx.save_for_backward(x, y1, a1, y2, lens_dev, w2)
        return x0

    @staticmethod
    def backward(ctx, dx0):
        x, y1, a1, y2, lens_dev, w2 = ctx.saved_tensors
        N, T, Tp, dtype = ctx.dims
        sv1, sv2 = ctx.sv
        c = ctx.mod._cache
        dx0 = dx0.contiguous().to(dtype)
        F0, F1, F2, rld = ctx.mod._F0, ctx.mod._F1, ctx.mod._F2, ctx.mod._rnn_ld
        R1, R2 = N * F1 * Tp, N * F2 * Tp
        dy2 = torch.empty_like(y2)
        dg2, dbe2 = ops.bn_bwd(dx0, y2, dy2, 2, sv2, R2, 32, rld, 32, 32, F=F2, Tp=Tp, N=N, lens=lens_dev)
        w2d = c.get(("w2d", dtype), [w2], lambda: [
            w2.detach()[:, :, q::2, :].flip(2, 3).permute(2, 3, 1, 0).contiguous().to(dtype) for q in (0, 1)])
        da1 = ops.conv2_dgrad(dy2, w2d[0], w2d[1], F0)
        dy1 = torch.empty_like(y1)
        dg1, dbe1 = ops.bn_bwd(da1, y1, dy1, 1, sv1, R1, 32, 32, 32, 32, F=F1, Tp=Tp, N=N, lens=lens_dev)
        db1 = ops.colsum(dy1.view(R1, 32))
        dw1k = ops.conv1_wgrad(x, dy1, Tp)
        dw1 = dw1k.t().reshape(32, 1, 41, 11).contiguous()
        # conv2's parameter gradients are off the dependent chain (dgrad -> BatchNorm backward -> conv1 weight gradient), so last;
        # on this stream: the second one is busy with the layer-0 weight-gradient GEMMs of the RNN stack for longer than this one
        db2 = ops.colsum(dy2.view(R2, 32))
        dw2 = ops.conv2_wgrad(dy2, a1, F0).view(21, 11, 32, 32).permute(2, 3, 0, 1).contiguous()
        return (None, dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2, None, None, None, None, None)


def _layer_param_count(layer):
    return (2 if layer.batch_norm is not None else 0) + 4 * layer.num_directions


_ROWS_STAGING, _CTC_STAGING = ops.PinnedRing(), ops.PinnedRing()


def _frame_rows(output_lengths, Tp, N, dev):
    """(lens_dev, rows): the output lengths on the device and -- when the batch carries enough padding to matter -- the list of the
    rows t*N + n with t < length[n] of a [T' x N] sequence matrix, in storage order: the frames pack_padded_sequence keeps
    (model.py:96).  The dense products over the frames (input projections, dX, weight gradients) visit only these rows.  One host
    buffer, one copy: [lengths | rows]."""
    ol = output_lengths.numpy().astype(np.int32)
    n_valid = int(np.minimum(ol, Tp).sum())
    if not ROW_LISTS or n_valid >= ROW_LIST_MIN_PADDING * Tp * N or n_valid == 0:
        return output_lengths.to(dev, torch.int32, non_blocking=True), None
    rows = np.flatnonzero(np.arange(Tp, dtype=np.int32)[:, None] < ol[None, :]).astype(np.int32)
    # staged in pinned memory (an asynchronous copy must not read a pageable temporary that is freed on return from .to()), in a
    # slot of a ring allocated once: see ops.PinnedRing for what a fresh pinned allocation per step costs the kernel in flight
    both = _ROWS_STAGING.stage(dev, [ol, rows])
    return both[:N], both[N:]


def _rnn_layer_forward(mod, li, X, lens_dev, N, Tp, dtype, training, h0, c0, lparams, rows=None):
    """One BatchRNN layer (model.py:94-102): [SequenceWise BatchNorm1d] -> input projection GEMM -> packed bi/uni GRU/LSTM/RNN
    sweep -> direction sum.  lparams: [bn.weight, bn.bias]? + per direction (weight_ih, weight_hh, bias_ih, bias_hh).
    rows: the batch's row list (_frame_rows) -- the input projection then skips the padding frames, whose gate pre-activations
    no sweep reads (pack_padded_sequence, model.py:96).
    Returns (out [T'N][H], hn, cn, saved = [X, Xh or None, hext, Sv], meta)."""
    c = mod._cache
    R = Tp * N
    layer = mod.rnns[li]
    kind, Ht, D = layer.kind, layer.hidden_size, layer.num_directions
    H = mod._Hp                                         # internal width: hidden_size rounded up to the 16-unit MFMA tile
    G = ops.GATES[kind]
    first = li == 0
    I = mod._rnn_ld if first else H                     # layer 0 sees the zero-padded conv-stack output
    has_bn = layer.batch_norm is not None
    bn_w, bn_b = (lparams[0], lparams[1]) if has_bn else (None, None)
    wts = lparams[2 if has_bn else 0:]
    wih = [wts[4 * d + 0] for d in range(D)]
    whh = [wts[4 * d + 1] for d in range(D)]
    bih = [wts[4 * d + 2] for d in range(D)]
    bhh = [wts[4 * d + 3] for d in range(D)]

    def build_ih():
        w = torch.cat([_pad_gate_rows(p.detach(), G, Ht, H) for p in wih], 0)
        w = _perm_cols_to_internal(w) if first else _pad_cols(w, H)
        out = ops.empty_padded(w.shape[0], w.shape[1], dtype, w.device)
        out.copy_(w)
        return out
    Wih = c.get(("wih", li, dtype), wih, build_ih)
    Whh = c.get(("whh", li, dtype), whh, lambda: torch.stack(
        [_pad_gate_rows(_pad_cols(p.detach(), H), G, Ht, H) for p in whh], 0).to(dtype).contiguous())
    if H == Ht:
        Bih, Bhh = mod._bias_views(li)
    else:
        Bih = c.get(("bih", li), bih, lambda: torch.cat([_pad_gate_rows(p.detach(), G, Ht, H) for p in bih], 0))
        Bhh = c.get(("bhh", li), bhh, lambda: torch.stack([_pad_gate_rows(p.detach(), G, Ht, H) for p in bhh], 0))
    sv, Xh = None, X
    if has_bn:
        bn = layer.batch_norm.module
        Xh = ops.empty_padded(R, I, X.dtype, X.device)       # row stride off the power of two (GEMM operand)
        sv = _bn_seq_fwd(bn, bn_w, bn_b, X, Xh, R, layer.input_size, I, training)
    if ops.split3_ok(dtype, R, D * G * H, I):
        # fp32 mode: one bf16 GEMM over the split operands (ops.split3); the split weights are cached like every other layout
        Wih3 = c.get(("wih3", li), wih, lambda: ops.split3(Wih, 1))
        GI = ops.gemm_nt(ops.split3(Xh, 0), Wih3, bias=Bih, out_dtype=torch.float32, rows=rows)
    else:
        GI = ops.gemm_nt(Xh, Wih, bias=Bih, rows=rows)                   # [R][D*G*H]; padding rows unwritten with a row list
    if H != Ht:
        h0 = _pad_cols(h0, H) if h0 is not None else None
        c0 = _pad_cols(c0, H) if c0 is not None else None
    hext, Sv, hn, cn = ops.rnn_fwd(kind, GI, Whh, Bhh, lens_dev, D, N, H, Tp, h0=h0, c0=c0)
    if H != Ht:
        hn = hn[..., :Ht].contiguous()
        cn = cn[..., :Ht].contiguous() if cn is not None else None
    del GI
    if D == 2:
        out = ops.add2(hext[0, 1:Tp + 1].reshape(R, H), hext[1, 1:Tp + 1].reshape(R, H))
    else:
        out = hext[0, 1:Tp + 1].reshape(R, H)      # one direction: the layer's output IS the stored state sequence (a contiguous view;
                                                   # rounds 1-5 cloned it: seven 123 MB copies per cfg5b step).  Nothing writes into it.
    return out, hn, cn, [X, Xh if has_bn else None, hext, Sv], (kind, H, D, G, I, has_bn, sv, Ht, layer.input_size)


def _rnn_layer_backward(mod, li, meta, saved, lparams, dout, lens_dev, N, Tp, dtype, main, side, wgrad_done, rows=None, state=None):
    """state = (h0, c0, want_dstate) when the forward was given an initial state (`hs`, reference model.py:224-230) -- the rare path:
    the BPTT then runs on the launch-per-time-step kernels (ds2_rnn_bwd takes h0 / c0 and returns d h0 / d c0), and the first steps'
    share of dW_hh -- h_{t-1} = h0 is not a row of the stored state sequence -- is added here.  Returns (dX, grads, wgrad_done[, dstate])."""
    if state is None:
        return _rnn_layer_backward_impl(mod, li, meta, saved, lparams, dout, lens_dev, N, Tp, dtype, main, side, wgrad_done, rows)
    kind, H, D, G, I, has_bn, sv, Ht, It = meta
    h0, c0, want = state
    h0p = _pad_cols(h0, H) if (h0 is not None and H != Ht) else h0
    c0p = _pad_cols(c0, H) if (c0 is not None and H != Ht) else c0
    cap = {"h0": h0p, "c0": c0p, "want": bool(want)}
    dX, grads, wgrad_done = _rnn_layer_backward_impl(mod, li, meta, saved, lparams, dout, lens_dev, N, Tp, dtype, main, side, wgrad_done,
                                                     rows, cap)
    if wgrad_done is not None:
        main.wait_event(wgrad_done)
    main.wait_stream(side)
    rg, GH, po = cap["rg"], G * H, (2 if has_bn else 0)
    if h0p is not None:
        n4 = (N + 3) // 4 * 4
        ar = torch.arange(N, device=lens_dev.device)
        for d in range(D):
            t_first = torch.zeros_like(ar) if d == 0 else (lens_dev.long().clamp(1, Tp) - 1)
            src = rg.dGH[d].reshape(Tp * N, GH) if (kind == "gru" and rg.dGH is not None) else rg.dGI[:, d * GH:(d + 1) * GH]
            g_first = src.index_select(0, t_first * N + ar).float()                     # [N][G*H]: the hidden-side gate gradient
            A = torch.zeros((GH, n4), dtype=torch.float32, device=g_first.device)         # of every clip's first step
            A[:, :N] = g_first.t()
            B = torch.zeros((H, n4), dtype=torch.float32, device=g_first.device)
            B[:, :N] = h0p[d].t()
            corr = ops.gemm_nt(A, B, out_dtype=torch.float32)                             # [G*H][H] = sum_n g_first[n]^T (x) h0[n]
            grads[po + 4 * d + 1] = grads[po + 4 * d + 1] + _unpad_gate_rows(corr, G, Ht, H)[:, :Ht]
    dstate = None
    if want:
        dstate = (rg.dh0[..., :Ht].contiguous(), rg.dc0[..., :Ht].contiguous() if rg.dc0 is not None else None)
    return dX, grads, None, dstate


def _rnn_layer_backward_impl(mod, li, meta, saved, lparams, dout, lens_dev, N, Tp, dtype, main, side, wgrad_done, rows=None, cap=None):
    """Backward of one BatchRNN layer.  bf16 (the performance mode): the BPTT sweep, then ONE launch with the layer's weight
    gradients (grouped TN products over the activations as stored) and its dX, then the BatchNorm backward -- all on the caller's
    stream; nothing runs beside a sweep (a co-runner costs the latency-bound sweep the chip's clock).  fp32 /
    small / odd shapes: the round-2 path (operand transposes + 128x128 GEMMs on the second HIP stream `side`; the caller joins the
    streams before the gradients reach autograd).  Returns (dX, gradients in the order of lparams, wgrad_done event or None)."""
    c = mod._cache
    R = Tp * N
    kind, H, D, G, I, has_bn, sv, Ht, It = meta
    X, Xh, hext, Sv = saved
    if Xh is None:
        Xh = X
    first = li == 0
    GH = G * H
    grads = [None] * len(lparams)
    po = 2 if has_bn else 0
    wts = lparams[po:po + 4 * D]
    wih = [wts[4 * d + 0] for d in range(D)]
    whh = [wts[4 * d + 1] for d in range(D)]
    WhhT = c.get(("whhT", li, dtype), list(whh), lambda: torch.stack(
        [_pad_gate_rows(_pad_cols(p.detach(), H), G, Ht, H).t() for p in whh], 0).to(dtype).contiguous())   # [D][H][G*H]

    def build_ihT():
        w = torch.cat([_pad_gate_rows(p.detach(), G, Ht, H) for p in wih], 0)
        w = _perm_cols_to_internal(w) if first else _pad_cols(w, H)
        return w.t().to(dtype).contiguous()                                                             # [I][D*G*H]
    WihT = c.get(("wihT", li, dtype), list(wih), build_ihT)
    # ---- BPTT sweep (caller's stream).  A persistent sweep wants every CU: it starts after the weight-gradient launch of the
    # layer above has drained (otherwise its first workgroups would spin on peers that are still waiting for a CU)
    if wgrad_done is not None:
        main.wait_event(wgrad_done)
        wgrad_done = None
    fast = ops.wgrad_tn_ok(dtype, R, D * GH, Xh.shape[1], lda=D * GH, ldb=Xh.stride(0)) and \
        ops.wgrad_tn_ok(dtype, R, GH, H, lda=D * GH, ldb=H)
    # one launch for the weight gradients + dX (below): every reader of the sweep's dGI / dQ then works from the row list
    n_probs = 1 + D * (2 if (kind == "gru" and (2 * H) % 256 != 0) else 1)
    fuse_dx = fast and WGRAD_WITH_DX and not WGRAD_BESIDE_DX and n_probs < 6 and WihT.dim() == 2 and WihT.data_ptr() % 16 == 0 and \
        ops.gemm8_nt_shape_ok(dtype, R, WihT.shape[0], D * GH, D * GH, WihT.stride(0))
    if cap is None:
        rg = ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens_dev, D, N, H, Tp, pad_rows_unread=bool(fuse_dx and rows is not None))
    else:
        rg = ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens_dev, D, N, H, Tp, h0=cap["h0"], c0=cap["c0"], want_dstate=cap["want"])
        cap["rg"] = rg
    dGI = rg.dGI
    dXh_fused = None
    if fast:
        # ---- weight gradients as ONE grouped launch of TN products (contraction over the T'*N rows, both operands as the
        # activations are stored: no transposes).  Never under a sweep: what co-resident GEMMs cost the sweeps was the
        # chip's CLOCK (1.7 instead of 2.15 GHz while they ran, profiles/r03a_coresidency3.txt) -- a latency-bound kernel
        # pays that one to one.  On the caller's stream, right behind the sweep (optionally beside the dX GEMM on the second
        # stream, WGRAD_BESIDE_DX: no gain measured).
        swept = torch.cuda.Event()
        swept.record(main)
        wstream = side if WGRAD_BESIDE_DX else main
        with torch.cuda.stream(wstream):
            if wstream is not main:
                wstream.wait_event(swept)
                for t_ in rg.tensors() + [Xh, hext]:
                    t_.record_stream(wstream)
            Iw = Xh.shape[1]
            dWih = torch.empty((D * GH, Iw), dtype=torch.float32, device=dGI.device)
            dWhh = torch.empty((D, GH, H), dtype=torch.float32, device=dGI.device)
            probs = [dict(At=dGI, Bt=Xh, M=D * GH, N=Iw, lda=D * GH, ldb=Xh.stride(0), out=dWih)]
            for d in range(D):
                # h_{t-1} of the forward direction is slot t of the guarded buffer, h_{t+1} of the reverse direction slot t+2
                hprev = (hext[d, 0:Tp] if d == 0 else hext[d, 2:Tp + 2]).reshape(R, H)
                if kind == "gru" and rg.dQ is not None:
                    # hidden-side gate gradient = [dr, dz (columns of dGI) | dQ]
                    if (2 * H) % 256 == 0:
                        probs.append(dict(At=dGI[:, d * GH:], At2=rg.dQ[d], lda2=H, m_split=2 * H, Bt=hprev, M=GH, N=H, lda=D * GH, ldb=H,
                                          out=dWhh[d]))
                    else:
                        probs.append(dict(At=dGI[:, d * GH:], Bt=hprev, M=2 * H, N=H, lda=D * GH, ldb=H, out=dWhh[d, :2 * H]))
                        probs.append(dict(At=rg.dQ[d], Bt=hprev, M=H, N=H, lda=H, ldb=H, out=dWhh[d, 2 * H:]))
                elif kind == "gru":
                    probs.append(dict(At=rg.dGH[d], Bt=hprev, M=GH, N=H, lda=GH, ldb=H, out=dWhh[d]))
                else:
                    probs.append(dict(At=dGI[:, d * GH:], Bt=hprev, M=GH, N=H, lda=D * GH, ldb=H, out=dWhh[d]))
            assert not fuse_dx or ops.gemm8_nt_ok(dGI, WihT, R, WihT.shape[0], D * GH, D * GH, WihT.stride(0))
            if fuse_dx:
                # + the layer's dX in the same launch; with a row list the padding frames are neither contracted over nor computed
                _, dXh_fused = ops.gemm8_tn_grouped(probs, R, dx=(dGI, WihT), rows=rows, zero_pad=(lens_dev, Tp, N))
            else:
                ops.gemm8_tn_grouped(probs, R, rows=rows)
            if first:
                dWih = torch.cat([_perm_cols_to_reference(dWih[d * GH:(d + 1) * GH], mod._rnn_in) for d in range(D)], 0)
            if rg.bacc is not None:
                dBih, dBhh_all = ops.rnn_bias_grads(kind, rg.bacc, D, N, H)          # one launch: [D*G*H], [D][G*H]
                dBhh_l = [dBhh_all[d] for d in range(D)]
            else:
                dBih = ops.colsum(dGI)
                dBhh_all = None
                if kind == "gru":
                    dBhh_l = [ops.colsum(rg.dGH[d].reshape(R, GH)) for d in range(D)]
                else:
                    dBhh_l = [dBih[d * GH:(d + 1) * GH] for d in range(D)]
            for d in range(D):
                grads[po + 4 * d:po + 4 * d + 4] = [dWih[d * GH:(d + 1) * GH], dWhh[d], dBih[d * GH:(d + 1) * GH], dBhh_l[d]]
            sync = getattr(mod, "_grad_sync", None)
            if sync is not None and H == Ht:
                own = [dWih, dWhh, dBih] + ([dBhh_all] if dBhh_all is not None else
                                            [dBhh_l[d] for d in range(D)] if kind == "gru" else [])
                sync.layer_ready(own, lparams[po:po + 4 * D])
            _unpad_layer_grads(grads, po, D, G, Ht, H, It, first)
            if wstream is not main:
                wgrad_done = torch.cuda.Event()
                wgrad_done.record(wstream)
    # ---- dependent chain (caller's stream): dX -> BatchNorm backward
    s3 = ops.split3_ok(dtype, GH, H, R, leaf=True)      # fp32 mode: the weight-gradient GEMMs on split operands (ops.split3_ok)
    if dXh_fused is not None:
        dXh = dXh_fused
    elif ops.split3_ok(dtype, R, H, GH):
        WihT3 = c.get(("wihT3", li), list(wih), lambda: ops.split3(WihT, 1))
        dXh = ops.gemm_nt(ops.split3(dGI, 0), WihT3, out_dtype=torch.float32, rows=rows, zero_pad=(lens_dev, Tp, N))
    else:
        dXh = ops.gemm_nt(dGI, WihT, rows=rows, zero_pad=(lens_dev, Tp, N))                             # [R][I]
    if has_bn:
        dX = torch.empty_like(dXh)
        grads[0], grads[1] = _bn_seq_bwd(dXh, X, dX, sv, R, It, I)
    else:
        dX = dXh
    if fast:
        return dX, grads, wgrad_done
    # the second stream starts this layer's weight gradients only once dX / BatchNorm backward are through, i.e. together
    # with the next layer's sweep: the dependent chain never competes with them for the CUs
    ready = torch.cuda.Event()
    ready.record(main)
    # ---- weight gradients (second stream, under the next layer's sweep): contraction over the T'*N rows.  Layer 0's run
    # under the conv backward instead, where no persistent sweep needs most of every CU's registers: the full-size tiles
    cores = li > 0
    with torch.cuda.stream(side):
        side.wait_event(ready)
        for t in rg.tensors() + [Xh, hext]:
            t.record_stream(side)
        if s3 and FP32_WGRAD_TN and Xh.shape[1] % 8 == 0 and ops.wgrad_tn_ok(torch.bfloat16, 3 * R, D * GH, Xh.shape[1]) and \
                ops.wgrad_tn_ok(torch.bfloat16, 3 * R, GH, H) and (kind != "gru" or rg.dQ is not None or rg.dGH is not None):
            # Round 6: the fp32-mode weight gradients as ONE grouped launch of TN products over the activations as stored, on
            # row-stacked split operands ([hi; hi; lo] x [hi; lo; hi]: ops.split3_rows) -- no operand transposes, no K-segment
            # copies, no concatenations (they were 6 transposes + 6 splits + 3 torch cat / copy kernels per layer)
            Iw = Xh.shape[1]
            A3 = ops.split3_rows(dGI, 0)                        # [3R][D*G*H]
            dWih = torch.empty((D * GH, Iw), dtype=torch.float32, device=dGI.device)
            dWhh_all = torch.empty((D, GH, H), dtype=torch.float32, device=dGI.device)
            X3 = ops.split3_rows(Xh, 1)                         # [3R][I]
            probs = [dict(At=A3, Bt=X3, M=D * GH, N=Iw, lda=A3.stride(0), ldb=X3.stride(0), out=dWih)]
            for d in range(D):
                hprev = (hext[d, 0:Tp] if d == 0 else hext[d, 2:Tp + 2]).reshape(R, H)
                H3 = ops.split3_rows(hprev, 1)
                if kind == "gru" and rg.dQ is not None:         # hidden-side gate gradient = [dr, dz (columns of dGI) | dQ]
                    probs.append(dict(At=A3[:, d * GH:], Bt=H3, M=2 * H, N=H, lda=A3.stride(0), ldb=H3.stride(0), out=dWhh_all[d, :2 * H]))
                    Q3 = ops.split3_rows(rg.dQ[d].reshape(R, H), 0)
                    probs.append(dict(At=Q3, Bt=H3, M=H, N=H, lda=Q3.stride(0), ldb=H3.stride(0), out=dWhh_all[d, 2 * H:]))
                elif kind == "gru":
                    G3 = ops.split3_rows(rg.dGH[d].reshape(R, GH), 0)
                    probs.append(dict(At=G3, Bt=H3, M=GH, N=H, lda=G3.stride(0), ldb=H3.stride(0), out=dWhh_all[d]))
                else:
                    probs.append(dict(At=A3[:, d * GH:], Bt=H3, M=GH, N=H, lda=A3.stride(0), ldb=H3.stride(0), out=dWhh_all[d]))
            ops.gemm8_tn_grouped(probs, 3 * R)
            if first:
                dWih = torch.cat([_perm_cols_to_reference(dWih[d * GH:(d + 1) * GH], mod._rnn_in) for d in range(D)], 0)
            if rg.bacc is not None:
                dBih, dBhh_all = ops.rnn_bias_grads(kind, rg.bacc, D, N, H)
                dBhh_l = [dBhh_all[d] for d in range(D)]
            else:
                dBih = ops.colsum(dGI)
                dBhh_all = None
                dBhh_l = [ops.colsum(rg.dGH[d].reshape(R, GH)) for d in range(D)] if kind == "gru" else \
                    [dBih[d * GH:(d + 1) * GH] for d in range(D)]
            for d in range(D):
                grads[po + 4 * d:po + 4 * d + 4] = [dWih[d * GH:(d + 1) * GH], dWhh_all[d], dBih[d * GH:(d + 1) * GH], dBhh_l[d]]
            sync = getattr(mod, "_grad_sync", None)
            if sync is not None and H == Ht:
                own = [dWih, dWhh_all, dBih] + ([dBhh_all] if dBhh_all is not None else
                                                [dBhh_l[d] for d in range(D)] if kind == "gru" else [])
                sync.layer_ready(own, lparams[po:po + 4 * D])
            _unpad_layer_grads(grads, po, D, G, Ht, H, It, first)
            return dX, grads, None
        dGI_T = ops.transpose(dGI)                              # [D*G*H][ldT]
        Xh_T = ops.transpose(Xh)                                # [I][ldT]
        ldT = dGI_T.shape[1]
        dGI_T3 = ops.split3(dGI_T, 0) if s3 else None           # [D*G*H][3*ldT]
        if s3:
            dWih = ops.gemm_nt(dGI_T3, ops.split3(Xh_T, 1), out_dtype=torch.float32, coresident=cores)
        else:
            dWih = ops.gemm_nt(dGI_T, Xh_T, out_dtype=torch.float32, splitk=_wgrad_splitk(dGI_T.shape[0], Xh_T.shape[0], ldT),
                               coresident=cores)
        del Xh_T
        if first:
            dWih = torch.cat([_perm_cols_to_reference(dWih[d * GH:(d + 1) * GH], mod._rnn_in) for d in range(D)], 0)
        # bias gradients: from the sweep's own per-sample sums (persistent kernels: a [N][NB*H] reduction instead of
        # column sums over the T'*N rows of dGI / dGH), else column sums
        bsum = None
        if rg.bacc is not None:
            bsum = [ops.colsum(rg.bacc[d]) for d in range(D)]                   # [NB*H] per direction
            dBih = torch.cat([b[:GH] for b in bsum], 0)
        else:
            dBih = ops.colsum(dGI)
        for d in range(D):
            # h_{t-1} of the forward direction is slot t of the guarded buffer, h_{t+1} of the reverse direction slot t+2
            hprev = hext[d, 0:Tp] if d == 0 else hext[d, 2:Tp + 2]
            Hp_T = ops.transpose(hprev.reshape(R, H))           # [H][ldT]
            if kind == "gru" and rg.dQ is not None:
                # hidden-side gate gradient = [dr, dz (rows of dGI^T) | dQ^T]: one GEMM whose A operand is two row blocks
                dQ_T = ops.transpose(rg.dQ[d].reshape(R, H))   # [H][ldT]
                if s3:
                    A3 = torch.empty((GH, 3 * ldT), dtype=torch.bfloat16, device=dGI.device)
                    A3[:2 * H].copy_(dGI_T3[d * GH:d * GH + 2 * H])
                    ops.split3(dQ_T, 0, out=A3[2 * H:])
                    dWhh = ops.gemm_nt(A3, ops.split3(Hp_T, 1), out_dtype=torch.float32, coresident=cores)
                else:
                    dWhh = ops.gemm_nt_rows2(dGI_T[d * GH:d * GH + 2 * H], dQ_T, 2 * H, Hp_T, GH, H, ldT, ldT, ldT,
                                             splitk=_wgrad_splitk(GH, H, ldT), coresident=cores)
                dBhh = torch.cat([bsum[d][:2 * H], bsum[d][3 * H:4 * H]], 0) if bsum is not None else \
                    torch.cat([dBih[d * GH:d * GH + 2 * H], ops.colsum(rg.dQ[d].reshape(R, H))], 0)
            else:
                if kind == "gru":
                    dGH_T = ops.transpose(rg.dGH[d].reshape(R, GH))    # [G*H][ldT]
                    dBhh = ops.colsum(rg.dGH[d].reshape(R, GH))
                else:
                    dGH_T = dGI_T[d * GH:(d + 1) * GH]
                    dBhh = dBih[d * GH:(d + 1) * GH]
                if s3:
                    A3 = ops.split3(dGH_T, 0) if kind == "gru" else dGI_T3[d * GH:(d + 1) * GH]
                    dWhh = ops.gemm_nt(A3, ops.split3(Hp_T, 1), out_dtype=torch.float32, coresident=cores)
                else:
                    dWhh = ops.gemm_nt(dGH_T, Hp_T, out_dtype=torch.float32, M=GH, N=H, K=ldT, lda=ldT, ldb=ldT,
                                       splitk=_wgrad_splitk(GH, H, ldT), coresident=cores)
            grads[po + 4 * d:po + 4 * d + 4] = [dWih[d * GH:(d + 1) * GH].contiguous(), dWhh,
                                                dBih[d * GH:(d + 1) * GH].contiguous(), dBhh.contiguous()]
        del dGI_T
        sync = getattr(mod, "_grad_sync", None)
        if sync is not None and H == Ht:
            # data parallel with the opt-in early hand-off (dist.OverlappedGradSync): this layer's gradients start
            # their all-reduce now, ordered after the GEMMs above, under the sweeps of the layers below.  The
            # returned gradients are views of dWih / dBih (and of dBih for the biases of LSTM / RNN cells).
            own = [dWih, dBih] + [grads[po + 4 * d + 1] for d in range(D)]
            if kind == "gru":
                own += [grads[po + 4 * d + 3] for d in range(D)]
            sync.layer_ready(own, lparams[po:po + 4 * D])
        _unpad_layer_grads(grads, po, D, G, Ht, H, It, first)
    return dX, grads, None


class _RnnLayerFn(torch.autograd.Function):
    """ONE BatchRNN layer as an autograd node (model.py:94-102).  The default graph is conv stack -> L of these -> [lookahead] ->
    head -> CTC: a layer's parameter gradients reach their AccumulateGrad nodes -- and with them the hooks of
    ``DistributedDataParallel``'s reducer, i.e. the bucketed RCCL all-reduce -- the moment the layer's backward returns, while
    the BPTT sweeps of the layers below are still to run: Lightning's ``strategy: ddp`` (configs/librispeech.yaml:14) overlaps
    the gradient exchange with backward as it does for the reference's nn.GRU layers."""

    @staticmethod
    def forward(ctx, X, mod, lens_dev, N, Tp, dtype, training, li, h0, c0, *lparams):
        ctx.rows = mod._frame_rows                   # the row list of the batch in flight (set by _logits)
        out, hn, cn, saved, meta = _rnn_layer_forward(mod, li, X, lens_dev, N, Tp, dtype, training, h0, c0, lparams, rows=ctx.rows)
        ctx.mod, ctx.dims, ctx.meta, ctx.li = mod, (N, Tp, dtype), meta, li
        ctx.had_state = h0 is not None or c0 is not None
        ctx.h0, ctx.c0 = h0, c0                      # (small [D][N][H] tensors; None in every ordinary training step)
        ctx.none_mask = [t is None for t in saved]
        ctx.save_for_backward(lens_dev, *[t for t in saved if t is not None], *lparams)
        ctx.n_saved = sum(1 for t in saved if t is not None)
        outs = (out, hn) + ((cn,) if cn is not None else ())
        ctx.mark_non_differentiable(*outs[1:])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, dout, *unused):
        N, Tp, dtype = ctx.dims
        mod = ctx.mod
        st = ctx.saved_tensors
        it = iter(st[1:1 + ctx.n_saved])
        saved = [None if is_none else next(it) for is_none in ctx.none_mask]
        lparams = st[1 + ctx.n_saved:]
        main = torch.cuda.current_stream()
        side = mod._wgrad_stream(dout.device)
        dstate = None
        if ctx.had_state:
            # round 6: backward through a forward that was given `hs` (reference model.py:224-230; the reference itself only passes
            # hs in inference) -- launch-per-time-step BPTT with the initial state, d h0 / d c0 returned to autograd
            want = bool(ctx.needs_input_grad[8] or (len(ctx.needs_input_grad) > 9 and ctx.needs_input_grad[9]))
            dX, grads, wgrad_done, dstate = _rnn_layer_backward(mod, ctx.li, ctx.meta, saved, lparams, dout.contiguous().to(dtype), st[0],
                                                                N, Tp, dtype, main, side, None, rows=ctx.rows,
                                                                state=(ctx.h0, ctx.c0, want))
        else:
            dX, grads, wgrad_done = _rnn_layer_backward(mod, ctx.li, ctx.meta, saved, lparams, dout.contiguous().to(dtype), st[0], N, Tp,
                                                        dtype, main, side, None, rows=ctx.rows)
        # the gradients leave this node now (AccumulateGrad, DDP's reducer): whatever the second stream produced must be complete.
        # The early all-reduces of the opt-in OverlappedGradSync are NOT waited for here -- that is their point -- but at the end
        # of backward (its finish callback)
        if wgrad_done is not None:
            main.wait_event(wgrad_done)
        main.wait_stream(side)
        for g in grads:
            if g is not None:
                g.record_stream(main)
        sync = getattr(mod, "_grad_sync", None)
        if sync is not None:
            grads = sync.defer_early(lparams, grads)      # in flight on RCCL's stream: they become .grad at the end of backward
        dh0 = dstate[0] if (dstate is not None and ctx.needs_input_grad[8]) else None
        dc0 = dstate[1] if (dstate is not None and ctx.c0 is not None and ctx.needs_input_grad[9]) else None
        return (dX, None, None, None, None, None, None, None, dh0, dc0, *grads)


class _RnnStackFn(torch.autograd.Function):
    """The whole stack of BatchRNN layers (model.py:228-230) as ONE autograd node (DS2_COMPOSITE_NODE=1; the round-1..3 graph): the
    backward of the fp32 / small-shape path keeps every layer's weight-gradient GEMMs on the second stream under the next layer's
    sweep and joins the streams once, at the very end -- at the price that DDP's reducer sees all recurrent gradients together when
    the node returns.  Same per-layer functions as _RnnLayerFn."""

    @staticmethod
    def forward(ctx, X, mod, lens_dev, N, Tp, dtype, training, n_layers, *rest):
        L = n_layers
        hs0, cs0, params = rest[:L], rest[L:2 * L], rest[2 * L:]
        ctx.rows = mod._frame_rows                   # the row list of the batch in flight (set by _logits)
        saved, meta, outs_h, outs_c, pos = [], [], [], [], 0
        for li in range(L):
            n = _layer_param_count(mod.rnns[li])
            out, hn, cn, sv4, mt = _rnn_layer_forward(mod, li, X, lens_dev, N, Tp, dtype, training, hs0[li], cs0[li], params[pos:pos + n],
                                                      rows=ctx.rows)
            pos += n
            saved += sv4
            meta.append(mt)
            outs_h.append(hn)
            if cn is not None:
                outs_c.append(cn)
            X = out
        ctx.mod, ctx.dims, ctx.meta = mod, (N, Tp, dtype, L), meta
        ctx.had_state = any(h is not None for h in hs0)
        ctx.n_params = len(params)
        ctx.save_for_backward(lens_dev, *saved, *params)
        ctx.mark_non_differentiable(*outs_h, *outs_c)
        return (X, *outs_h, *outs_c)

    @staticmethod
    def backward(ctx, dout, *unused):
        N, Tp, dtype, L = ctx.dims
        if ctx.had_state:
            raise Ds2HipError("backward through a forward that was given initial hidden states (hs) is not supported")
        mod = ctx.mod
        st = ctx.saved_tensors
        lens_dev, saved, params = st[0], st[1:1 + 4 * L], st[1 + 4 * L:]
        main = torch.cuda.current_stream()
        side = mod._wgrad_stream(dout.device)
        offs, pos = [], 0
        for li in range(L):
            offs.append(pos)
            pos += _layer_param_count(mod.rnns[li])
        grads = [None] * len(params)
        dout = dout.contiguous().to(dtype)
        wgrad_done = None
        for li in reversed(range(L)):
            n = _layer_param_count(mod.rnns[li])
            dout, lg, wgrad_done = _rnn_layer_backward(mod, li, ctx.meta[li], list(saved[4 * li:4 * li + 4]), params[offs[li]:offs[li] + n],
                                                       dout, lens_dev, N, Tp, dtype, main, side, wgrad_done, rows=ctx.rows)
            grads[offs[li]:offs[li] + n] = lg
        if wgrad_done is not None:
            main.wait_event(wgrad_done)
        if getattr(ctx, "defer_join", False):
            ctx.pending = (main, side, grads, mod)     # the composite node joins after the conv backward
        else:
            _join_side(main, side, grads, mod)
        return (dout, None, None, None, None, None, None, None, *([None] * (2 * L)), *grads)


def _unpad_layer_grads(grads, po, D, G, Ht, H, It, first):
    """Models whose hidden size is not a multiple of the tile: cut the padded units out of the layer's parameter gradients."""
    if H == Ht:
        return
    for d in range(D):
        dWih, dWhh, dBih, dBhh = grads[po + 4 * d:po + 4 * d + 4]
        dWih = _unpad_gate_rows(dWih, G, Ht, H)
        grads[po + 4 * d] = dWih if first else dWih[:, :It].contiguous()
        grads[po + 4 * d + 1] = _unpad_gate_rows(dWhh, G, Ht, H)[:, :Ht].contiguous()
        grads[po + 4 * d + 2] = _unpad_gate_rows(dBih, G, Ht, H)
        grads[po + 4 * d + 3] = _unpad_gate_rows(dBhh, G, Ht, H)


def _join_side(main, side, grads, mod=None):
    """Every parameter gradient produced on the second stream is complete (and, with the opt-in early hand-off, averaged
    over the ranks) before autograd / DDP sees it."""
    main.wait_stream(side)
    sync = getattr(mod, "_grad_sync", None)
    if sync is not None:
        sync.wait_early()
    for g in grads:
        if g is not None:
            g.record_stream(main)


class _SubCtx:
    """Stands in for an autograd ctx when a stage runs as a plain function inside the composite node below."""

    def save_for_backward(self, *t):
        self.saved_tensors = t

    def mark_non_differentiable(self, *a):
        pass


class _FrontFn(torch.autograd.Function):
    """Conv front-end + RNN stack as ONE autograd node (model.py:217-230).  Same arithmetic as running _ConvStackFn and
    _RnnStackFn back to back; being one node lets backward keep the layer-0 weight-gradient GEMMs on the second stream
    while the conv backward runs on the caller's stream, and join the streams once, at the very end."""

    @staticmethod
    def forward(ctx, x, mod, lens_dev, N, Tp, dtype, training, n_layers, *rest):
        conv_params, rnn_rest = rest[:8], rest[8:]
        c1, c2 = _SubCtx(), _SubCtx()
        X0 = _ConvStackFn.forward(c1, x, *conv_params, mod, lens_dev, Tp, dtype, training)
        if mod._prep_done is not None:           # RNN weight re-layouts were prepared on the second stream meanwhile
            torch.cuda.current_stream().wait_event(mod._prep_done)
            mod._prep_done = None
        outs = _RnnStackFn.forward(c2, X0, mod, lens_dev, N, Tp, dtype, training, n_layers, *rnn_rest)
        ctx.c1, ctx.c2, ctx.L = c1, c2, n_layers
        ctx.mark_non_differentiable(*outs[1:])
        ctx.set_materialize_grads(False)           # no zero tensors for the (non-differentiable) state outputs in backward
        return outs

    @staticmethod
    def backward(ctx, dout, *unused):
        c1, c2, L = ctx.c1, ctx.c2, ctx.L
        if c1 is None:
            # the saved activations (hundreds of MB) are dropped after the first backward, like autograd frees its buffers
            raise RuntimeError("DeepSpeech (gfx950): trying to backward through the conv + RNN stack a second time -- its saved "
                               "activations were freed by the first backward; run the forward again (retain_graph is not supported)")
        c2.defer_join = True
        r = _RnnStackFn.backward(c2, dout)
        rnn_grads = r[8 + 2 * L:]
        cg = _ConvStackFn.backward(c1, r[0])
        _join_side(*c2.pending)
        ctx.c1 = ctx.c2 = None
        return (None,) * 8 + tuple(cg[1:9]) + (None,) * (2 * L) + tuple(rnn_grads)


class _LookaheadFn(torch.autograd.Function):
    """Lookahead + Hardtanh (model.py:105-135, 189-193), uni-directional models."""

    @staticmethod
    def forward(ctx, X, w, N, Tp):
        Ht, ctxlen = w.shape[0], w.shape[2]
        H = X.shape[1]                                        # internal width (hidden_size rounded up to the tile)
        wf = w.detach().reshape(Ht, ctxlen)
        if H != Ht:
            wf = torch.cat([wf, wf.new_zeros((H - Ht, ctxlen))], 0)
        wf = wf.contiguous()
        y, pre = ops.lookahead_fwd(X, wf, Tp, N, H)
        ctx.dims = (N, Tp, H, ctxlen, Ht)
        ctx.save_for_backward(X, wf, pre)
        return y

    @staticmethod
    def backward(ctx, dy):
        X, wf, pre = ctx.saved_tensors
        N, Tp, H, ctxlen, Ht = ctx.dims
        dx, dw = ops.lookahead_bwd(X, wf, pre, dy.contiguous().to(X.dtype), Tp, N, H)
        return dx, dw[:Ht].reshape(Ht, 1, ctxlen).contiguous(), None, None


class _HeadFn(torch.autograd.Function):
    """fc = SequenceWise(BatchNorm1d(H) -> Linear(H, C, bias=False)) (model.py:195-201).  Output: fp32 logits [T'*N][32]
    (C = 29 valid columns, the rest zero) -- the padded leading dimension is what the CTC kernel and the GEMMs want."""

    @staticmethod
    def forward(ctx, X, bn_w, bn_b, wfc, mod, N, Tp, dtype, training):
        bn = mod.fc[0].module[0]
        Ht, Cc = wfc.shape[1], wfc.shape[0]
        H, Cp = mod._Hp, mod._Cp                     # internal (tile-aligned) width and class count
        R = Tp * N
        c = mod._cache
        Xh = torch.empty_like(X)
        sv = _bn_seq_fwd(bn, bn_w, bn_b, X, Xh, R, Ht, H, training)
        Wp = c.get(("wfc", dtype), [wfc], lambda: _pad_cols(torch.cat(
            [wfc.detach(), torch.zeros(Cp - Cc, Ht, device=wfc.device)], 0), H).to(dtype).contiguous())      # [Cp][H]
        logits = ops.gemm_nt(Xh, Wp, out_dtype=torch.float32)                                                # [R][Cp]
        ctx.mod, ctx.dims, ctx.sv = mod, (N, Tp, dtype, H, Cc, Ht, Cp), sv
        ctx.save_for_backward(X, Xh, wfc)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        X, Xh, wfc = ctx.saved_tensors
        N, Tp, dtype, H, Cc, Ht, Cp = ctx.dims
        R = Tp * N
        c = ctx.mod._cache
        dl = dlogits.contiguous().to(dtype)                                                                  # [R][Cp]
        WpT = c.get(("wfcT", dtype), [wfc], lambda: _pad_cols(torch.cat(
            [wfc.detach(), torch.zeros(Cp - Cc, Ht, device=wfc.device)], 0), H).t().to(dtype).contiguous())  # [H][Cp]
        dXh = ops.gemm_nt(dl, WpT)                                                                           # [R][H]
        dl_T, Xh_T = ops.transpose(dl), ops.transpose(Xh)
        dW = ops.gemm_nt(dl_T, Xh_T, out_dtype=torch.float32, splitk=max(1, min(32, R // 2048)))             # [Cp][H]
        dX = torch.empty_like(dXh)
        dg, db = _bn_seq_bwd(dXh, X, dX, ctx.sv, R, Ht, H)
        return dX, dg, db, dW[:Cc, :Ht].contiguous(), None, None, None, None, None


class _CtcFn(torch.autograd.Function):
    """log_softmax + CTCLoss(blank, 'sum', zero_infinity=True) (model.py:246,203,248); gradient computed in the same
    launch as the loss and scaled by the upstream gradient in backward."""

    @staticmethod
    def forward(ctx, logits, targets, out_lens_dev, target_sizes, N, Tp, Cc, blank):
        dev = logits.device
        tsz = target_sizes.to(torch.int64).cpu()
        offs = torch.zeros(N, dtype=torch.int64)
        if N > 1:
            offs[1:] = torch.cumsum(tsz, 0)[:-1]
        max_tl = int(tsz.max().item()) if N > 0 else 0
        # one host staging buffer, one host-to-device copy: [target offsets | target sizes | targets] as int32
        tg = targets.cpu() if targets.is_cuda else targets
        parts = [offs.to(torch.int32).numpy(), tsz.to(torch.int32).numpy(), tg.reshape(-1).to(torch.int32).numpy()]
        meta = _CTC_STAGING.stage(dev, parts)       # pinned ring slot: see _frame_rows
        loss, nll, dl = ops.ctc_loss_grad(logits, meta[2 * N:], meta[:N], out_lens_dev, meta[N:2 * N], Tp, N, Cc, blank, max_tl)
        ctx.save_for_backward(dl)
        ctx.consumed = False
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        if ctx.consumed:
            raise RuntimeError("DeepSpeech (gfx950): backward through the CTC loss a second time -- its gradient buffer was scaled in "
                               "place by the first backward; run the forward again (retain_graph is not supported)")
        ctx.consumed = True
        # the gradient was computed with the loss for a unit upstream gradient; scale it in place (backward runs once per forward)
        return ops.scale_by_(dl, g.detach().float()), None, None, None, None, None, None, None


class InferenceBatchSoftmax(nn.Module):  # model.py:72-77: identity in train mode, softmax over the classes in eval mode
    def forward(self, input_):
        if self.training:
            return input_
        if not input_.is_cuda:
            raise Ds2HipError("InferenceBatchSoftmax (gfx950) needs a HIP tensor; there is no CPU fallback")
        x = input_.float().contiguous()
        return ops.softmax_rows(x.view(-1, x.shape[-1]), x.shape[-1]).view(x.shape)


class CTCLossHip(nn.Module):
    """``self.criterion`` of the reference (model.py:203: ``CTCLoss(blank, reduction='sum', zero_infinity=True)``) on the HIP
    CTC kernels.  Called like the reference calls it (model.py:248): ``criterion(out (T',N,C), targets, output_sizes,
    target_sizes)``; `out` may be log-probabilities or raw logits (the kernel applies log_softmax, which is the identity on
    log-probabilities).  Differentiable with respect to `out`."""

    def __init__(self, blank=0):
        super().__init__()
        self.blank, self.reduction, self.zero_infinity = blank, "sum", True

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        Tp, N, Cc = log_probs.shape
        if not log_probs.is_cuda:
            raise Ds2HipError("CTCLossHip needs a HIP tensor; there is no CPU fallback")
        x = torch.zeros((Tp * N, (Cc + 31) // 32 * 32), dtype=torch.float32, device=log_probs.device)
        x[:, :Cc] = log_probs.reshape(Tp * N, Cc).float()
        loss = _CtcFn.apply(x, targets, input_lengths.to(log_probs.device, torch.int32), target_lengths, N, Tp, Cc, self.blank)
        return loss


def _reference_metrics(decoder):
    """The reference's own torchmetrics classes (validation.py:48-132; they sync across DDP ranks) when its package and their
    dependencies are importable, else the single-process stand-ins of .decoder."""
    try:
        from deepspeech_pytorch.validation import CharErrorRate, WordErrorRate
    except Exception:
        from .decoder import CharErrorRate, WordErrorRate
    return (WordErrorRate(decoder=decoder, target_decoder=decoder), CharErrorRate(decoder=decoder, target_decoder=decoder))


# ==================================================================================================================
# the model
# ==================================================================================================================
class DeepSpeech(_Base):
    def __init__(self, labels: List, model_cfg, precision, optim_cfg, spect_cfg):
        super().__init__()
        self.save_hyperparameters()
        self.model_cfg = model_cfg
        self.precision = precision
        self.optim_cfg = optim_cfg
        self.spect_cfg = spect_cfg
        self.bidirectional = _cfg_type_name(model_cfg) == "BiDirectionalConfig"   # model.py:152
        self.labels = labels
        num_classes = len(self.labels)
        kind = rnn_kind(model_cfg.rnn_type)
        H, L = int(model_cfg.hidden_size), int(model_cfg.hidden_layers)
        if num_classes > 8192:      # the reference takes any labels file; ds2_ctc_loss_grad keeps per-class sums of a frame in LDS
            raise ValueError("the CTC gradient kernel holds one row of per-class sums per wave in LDS: at most 8192 classes, got %d" % num_classes)
        # Internal, tile-aligned sizes (never visible in the state_dict): the recurrent kernels tile the hidden units by 16 and
        # the head / CTC kernels the classes by 32.  Extra hidden units carry zero weights and biases (they stay exactly 0 in
        # GRU, LSTM and tanh cells), extra classes zero weight rows that nothing reads.
        self._Hp = _padded_hidden(H, kind, precision, self.bidirectional)
        self._Cp = (num_classes + 31) // 32 * 32

        self.conv = MaskConv(nn.Sequential(
            nn.Conv2d(1, 32, kernel_size=(41, 11), stride=(2, 2), padding=(20, 5)),
            nn.BatchNorm2d(32),
            nn.Hardtanh(0, 20, inplace=True),
            nn.Conv2d(32, 32, kernel_size=(21, 11), stride=(2, 1), padding=(10, 5)),
            nn.BatchNorm2d(32),
            nn.Hardtanh(0, 20, inplace=True)
        ))
        rnn_input_size = int(math.floor((self.spect_cfg.sample_rate * self.spect_cfg.window_size) / 2) + 1)
        rnn_input_size = int(math.floor(rnn_input_size + 2 * 20 - 41) / 2 + 1)
        rnn_input_size = int(math.floor(rnn_input_size + 2 * 10 - 21) / 2 + 1)
        rnn_input_size *= 32
        # frequency geometry (model.py:166-169): any SpectConfig the reference accepts.  161 bins (16 kHz / 20 ms) run the tuned
        # matrix-pipe kernels of the bf16 path; other geometries the general conv kernels (ds2hip.h, conv front-end)
        self._F0 = int(math.floor((self.spect_cfg.sample_rate * self.spect_cfg.window_size) / 2) + 1)
        self._F1, self._F2 = ops.conv_rows(self._F0)
        self._rnn_in, self._rnn_ld = 32 * self._F2, _round64(32 * self._F2)
        if rnn_input_size != self._rnn_in or self._F2 < 1:
            raise ValueError("unsupported spectrogram geometry: %d frequency bins" % self._F0)
        self.rnns = nn.Sequential(
            BatchRNN(rnn_input_size, H, kind, bidirectional=self.bidirectional, batch_norm=False),
            *(BatchRNN(H, H, kind, bidirectional=self.bidirectional) for _ in range(L - 1))
        )
        self.lookahead = nn.Sequential(
            Lookahead(H, context=model_cfg.lookahead_context),
            nn.Hardtanh(0, 20, inplace=True)
        ) if not self.bidirectional else None
        fully_connected = nn.Sequential(nn.BatchNorm1d(H), nn.Linear(H, num_classes, bias=False))
        self.fc = nn.Sequential(SequenceWise(fully_connected))
        self.blank_index = self.labels.index('_')      # model.py:203
        # model.py:202-212: the attributes validation_step / external callers use, with the reference's names
        from .decoder import GreedyDecoder
        self.inference_softmax = InferenceBatchSoftmax()
        self.criterion = CTCLossHip(blank=self.blank_index)
        self.evaluation_decoder = GreedyDecoder(self.labels, blank_index=self.blank_index)   # arg-max + collapse on the device
        self.wer, self.cer = _reference_metrics(self.evaluation_decoder)
        self._cache = _WeightCache()
        self._flat_bias = {}
        self._side_streams = {}
        self._prep_done = None
        self._frame_rows = None          # row list of the batch in flight (_frame_rows)
        self._weights_dirty = False
        self._kind = kind

    def _prep_rnn_weights(self, dtype, need_backward):
        """Builds (or revalidates) every kernel-layout copy of the RNN weights -- bf16 casts, direction stacking, transposes
        for the dX / BPTT kernels -- in one place, so that forward can run it on the second stream under the conv stack
        (after an optimizer step every copy is stale: ~130 MB of fp32 reads per step)."""
        c = self._cache
        if self._Hp != self.rnns[0].hidden_size:
            return          # padded hidden size (rare): the stages build their zero-padded operands themselves, through the cache
        for li, layer in enumerate(self.rnns):
            p, D, first = layer.rnn, layer.num_directions, li == 0
            sufs = [""] + (["_reverse"] if D == 2 else [])
            wih = [getattr(p, "weight_ih_l0" + s_) for s_ in sufs]
            whh = [getattr(p, "weight_hh_l0" + s_) for s_ in sufs]
            bih = [getattr(p, "bias_ih_l0" + s_) for s_ in sufs]
            bhh = [getattr(p, "bias_hh_l0" + s_) for s_ in sufs]

            if dtype == torch.bfloat16:
                # one fused cast(+transpose) kernel per parameter matrix: 4 B read, 2 (+2) B written per element
                keys = [("wih", li, dtype), ("whh", li, dtype)]
                if need_backward:
                    keys += [("wihT", li, dtype), ("whhT", li, dtype)]
                if not all(c.valid(k, wih if k[0].startswith("wih") else whh) for k in keys):
                    GH, H = whh[0].shape
                    I = wih[0].shape[1]
                    Io = self._rnn_ld if first else I
                    perm = (32, self._F2) if first else None
                    dev = wih[0].device
                    Wih = ops.empty_padded(D * GH, Io, dtype, dev)
                    Whh = torch.empty((D, GH, H), dtype=dtype, device=dev)
                    WihT = torch.empty((Io, D * GH), dtype=dtype, device=dev) if need_backward else None
                    WhhT = torch.empty((D, H, GH), dtype=dtype, device=dev) if need_backward else None
                    with torch.no_grad():
                        for d in range(D):
                            ops.cast_transpose_bf16(wih[d].detach(), Wih[d * GH:], Wih.stride(0),
                                                    WihT[:, d * GH:] if need_backward else None, D * GH, perm=perm, cout=Io)
                            ops.cast_transpose_bf16(whh[d].detach(), Whh[d], H, WhhT[d] if need_backward else None, GH)
                    c.put(("wih", li, dtype), wih, Wih)
                    c.put(("whh", li, dtype), whh, Whh)
                    if need_backward:
                        c.put(("wihT", li, dtype), wih, WihT)
                        c.put(("whhT", li, dtype), whh, WhhT)

            def cat_ih(wih=wih, first=first):
                w = torch.cat([q.detach() for q in wih], 0)
                return _perm_cols_to_internal(w) if first else w
            def padded_ih():
                w = cat_ih()
                out = ops.empty_padded(w.shape[0], w.shape[1], dtype, w.device)
                out.copy_(w)
                return out
            c.get(("wih", li, dtype), wih, padded_ih)
            c.get(("whh", li, dtype), whh, lambda whh=whh: torch.stack([q.detach() for q in whh], 0).to(dtype).contiguous())
            self._bias_views(li)
            if need_backward:
                c.get(("whhT", li, dtype), list(whh),
                      lambda whh=whh: torch.stack([q.detach().t() for q in whh], 0).to(dtype).contiguous())
                c.get(("wihT", li, dtype), list(wih), lambda: cat_ih().t().to(dtype).contiguous())

    def _bias_views(self, li):
        """(bias_ih of all directions [D*G*H], bias_hh [D][G*H]) as VIEWS of the parameters: the directions' bias vectors of a
        layer share one flat storage (set up here, like nn.RNNBase.flatten_parameters does for cuDNN/MIOpen; re-done if a
        .to() / load replaced the tensors), so no per-step concatenation is needed and in-place optimizer updates are seen."""
        layer = self.rnns[li]
        p, D = layer.rnn, layer.num_directions
        sufs = [""] + (["_reverse"] if D == 2 else [])
        out = []
        for name in ("bias_ih_l0", "bias_hh_l0"):
            ps = [getattr(p, name + s_) for s_ in sufs]
            n = ps[0].numel()
            key = (li, name)
            flat = self._flat_bias.get(key)
            ok = flat is not None and all(q.data_ptr() == flat.data_ptr() + 4 * n * i and q.is_contiguous() for i, q in enumerate(ps))
            if not ok:
                with torch.no_grad():
                    flat = torch.cat([q.detach().reshape(-1) for q in ps]).contiguous()
                    for i, q in enumerate(ps):
                        q.data = flat[i * n:(i + 1) * n]
                self._flat_bias[key] = flat
            out.append(flat)
        return out[0], out[1].view(D, -1)

    def _prep_small_weights(self, dtype):
        """conv / head weights in kernel layout: one launch when the parameters changed (every training step)."""
        c = self._cache
        sm, fcw = self.conv.seq_module, self.fc[0].module[1].weight
        w1, w2 = sm[0].weight, sm[3].weight
        keys = (("w1k", [w1]), (("w2t", dtype), [w2]), (("w2d", dtype), [w2]), (("wfc", dtype), [fcw]), (("wfcT", dtype), [fcw]))
        if all(c.valid(k, ps) for k, ps in keys[:3]) and (c.valid(*keys[3]) or fcw.shape[0] > 32 or self._Hp != fcw.shape[1]):
            return
        plain_head = fcw.shape[0] <= 32 and self._Hp == fcw.shape[1]
        with torch.no_grad():
            vals = ops.small_weight_layouts(w1.detach().contiguous(), w2.detach().contiguous(),
                                            fcw.detach().contiguous() if plain_head else fcw.detach()[:min(32, fcw.shape[0])].contiguous(), dtype)
        for (k, ps), v in zip(keys, vals):
            if plain_head or k[0] not in ("wfc", "wfcT"):       # > 32 classes / padded hidden size: _HeadFn builds the head's operands
                c.put(k, ps, v)

    def _wgrad_stream(self, device):
        """Second HIP stream of this module on `device` (weight-gradient work of the RNN stack's backward)."""
        key = device.index if device.index is not None else torch.cuda.current_device()
        st = self._side_streams.get(key)
        if st is None:
            st = self._side_streams[key] = torch.cuda.Stream(device=device)
        return st

    # ---- precision policy -------------------------------------------------------------------------------------
    def compute_dtype(self):
        """bf16 storage/MFMA operands under autocast or when constructed with a 16-bit precision; fp32 otherwise
        (the 1e-3 parity mode).  Parameters, statistics, gate math, CTC and parameter gradients are always fp32."""
        if torch.is_autocast_enabled():
            return torch.bfloat16
        if str(self.precision) in ("16", "bf16", "16-mixed", "bf16-mixed"):
            return torch.bfloat16
        return torch.float32

    # ---- forward ------------------------------------------------------------------------------------------------
    def _logits(self, x, lengths, hs=None):
        if not x.is_cuda:
            raise Ds2HipError("DeepSpeech (gfx950) needs its input on a HIP device; there is no CPU fallback")
        dtype = self.compute_dtype()
        if self.training:
            self._cache.epoch += 1                         # parameters may have been updated in place since the last forward
            self._weights_dirty = True
        elif self._weights_dirty:
            self._cache.epoch += 1
            self._weights_dirty = False
        lengths = lengths.cpu().int()                      # model.py:215
        output_lengths = self.get_seq_lens(lengths)        # model.py:216
        N, T = x.shape[0], x.shape[3]
        Tp = (T + 2 * 5 - 10 - 1) // 2 + 1
        dev = x.device
        lens_dev, self._frame_rows = _frame_rows(output_lengths, Tp, N, dev)
        training = self.training
        sm = self.conv.seq_module
        with torch.autocast("cuda", enabled=False):
            # weight re-layouts for the RNN stack: second stream, hidden under the conv front-end
            self._prep_small_weights(dtype)
            if self._Hp == self.rnns[0].hidden_size:
                for li in range(len(self.rnns)):     # one-time flattening of the bias storages: on THIS stream (its torch.cat and the
                    self._bias_views(li)             # release of the old storages must not straddle two streams' allocator pools)
            main, side = torch.cuda.current_stream(), self._wgrad_stream(dev)
            side.wait_stream(main)
            ready = getattr(self, "_layouts_ready", None)
            if ready is not None:                # bf16 layouts written by the fused optimizer on ITS stream
                side.wait_event(ready)
                self._layouts_ready = None
            with torch.cuda.stream(side):
                self._prep_rnn_weights(dtype, training and torch.is_grad_enabled())
            prep_done = torch.cuda.Event()
            prep_done.record(side)
            self._prep_done = prep_done          # awaited inside the composite node, between the conv and the RNN stack
            L = len(self.rnns)
            if hs is None:
                hs = [None] * L
            h0s, c0s, params = [], [], []
            for i, layer in enumerate(self.rnns):
                h0 = c0 = None
                if hs[i] is not None:
                    if self._kind == "lstm":
                        h0, c0 = hs[i][0].float().contiguous(), hs[i][1].float().contiguous()
                    else:
                        h0 = hs[i].float().contiguous()
                h0s.append(h0)
                c0s.append(c0)
                if layer.batch_norm is not None:
                    bn = layer.batch_norm.module
                    params += [bn.weight, bn.bias]
                p = layer.rnn
                for suf in [""] + (["_reverse"] if self.bidirectional else []):
                    params += [getattr(p, "weight_ih_l0" + suf), getattr(p, "weight_hh_l0" + suf),
                               getattr(p, "bias_ih_l0" + suf), getattr(p, "bias_hh_l0" + suf)]
            # bf16 performance mode: every layer's backward runs on the caller's stream (BPTT sweep, one launch with the weight
            # gradients + dX), so one node per layer costs nothing and lets DDP's reducer overlap.  fp32 / small / odd shapes keep
            # their weight-gradient GEMMs on the second stream under the next layer's sweep: there the composite node (one join at
            # the end) is 1.1 ms faster on config 2 (12.4 vs 13.6 ms, profiles/r04a), and it is used unless a data-parallel
            # wrapper is attached (DS2_COMPOSITE_NODE=0 forces one node per layer everywhere).
            Hh = self._Hp
            GH = ops.GATES[self._kind] * Hh
            Dd = 2 if self.bidirectional else 1
            fast_bwd = (ops.wgrad_tn_ok(dtype, Tp * N, Dd * GH, Hh, lda=Dd * GH, ldb=ops.pad_ld(Hh, dtype)) and
                        ops.wgrad_tn_ok(dtype, Tp * N, GH, Hh, lda=Dd * GH, ldb=Hh))
            composite = COMPOSITE_NODE if COMPOSITE_NODE is not None else (not fast_bwd and not _data_parallel_active())
            if any(h is not None for h in h0s) and torch.is_grad_enabled():
                composite = False                  # backward through a given `hs` lives in the per-layer nodes (_RnnLayerFn)
            if composite:
                res = _FrontFn.apply(x, self, lens_dev, N, Tp, dtype, training, L, sm[0].weight, sm[0].bias, sm[1].weight, sm[1].bias,
                                     sm[3].weight, sm[3].bias, sm[4].weight, sm[4].bias, *h0s, *c0s, *params)
                X = res[0]
                hn_l = list(res[1:1 + L])
                cn_l = list(res[1 + L:1 + 2 * L]) if self._kind == "lstm" else []
            else:
                # default graph: conv stack -> one node per BatchRNN layer (a layer's gradients reach DDP's reducer as soon as the
                # layer's backward returns, under the sweeps of the layers below)
                X = _ConvStackFn.apply(x, sm[0].weight, sm[0].bias, sm[1].weight, sm[1].bias, sm[3].weight, sm[3].bias, sm[4].weight,
                                       sm[4].bias, self, lens_dev, Tp, dtype, training)
                if self._prep_done is not None:           # RNN weight re-layouts were prepared on the second stream meanwhile
                    torch.cuda.current_stream().wait_event(self._prep_done)
                    self._prep_done = None
                hn_l, cn_l, pos = [], [], 0
                for i, layer in enumerate(self.rnns):
                    n = _layer_param_count(layer)
                    outs = _RnnLayerFn.apply(X, self, lens_dev, N, Tp, dtype, training, i, h0s[i], c0s[i], *params[pos:pos + n])
                    pos += n
                    X = outs[0]
                    hn_l.append(outs[1])
                    if len(outs) > 2:
                        cn_l.append(outs[2])
            if self._kind == "lstm":
                new_hs = [(hn_l[i], cn_l[i]) for i in range(L)]
            else:
                new_hs = list(hn_l)
            if not self.bidirectional:
                X = _LookaheadFn.apply(X, self.lookahead[0].conv.weight, N, Tp)
            fcm = self.fc[0].module
            logits = _HeadFn.apply(X, fcm[0].weight, fcm[0].bias, fcm[1].weight, self, N, Tp, dtype, training)
        return logits, output_lengths, new_hs, lens_dev, N, Tp

    def forward(self, x, lengths, hs=None):
        logits, output_lengths, new_hs, _, N, Tp = self._logits(x, lengths, hs)
        Cc = len(self.labels)
        if self.training:
            out = logits.view(Tp, N, self._Cp)[:, :, :Cc].transpose(0, 1)  # (N, T', C) view, model.py:236
        else:
            out = ops.softmax_rows(logits, Cc).view(Tp, N, Cc).transpose(0, 1)   # model.py:238, 72-77
        return out, output_lengths, new_hs

    def training_step(self, batch, batch_idx):
        inputs, targets, input_percentages, target_sizes = batch
        if inputs.is_cuda:
            ops.poll_persistent_error(inputs.device)      # a sweep that timed out in an earlier step raises here (no host sync)
        input_sizes = input_percentages.mul_(int(inputs.size(3))).int()       # model.py:243
        logits, output_sizes, _, lens_dev, N, Tp = self._logits(inputs, input_sizes)
        return _CtcFn.apply(logits, targets, lens_dev, target_sizes, N, Tp, len(self.labels), self.blank_index)

    def validation_step(self, batch, batch_idx):
        # model.py:251-271, statement for statement.  The reference's `autocast(enabled=self.precision == 16)` is fp16
        # autocast; the HIP path has bf16 and fp32 kernels, so a 16-bit precision selects bf16 storage (compute_dtype) and the
        # autocast region is entered only to keep the dtype of any torch op a subclass adds consistent with the reference.
        inputs, targets, input_percentages, target_sizes = batch
        input_sizes = input_percentages.mul_(int(inputs.size(3))).int()
        inputs = inputs.to(self.device)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=str(self.precision) in ("16", "bf16", "16-mixed", "bf16-mixed")):
            out, output_sizes, hs = self(inputs, input_sizes)
        decoded_output, _ = self.evaluation_decoder.decode(out, output_sizes)
        self.wer(preds=out, preds_sizes=output_sizes, targets=targets, target_sizes=target_sizes)
        self.cer(preds=out, preds_sizes=output_sizes, targets=targets, target_sizes=target_sizes)
        self.log('wer', self.wer.compute(), prog_bar=True, on_epoch=True)
        self.log('cer', self.cer.compute(), prog_bar=True, on_epoch=True)

    def attach_evaluation(self, decoder, wer=None, cer=None):
        """Swap in other decoder / metric objects (e.g. the reference's beam decoder); any object with the reference's
        ``decode(probs, sizes)`` / ``__call__(preds, preds_sizes, targets, target_sizes)`` + ``compute()`` interface."""
        self.evaluation_decoder = decoder
        if wer is not None:
            self.wer = wer
        if cer is not None:
            self.cer = cer

    def configure_optimizers(self):  # model.py:273-297
        # Same optimizers and hyper-parameters as the reference.  On a HIP device they are the subclasses of .optim whose
        # step() runs the multi-tensor HIP kernels (same arithmetic, same state_dict; the recurrent weights' bf16 layouts for
        # the next step are written in the same pass); a model still on the CPU gets the stock torch classes.
        from .optim import FusedAdamW, FusedSGD
        name = _cfg_type_name(self.optim_cfg)
        on_gpu = all(p.is_cuda for p in self.parameters())
        if name == "SGDConfig":
            kw = dict(params=self.parameters(), lr=self.optim_cfg.learning_rate, momentum=self.optim_cfg.momentum, nesterov=True,
                      weight_decay=self.optim_cfg.weight_decay)
            optimizer = FusedSGD(model=self, **kw) if on_gpu else torch.optim.SGD(**kw)
        elif name == "AdamConfig":
            kw = dict(params=self.parameters(), lr=self.optim_cfg.learning_rate, betas=tuple(self.optim_cfg.betas),
                      eps=self.optim_cfg.eps, weight_decay=self.optim_cfg.weight_decay)
            optimizer = FusedAdamW(model=self, **kw) if on_gpu else torch.optim.AdamW(**kw)
        else:
            raise ValueError("Optimizer has not been specified correctly.")
        scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer=optimizer, gamma=self.optim_cfg.learning_anneal)
        return [optimizer], [scheduler]

    def get_seq_lens(self, input_length):  # model.py:299-310
        seq_len = input_length
        for m in self.conv.modules():
            if type(m) == nn.modules.conv.Conv2d:
                seq_len = ((seq_len + 2 * m.padding[1] - m.dilation[1] * (m.kernel_size[1] - 1) - 1) // m.stride[1] + 1)
        return seq_len.int()
