"""The optimizer step on the HIP kernels of csrc/ds2_optim.hip.

``FusedAdamW`` / ``FusedSGD`` ARE ``torch.optim.AdamW`` / ``torch.optim.SGD`` (subclasses: same constructor, param_groups,
state and ``state_dict`` keys -- 'step', 'exp_avg', 'exp_avg_sq' / 'momentum_buffer' -- so Lightning's checkpointing, the
ExponentialLR scheduler of reference model.py:292-296 and resume work unchanged) with ``step()`` replaced by

  1. optionally the global-norm clip (``clip_grad_norm``: what Lightning's gradient_clip_val: 400 does through
     ``torch.nn.utils.clip_grad_norm_``, configs/an4.yaml:12) computed on the device, no host synchronisation;
  2. one multi-tensor launch for the small parameters (conv, BatchNorm, biases, head);
  3. one launch for the recurrent weight matrices (round 6; the column-permuted ``rnns.0.weight_ih`` separately) that updates them
     AND writes the bf16 operand copy and transpose the next forward / backward of the drop-in model needs (they land in the
     model's weight cache, stamped valid for the next step).

Arithmetic = torch's single-tensor AdamW / SGD-Nesterov in fp32 (tests/test_gpu_optim.py: <= a few ulp per step against
torch.optim on the CPU).  There is no CPU path: CPU parameters raise."""
import ctypes as C
import math

import torch

from . import ops
from ._lib import Ds2HipError, call, query


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else 0 for t in tensors])


def _long_array(vals):
    return (C.c_long * len(vals))(*vals)


def _int_array(vals):
    return (C.c_int * len(vals))(*vals)


def _float_array(vals):
    return (C.c_float * len(vals))(*vals)


class _FusedStep:
    """Shared machinery: parameter classification (recurrent weight matrices of an attached drop-in model vs the rest),
    clipping, launches."""

    def _init_fused(self, model=None, clip_grad_norm=None):
        self.clip_grad_norm = clip_grad_norm
        self._model = model
        self._clip_out = {}
        self._matrix_plan = None

    # ---- which parameters are recurrent weight matrices of the attached model, and where their bf16 layouts go
    def _plan(self):
        if self._matrix_plan is not None:
            return self._matrix_plan
        plan = {}
        m = self._model
        if m is not None:
            for li, layer in enumerate(m.rnns):
                D = layer.num_directions
                for d, suf in enumerate([""] + (["_reverse"] if D == 2 else [])):
                    wih, whh = getattr(layer.rnn, "weight_ih_l0" + suf), getattr(layer.rnn, "weight_hh_l0" + suf)
                    first = li == 0
                    plan[id(wih)] = ("ih", li, d, D, (32, m._F2) if first else None, m._rnn_ld if first else wih.shape[1])
                    plan[id(whh)] = ("hh", li, d, D, None, whh.shape[1])
        self._matrix_plan = plan
        return plan

    def _layout_buffers(self, li, D, GH, H, Io, dev):
        """bf16 operand tensors of layer li in the model's cache layout (allocated once, rewritten in place every step)."""
        key = ("opt_layouts", li)
        buf = getattr(self, "_layouts", None)
        if buf is None:
            buf = self._layouts = {}
        if key not in buf:
            bt = torch.bfloat16
            buf[key] = dict(Wih=ops.empty_padded(D * GH, Io, bt, dev), Whh=torch.empty((D, GH, H), dtype=bt, device=dev),
                            WihT=torch.empty((Io, D * GH), dtype=bt, device=dev), WhhT=torch.empty((D, H, GH), dtype=bt, device=dev))
        return buf[key]

    def _clip(self, grads, dev):
        if not self.clip_grad_norm:
            return None
        out = self._clip_out.get(dev)
        if out is None:
            out = self._clip_out[dev] = torch.empty(2, dtype=torch.float32, device=dev)
        n = _long_array([g.numel() for g in grads])
        ws = torch.empty(query("ds2_clip_ws_floats", len(grads), n), dtype=torch.float32, device=dev)
        call("ds2_clip_coef", len(grads), _ptr_array(grads), n, float(self.clip_grad_norm), ops.P(out), ops.P(ws), ops.S())
        return out

    @property
    def last_grad_norm(self):
        """Total gradient norm of the last step (device tensor; reading it synchronises) -- what clip_grad_norm_ returns."""
        return {d: o[0] for d, o in self._clip_out.items()}

    def _run(self, mode, entries, hp_of_group, first_of_group):
        """entries: list of (group index, param, grad, state1, state2 or None)."""
        if not entries:
            return
        dev = entries[0][1].device
        for _, p, g, s1, s2 in entries:
            if not p.is_cuda:
                raise Ds2HipError("FusedAdamW / FusedSGD need the parameters on a HIP device; there is no CPU path")
            if p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous():
                raise Ds2HipError("FusedAdamW / FusedSGD update contiguous float32 parameters with float32 gradients")
        grads = [e[2] if e[2].is_contiguous() else e[2].contiguous() for e in entries]
        clip = self._clip(grads, dev)
        plan = self._plan()
        m = self._model
        emit = m is not None and m.training and m.compute_dtype() == torch.bfloat16 and m._Hp == m.rnns[0].hidden_size
        small = {}
        mats = {}
        touched = {}
        for (gi, p, _, s1, s2), g in zip(entries, grads):
            info = plan.get(id(p))
            if info is None or p.shape[0] % 16 != 0:
                small.setdefault(gi, []).append((p, g, s1, s2))
                continue
            kind, li, d, D, perm, Io = info
            GH, Cc = p.shape
            dst = dstT = None
            ldd = lddT = 0
            if emit:
                layer = m.rnns[li]
                io_layer = plan[id(layer.rnn.weight_ih_l0)][5]          # padded input width of this layer's W_ih layouts
                L = self._layout_buffers(li, D, GH, layer.hidden_size, io_layer, dev)
                if kind == "ih":
                    dst, ldd, dstT, lddT = L["Wih"][d * GH:], L["Wih"].stride(0), L["WihT"][:, d * GH:], D * GH
                else:
                    dst, ldd, dstT, lddT = L["Whh"][d], Cc, L["WhhT"][d], GH
                touched[li] = (L, D)
            pc, pf = perm if perm is not None else (0, 0)
            mats.setdefault(gi, []).append((p, g, s1, s2, GH, Cc, pc, pf, Io if kind == "ih" else Cc, dst, ldd, dstT, lddT))
        for gi, lst in mats.items():      # one launch per group for the matrices without a column permutation (ds2_opt_matrices)
            col = list(zip(*lst))
            call("ds2_opt_matrices", mode, len(lst), _ptr_array(col[0]), _ptr_array(col[1]), _ptr_array(col[2]), _ptr_array(col[3]),
                 _int_array(col[4]), _int_array(col[5]), _int_array(col[6]), _int_array(col[7]), _int_array(col[8]), _ptr_array(col[9]),
                 _long_array(col[10]), _ptr_array(col[11]), _long_array(col[12]), _float_array(hp_of_group[gi]), first_of_group[gi],
                 ops.P(clip), ops.S())
        for gi, lst in small.items():
            n = _long_array([p.numel() for p, _, _, _ in lst])
            call("ds2_opt_multi", mode, len(lst), _ptr_array([x[0] for x in lst]), _ptr_array([x[1] for x in lst]),
                 _ptr_array([x[2] for x in lst]), _ptr_array([x[3] for x in lst]), n, _float_array(hp_of_group[gi]),
                 first_of_group[gi], ops.P(clip), ops.S())
        if emit and touched:
            # the layouts written above are those of the UPDATED weights: valid for the next training forward (which advances the
            # model's cache epoch by one before it looks)
            c = m._cache
            ev = torch.cuda.Event()
            ev.record()
            for li, (L, D) in touched.items():
                layer = m.rnns[li]
                sufs = [""] + (["_reverse"] if D == 2 else [])
                wih = [getattr(layer.rnn, "weight_ih_l0" + s_) for s_ in sufs]
                whh = [getattr(layer.rnn, "weight_hh_l0" + s_) for s_ in sufs]
                if all(id(w) in plan and w.grad is not None for w in wih + whh):
                    bt = torch.bfloat16
                    c.put(("wih", li, bt), wih, L["Wih"], epoch=c.epoch + 1)
                    c.put(("whh", li, bt), whh, L["Whh"], epoch=c.epoch + 1)
                    c.put(("wihT", li, bt), wih, L["WihT"], epoch=c.epoch + 1)
                    c.put(("whhT", li, bt), whh, L["WhhT"], epoch=c.epoch + 1)
            m._layouts_ready = ev          # the forward's second stream orders itself after the optimizer's stream


class FusedAdamW(torch.optim.AdamW, _FusedStep):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, model=None, clip_grad_norm=None):
        torch.optim.AdamW.__init__(self, params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._init_fused(model, clip_grad_norm)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries, hp, first = [], {}, {}
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                raise Ds2HipError("FusedAdamW implements the reference's configuration (model.py:283-289): no amsgrad / maximize")
            beta1, beta2 = group["betas"]
            lr, wd, eps = group["lr"], group["weight_decay"], group["eps"]
            step = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st["step"].is_cuda:              # a checkpoint of torch's fused AdamW keeps 'step' on the device: normalise once,
                    st["step"] = st["step"].cpu()   # or every step would synchronise once per parameter
                st["step"] += 1
                s_ = float(st["step"])
                if step is not None and s_ != step:
                    raise Ds2HipError("FusedAdamW: parameters of one group must share the step count")
                step = s_
                entries.append((gi, p, p.grad, st["exp_avg"], st["exp_avg_sq"]))
            if step is not None:
                bc1 = 1 - beta1 ** step                      # python doubles, exactly as torch/optim/adam.py
                bc2 = 1 - beta2 ** step
                hp[gi] = [1 - lr * wd, 1 - beta1, beta2, 1 - beta2, bc2 ** 0.5, eps, -(lr / bc1)]
                first[gi] = 0
        self._run(0, entries, hp, first)
        return loss


class FusedSGD(torch.optim.SGD, _FusedStep):
    def __init__(self, params, lr=1e-3, momentum=0.9, weight_decay=0.0, nesterov=True, model=None, clip_grad_norm=None):
        torch.optim.SGD.__init__(self, params, lr=lr, momentum=momentum, nesterov=nesterov, weight_decay=weight_decay)
        self._init_fused(model, clip_grad_norm)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries, hp, first = [], {}, {}
        for gi, group in enumerate(self.param_groups):
            if not group["nesterov"] or group["dampening"] != 0 or group.get("maximize") or group["momentum"] <= 0:
                raise Ds2HipError("FusedSGD implements the reference's configuration (model.py:275-281): Nesterov momentum")
            fresh = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                is_first = "momentum_buffer" not in st or st["momentum_buffer"] is None
                if is_first:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if fresh is not None and fresh != is_first:
                    raise Ds2HipError("FusedSGD: parameters of one group must share the step count")
                fresh = is_first
                entries.append((gi, p, p.grad, st["momentum_buffer"], None))
            if fresh is not None:
                hp[gi] = [group["weight_decay"], group["momentum"], 0.0, 0.0, 1.0, 0.0, -group["lr"]]
                first[gi] = 1 if fresh else 0
        self._run(1, entries, hp, first)
        return loss
