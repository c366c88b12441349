"""Data-parallel plumbing of the training step: one process per GPU, ``torch.distributed`` over RCCL (backend "nccl" on
ROCm); the only exchange of the path is the gradient mean over the ranks.

The reference has no explicit collective: Lightning's ``strategy: ddp`` (configs/an4.yaml:13, librispeech.yaml:14) wraps the
LightningModule in ``DistributedDataParallel`` with ``sync_batchnorm=False`` (lightning_config.py:53), i.e. BatchNorm statistics
stay rank-local.  The drop-in class works unchanged under that wrapper, and since round 4 it OVERLAPS there: every BatchRNN layer is
its own autograd node (model._RnnLayerFn), so a layer's weight gradients reach the reducer's hooks the moment the layer's backward
returns (its BPTT sweep, then one launch with the weight gradients + dX, all on the caller's stream) and the bucket's all-reduce
runs on RCCL's stream under the sweeps of the layers below -- exactly how DDP overlaps the reference's nn.GRU layers.  Buckets are
sized for xGMI (point-to-point links: a ring collective is bound by ONE link): 64 MB, about one GRU-1024 layer (~50 MB fp32) per
bucket -- few, large collectives.  tests/test_dist.py proves on two gloo ranks, with a stand-in of the same node structure, that
the buckets' all-reduces are launched before backward ends.

``wrap_data_parallel`` returns torch DDP by default (round 4; the advisor's round-3 finding: the custom wrapper below had never run
on real multi-GPU hardware).  ``OverlappedDataParallel`` / ``OverlappedGradSync`` stay available behind DS2_OVERLAP_ALLREDUCE=1: one
in-place all-reduce per layer straight on the gradient storages (no bucket copies), started inside the layer's backward.
DS2_FORCE_DDP=1 wraps (and initialises a 1-rank process group) even for world == 1, to exercise the wrapper + RCCL on a single-GPU
box.

Everything here is device-agnostic so that the N > 1 path is covered on CPU with the gloo backend (tests/test_dist.py).
"""
import os
import time

import torch
import torch.distributed as dist


class StepModule(torch.nn.Module):
    """forward(batch...) = model.training_step(batch): what Lightning's DDP strategy does with a LightningModule, so that
    DDP's forward/backward hooks bracket the whole training step of the drop-in class."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, inputs, targets, input_percentages, target_sizes):
        return self.model.training_step((inputs, targets, input_percentages, target_sizes), 0)


def force_ddp():
    return os.environ.get("DS2_FORCE_DDP", "0") not in ("", "0")


def init_from_env(backend):
    """(rank, world, local_rank) from the launcher's environment (torch.distributed.run); initialises the process group
    when WORLD_SIZE > 1.  Rendezvous defaults to 127.0.0.1 (single node)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_ddp()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


class OverlappedGradSync:
    """Gradient mean over the ranks for a model whose backward hands finished gradients over EARLY (opt-in, see
    wrap_data_parallel).  The drop-in class computes each recurrent layer's weight gradients on a second stream while the
    BPTT sweeps of the layers below are still running; with this object attached (``model._grad_sync``) the composite
    backward node calls ``layer_ready`` right after enqueueing those GEMMs, in that stream's context, so RCCL's stream picks
    the all-reduce up as soon as the gradients exist and the exchange runs under the remaining sweeps -- what DDP's
    reducer cannot do for a node that returns all its gradients at once.  Everything not handed over early (conv, BN,
    head: ~1 MB) goes through one flat bucket in ``finish`` (queued as an autograd-engine callback, like DDP's finalize).
    Device-agnostic: tests/test_dist.py drives it over gloo."""

    def __init__(self, world, group=None):
        self.world, self.group = world, group
        self._params = []
        self._handles, self._early, self._deferred = [], set(), []
        self._queued = self._done = False

    def attach(self, params):
        self._params = list(params)

    def begin_step(self):
        if (self._handles or self._deferred) and not self._done:
            # gradients were handed over early (all-reduces in flight, some withheld from autograd) but the end-of-backward reduction
            # never ran: the arm hook did not fire (the loss that was backpropagated did not depend on the wrapper's outputs?) and
            # finish_backward() was not called -- the previous step's gradients would be silently missing / unreduced
            raise RuntimeError("OverlappedGradSync: the previous step's gradient reduction never finished -- call finish_backward() "
                               "after loss.backward(), or use torch DDP (DS2_OVERLAP_ALLREDUCE=0)")
        self._handles, self._early, self._deferred = [], set(), []
        self._queued = self._done = False

    def _start(self, t):
        if dist.get_backend(self.group) == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        t.mul_(1.0 / self.world)                       # gloo has no AVG
        return dist.all_reduce(t, group=self.group, async_op=True)

    def arm(self, *_):
        """Queues `finish` behind the running backward pass (the autograd engine's end-of-backward callback, what DDP's reducer
        uses for its own finalize).  Only valid while a backward pass is running: OverlappedDataParallel calls it from a hook
        on the step's loss, i.e. when backward starts -- models that never hand a layer over early are reduced too."""
        if not self._queued:
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
            self._queued = True

    def layer_ready(self, tensors, params):
        """tensors: gradient storages that are final (distinct storages; the gradients returned to autograd may be views
        of them); params: the parameters they belong to.  Call inside backward, on the stream that produces them."""
        self.arm()
        for t in tensors:
            self._handles.append(self._start(t))
        self._early.update(p.data_ptr() for p in params)

    def defer_early(self, params, grads):
        """Per-layer graph (model._RnnLayerFn): the layer's backward returns while its all-reduce is still in flight, so the
        gradients handed over early must not travel through autograd (AccumulateGrad may clone or add to them on the caller's
        stream).  They are withheld here -- the node returns None for them -- and become ``p.grad`` in ``finish``, after the
        all-reduces.  Returns `grads` with those entries replaced by None."""
        out = list(grads)
        for i, (p, g) in enumerate(zip(params, grads)):
            if g is not None and p.data_ptr() in self._early:
                self._deferred.append((p, g))
                out[i] = None
        return out

    def wait_early(self):
        """Orders the caller's stream (CPU: the caller) after every all-reduce started so far; call before the early
        gradients are returned to autograd."""
        for h in self._handles:
            h.wait()
        self._handles = []

    def finish(self):
        if self._done:
            return
        self.wait_early()
        with torch.no_grad():
            for p, g in self._deferred:                   # withheld from autograd by defer_early: averaged by now
                p.grad = g if p.grad is None else p.grad.add_(g)
            self._deferred = []
            # everything else in ONE flat bucket; a parameter that got no gradient on THIS rank contributes zeros (every rank
            # reduces the same set: the trainable parameters that were not handed over early)
            rest = [p for p in self._params if p.requires_grad and p.data_ptr() not in self._early]
            if rest:
                flat = torch.cat([(p.grad.reshape(-1).float() if p.grad is not None else p.new_zeros(p.numel(), dtype=torch.float32))
                                  for p in rest])
                self._start(flat).wait()
                off = 0
                for p in rest:
                    n = p.numel()
                    if p.grad is None:
                        p.grad = flat[off:off + n].view_as(p).to(p.dtype).clone()
                    else:
                        p.grad.copy_(flat[off:off + n].view_as(p.grad))
                    off += n
        self._done = True


def _grad_tensors(out):
    """Every tensor that requires grad in a (nested) tuple / list / dict of outputs."""
    if torch.is_tensor(out):
        return [out] if out.requires_grad else []
    if isinstance(out, dict):
        out = list(out.values())
    found = []
    if isinstance(out, (tuple, list)):
        for o in out:
            found += _grad_tensors(o)
    return found


class OverlappedDataParallel(torch.nn.Module):
    """The step module with an OverlappedGradSync attached to its model (opt-in alternative to torch DDP).  Replicas
    start from rank 0's parameters and buffers, as under DDP; BatchNorm statistics stay rank-local afterwards."""

    def __init__(self, step_module, world):
        super().__init__()
        self.module = step_module
        self.sync = OverlappedGradSync(world)
        self.sync.attach(step_module.parameters())
        with torch.no_grad():
            for t in list(step_module.parameters()) + list(step_module.buffers()):
                dist.broadcast(t, 0)
        step_module.model._grad_sync = self.sync

    def forward(self, *args):
        self.sync.begin_step()
        out = self.module(*args)
        if torch.is_grad_enabled():
            ts = _grad_tensors(out)
            if not ts:
                raise RuntimeError("OverlappedDataParallel: the wrapped module returned nothing that requires grad -- the end-of-backward "
                                   "gradient reduction cannot be armed (use torch DDP: DS2_OVERLAP_ALLREDUCE=0)")
            for t in ts:                           # EVERY such output: whichever of them the loss depends on arms the reduction when
                t.register_hook(self.sync.arm)     # backward reaches it (arm is idempotent)
        return out

    def finish_backward(self):
        """Explicit end of backward for loops that want it (idempotent: the engine callback normally ran already)."""
        self.sync.finish()


def overlap_allreduce():
    """Default for world > 1: torch DDP -- with one autograd node per BatchRNN layer its reducer overlaps the bucketed all-reduce
    with the BPTT sweeps of the layers below, as Lightning's ``strategy: ddp`` does in the reference's own loop.
    DS2_OVERLAP_ALLREDUCE=1 selects OverlappedDataParallel instead (one in-place all-reduce per layer, started inside the layer's
    backward; never validated on more than one GPU: opt-in).  DS2_USE_DDP=1 forces DDP whatever the other variable says."""
    if os.environ.get("DS2_USE_DDP", "0") not in ("", "0"):
        return False
    return os.environ.get("DS2_OVERLAP_ALLREDUCE", "0") not in ("", "0")


def wrap_data_parallel(step_module, device, world, bucket_cap_mb=64, overlap=None):
    """Data-parallel wrapper of the step module (identity for world == 1): torch DDP by default (broadcast_buffers=False: BatchNorm
    running statistics are per-rank like the reference's sync_batchnorm=False; gradient_as_bucket_view avoids one copy per
    parameter; 64 MB buckets ~ one recurrent layer each), OverlappedDataParallel with overlap=True / DS2_OVERLAP_ALLREDUCE=1."""
    if world == 1 and not (force_ddp() and dist.is_initialized()):
        return step_module
    try:      # the drop-in class then builds one autograd node per layer for every shape (the reducer sees gradients progressively)
        from . import model as _model
        _model.PER_LAYER_NODES_FOR_DDP[0] = True
        from . import ops as _ops
        _ops.FORCE_DATA_PARALLEL_BUDGET[0] = True      # sweeps wait for their CUs behind a collective (ops.persist_startup_ms)
    except Exception:  # noqa: BLE001  (stand-in modules in the CPU tests)
        pass
    if overlap_allreduce() if overlap is None else overlap:
        return OverlappedDataParallel(step_module, world)
    ids = [device.index] if device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(step_module, device_ids=ids, broadcast_buffers=False,
                                                     bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)


def _sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def timed_steps(step_fn, steps, device, world):
    """Runs `steps` calls of step_fn bracketed by barrier + device synchronisation on both sides; returns (seconds on this
    rank, last return value)."""
    _sync(device)
    if world > 1:
        dist.barrier()
    _sync(device)
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step_fn()
    _sync(device)
    if world > 1:
        dist.barrier()
    _sync(device)
    return time.perf_counter() - t0, last


def aggregate(seconds, units, device, world):
    """(max over ranks of the timed seconds, sum over ranks of the processed units)."""
    if world == 1:
        return float(seconds), float(units)
    t = torch.tensor([seconds, units], dtype=torch.float64, device=device)
    tmax, tsum = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    return float(tmax[0]), float(tsum[1])


def shutdown(world):
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
