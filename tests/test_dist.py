"""N > 1 path on CPU: two processes, gloo backend (runs with -m "not gpu").  Covers the data-parallel plumbing that
bench.py and a Lightning-style DDP launch use around the drop-in class (deepspeech/pytorch_amd/dist.py): the StepModule
wrapper under torch DDP (gradient mean over ranks, rank-local BatchNorm buffers), the barrier-bracketed timing, the
max-over-ranks / sum-over-ranks aggregation of the metric, and the per-rank sharding of the synthetic minibatches.
The HIP kernels themselves cannot run here, so a small CPU stand-in with the drop-in class's training_step signature is
stepped instead; on the GPU the same wrapper carries the real class (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _StandIn(torch.nn.Module):
    """training_step((inputs, targets, input_percentages, target_sizes), batch_idx) -> scalar loss (sum over the batch)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.bn = torch.nn.BatchNorm1d(8)
        self.fc = torch.nn.Linear(8, 4)

    def training_step(self, batch, batch_idx):
        x, targets, pct, tsz = batch
        return (self.fc(self.bn(x)) ** 2).sum() * float(pct.sum())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, use_ddp=False):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("DS2_USE_DDP", None)
    if use_ddp:
        os.environ.pop("DS2_OVERLAP_ALLREDUCE", None)          # the default: torch DDP
    else:
        os.environ["DS2_OVERLAP_ALLREDUCE"] = "1"                # opt-in: the in-place per-layer all-reduce
    from deepspeech.pytorch_amd import dist as dsdist
    r, w, lr = dsdist.init_from_env("gloo")
    assert (r, w, lr) == (rank, world, rank)
    dev = torch.device("cpu")
    model = _StandIn()
    step_mod = dsdist.wrap_data_parallel(dsdist.StepModule(model), dev, world)
    # default = torch DDP; DS2_OVERLAP_ALLREDUCE=1 = the overlapped wrapper (a model without early hand-offs is reduced by its
    # end-of-backward bucket)
    assert isinstance(step_mod, torch.nn.parallel.DistributedDataParallel if use_ddp else dsdist.OverlappedDataParallel)
    g = torch.Generator().manual_seed(100 + rank)          # every rank has its own minibatch
    x = torch.randn(6, 8, generator=g)
    pct = torch.ones(6)

    def step():
        model.zero_grad()
        loss = step_mod(x, torch.zeros(1), pct.clone(), torch.ones(6, dtype=torch.int32))
        loss.backward()
        return loss

    secs, last = dsdist.timed_steps(step, 3, dev, world)
    grad = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # reference: mean over ranks of the gradients of the per-rank losses, computed locally without DDP
    refs = []
    for rr in range(world):
        m2 = _StandIn()
        gg = torch.Generator().manual_seed(100 + rr)
        x2 = torch.randn(6, 8, generator=gg)
        l2 = m2.training_step((x2, None, torch.ones(6), None), 0)
        l2.backward()
        refs.append(torch.cat([p.grad.reshape(-1) for p in m2.parameters()]))
    ref = torch.stack(refs).mean(0)
    tmax, total = dsdist.aggregate(float(rank + 1), 10.0 * (rank + 1), dev, world)
    # the start-up budget a persistent sweep of THIS rank would be launched with (round 6): the process group's time-out, not 300 ms
    from deepspeech.pytorch_amd import ops
    out[rank] = dict(grad_err=float((grad - ref).abs().max()), grad_norm=float(ref.abs().max()), tmax=tmax, total=total,
                     bn_mean=model.bn.running_mean.clone().numpy(), secs=secs, loss=float(last.detach()),
                     ranks=ops.data_parallel_ranks(), startup_ms=ops.persist_startup_ms())
    dsdist.shutdown(world)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("use_ddp", [False, True])
def test_two_rank_gloo_data_parallel_step(use_ddp):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, use_ddp), nprocs=world, join=True)
    assert sorted(out.keys()) == [0, 1]
    for r in range(world):
        o = out[r]
        assert o["grad_err"] <= 1e-5 * max(1.0, o["grad_norm"]), o       # DDP: every rank holds the MEAN gradient
        assert o["tmax"] == 2.0 and o["total"] == 30.0                     # max over ranks / sum over ranks
        assert o["secs"] > 0
        # a sweep behind an RCCL collective that waits for a late peer must WAIT (ops.persist_startup_ms): >= 30 s under 2 ranks
        assert o["ranks"] == 2 and o["startup_ms"] >= 30_000, o
    # BatchNorm statistics are rank-local (sync_batchnorm=False in the reference): different minibatches -> different buffers
    assert not np.allclose(out[0]["bn_mean"], out[1]["bn_mean"])


class _EarlyNode(torch.autograd.Function):
    """Stand-in for the drop-in class's composite backward node: two 'layers' whose weight gradients are final before the
    node returns; each is handed to model._grad_sync the moment it exists (as views of one stacked storage, like the
    direction-stacked dW_ih), and the node waits for the early all-reduces before it returns them to autograd."""

    @staticmethod
    def forward(ctx, x, w1, w2, owner):
        h = x @ w1
        ctx.save_for_backward(x, w1, w2, h)
        ctx.owner = owner
        return h @ w2

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, h = ctx.saved_tensors
        sync = getattr(ctx.owner, "_grad_sync", None)
        stacked = torch.cat([h.t() @ dy, torch.zeros(1, w2.shape[1])], 0)      # gradient = a view of a larger storage
        g2 = stacked[:w2.shape[0]]
        if sync is not None:
            sync.layer_ready([stacked], [w2])
        dh = dy @ w2.t()
        g1 = x.t() @ dh
        if sync is not None:
            sync.layer_ready([g1], [w1])
            sync.wait_early()
        return None, g1, g2, None


class _EarlyStandIn(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(1)
        self.w1 = torch.nn.Parameter(torch.randn(8, 8))
        self.w2 = torch.nn.Parameter(torch.randn(8, 4))
        self.bn = torch.nn.BatchNorm1d(4)            # parameters that are NOT handed over early: the flat bucket of finish()

    def training_step(self, batch, batch_idx):
        x, targets, pct, tsz = batch
        return (self.bn(_EarlyNode.apply(x, self.w1, self.w2, self)) ** 2).sum() * float(pct.sum())


def _worker_overlap(rank, world, port, out):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from deepspeech.pytorch_amd import dist as dsdist
    dsdist.init_from_env("gloo")
    dev = torch.device("cpu")
    model = _EarlyStandIn()
    with torch.no_grad():
        model.w1.add_(float(rank))                   # replicas differ before wrapping: the wrapper must broadcast rank 0's
    step_mod = dsdist.wrap_data_parallel(dsdist.StepModule(model), dev, world, overlap=True)
    assert isinstance(step_mod, dsdist.OverlappedDataParallel)
    g = torch.Generator().manual_seed(200 + rank)
    x = torch.randn(6, 8, generator=g)
    errs = []
    for it in range(2):                              # two steps: the per-step state of the sync object is reset
        model.zero_grad()
        loss = step_mod(x, torch.zeros(1), torch.ones(6), torch.ones(6, dtype=torch.int32))
        loss.backward()                              # finish() runs as an engine callback: no explicit call needed
        assert step_mod.sync._done
        step_mod.finish_backward()                   # idempotent
        grad = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        refs = []
        for rr in range(world):
            m2 = _EarlyStandIn()
            gg = torch.Generator().manual_seed(200 + rr)
            l2 = m2.training_step((torch.randn(6, 8, generator=gg), None, torch.ones(6), None), 0)
            l2.backward()
            refs.append(torch.cat([p.grad.reshape(-1) for p in m2.parameters()]))
        ref = torch.stack(refs).mean(0)
        errs.append(float((grad - ref).abs().max() / ref.abs().max()))
    out[rank] = dict(errs=errs, early=len(step_mod.sync._early), w1=model.w1.detach().clone().numpy())
    dsdist.shutdown(world)


@pytest.mark.timeout(300)
def test_two_rank_gloo_overlapped_gradient_sync():
    """The opt-in early hand-off (dist.OverlappedGradSync): gradients handed over inside backward and the flat rest bucket
    both end up as the mean over the ranks; replicas are broadcast from rank 0 at wrap time."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_overlap, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        assert max(out[r]["errs"]) <= 1e-5, out[r]
        assert out[r]["early"] == 2                                          # w1 and w2 went early, the BN pair through finish()
    assert np.array_equal(out[0]["w1"], out[1]["w1"])


class _LayerNode(torch.autograd.Function):
    """Stand-in for model._RnnLayerFn: ONE autograd node per layer whose backward returns the layer's weight gradient together with
    dX -- the node structure of the drop-in class since round 4 (conv stack -> one node per BatchRNN layer -> head)."""

    @staticmethod
    def forward(ctx, x, w, log, li):
        ctx.save_for_backward(x, w)
        ctx.log, ctx.li = log, li
        return torch.tanh(x @ w)

    @staticmethod
    def backward(ctx, dy):
        import time
        x, w = ctx.saved_tensors
        ctx.log.append(("layer_bwd_start", ctx.li, time.perf_counter()))
        y = torch.tanh(x @ w)
        dz = dy * (1 - y * y)
        time.sleep(0.05)                              # the layer's BPTT sweep: the time an overlapped all-reduce can hide under
        return dz @ w.t(), x.t() @ dz, None, None


class _CompositeNode(torch.autograd.Function):
    """The round-1..3 structure: all layers inside ONE node -- every weight gradient leaves it together, at the end."""

    @staticmethod
    def forward(ctx, x, log, *ws):
        acts = [x]
        for w in ws:
            acts.append(torch.tanh(acts[-1] @ w))
        ctx.save_for_backward(*acts, *ws)
        ctx.log, ctx.n = log, len(ws)
        return acts[-1]

    @staticmethod
    def backward(ctx, dy):
        import time
        st = ctx.saved_tensors
        acts, ws = st[:ctx.n + 1], st[ctx.n + 1:]
        grads = [None] * ctx.n
        for li in reversed(range(ctx.n)):
            ctx.log.append(("layer_bwd_start", li, time.perf_counter()))
            dz = dy * (1 - acts[li + 1] * acts[li + 1])
            grads[li] = acts[li].t() @ dz
            dy = dz @ ws[li].t()
            time.sleep(0.05)
        return (dy, None, *grads)


class _LayeredStandIn(torch.nn.Module):
    def __init__(self, per_layer_nodes, width=600, layers=3):
        super().__init__()
        torch.manual_seed(2)
        self.ws = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(width, width) / width ** 0.5) for _ in range(layers)])   # 1.4 MB each
        self.per_layer_nodes, self.log = per_layer_nodes, []

    def training_step(self, batch, batch_idx):
        x = batch[0]
        if self.per_layer_nodes:
            for li, w in enumerate(self.ws):
                x = _LayerNode.apply(x, w, self.log, li)
        else:
            x = _CompositeNode.apply(x, self.log, *self.ws)
        return (x ** 2).sum()


def _worker_progressive(rank, world, port, out, per_layer_nodes):
    import time
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("DS2_OVERLAP_ALLREDUCE", None)
    os.environ.pop("DS2_USE_DDP", None)
    from deepspeech.pytorch_amd import dist as dsdist
    dsdist.init_from_env("gloo")
    dev = torch.device("cpu")
    model = _LayeredStandIn(per_layer_nodes)
    ddp = dsdist.wrap_data_parallel(dsdist.StepModule(model), dev, world, bucket_cap_mb=1)      # one layer (1.4 MB) per bucket
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    log = model.log

    def hook(state, bucket):                                  # what DDP's default hook does, plus a time stamp when the reducer launches it
        log.append(("allreduce_start", bucket.index(), time.perf_counter()))
        fut = dist.all_reduce(bucket.buffer(), async_op=True).get_future()
        return fut.then(lambda f: f.value()[0] / world)
    ddp.register_comm_hook(None, hook)
    g = torch.Generator().manual_seed(300 + rank)
    x = torch.randn(16, 600, generator=g)
    for it in range(2):                                       # the second step runs on the rebuilt (gradient-ready-order) buckets
        del log[:]
        model.zero_grad()
        loss = ddp(x, torch.zeros(1), torch.ones(16), torch.ones(16, dtype=torch.int32))
        loss.backward()
    starts = sorted(t for k, _, t in log if k == "allreduce_start")
    last_layer_bwd = [t for k, li, t in log if k == "layer_bwd_start" and li == 0][0]      # layer 0 runs its backward LAST
    out[rank] = dict(n_allreduce=len(starts), before_last_layer=sum(1 for t in starts if t < last_layer_bwd),
                     grad=float(sum(p.grad.abs().sum() for p in model.parameters())))
    dsdist.shutdown(world)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("per_layer_nodes", [True, False])
def test_ddp_overlaps_with_per_layer_nodes(per_layer_nodes):
    """Plain ``DistributedDataParallel`` -- what Lightning's ``strategy: ddp`` builds around the module (reference training.py:42-47,
    configs/librispeech.yaml:14) -- launches a layer's bucket all-reduce BEFORE the backward of the layers below has run when every
    layer is its own autograd node (the drop-in class's graph since round 4), and only after the whole backward when the stack is one
    composite node (rounds 1-3: the exposed ~3 ms of VERDICT round 3, missing #2)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_progressive, args=(world, _free_port(), out, per_layer_nodes), nprocs=world, join=True)
    for r in range(world):
        assert out[r]["n_allreduce"] == 3, out[r]
        if per_layer_nodes:
            assert out[r]["before_last_layer"] >= 2, out[r]          # layers 2 and 1 are on the wire before layer 0's backward starts
        else:
            assert out[r]["before_last_layer"] == 0, out[r]
    assert abs(out[0]["grad"] - out[1]["grad"]) <= 1e-4 * abs(out[0]["grad"])      # both ranks hold the same (mean) gradients


def test_default_wrapper_is_torch_ddp(monkeypatch):
    from deepspeech.pytorch_amd import dist as dsdist
    monkeypatch.delenv("DS2_OVERLAP_ALLREDUCE", raising=False)
    monkeypatch.delenv("DS2_USE_DDP", raising=False)
    assert dsdist.overlap_allreduce() is False
    monkeypatch.setenv("DS2_OVERLAP_ALLREDUCE", "1")
    assert dsdist.overlap_allreduce() is True
    monkeypatch.setenv("DS2_USE_DDP", "1")
    assert dsdist.overlap_allreduce() is False


def test_single_rank_is_identity():
    from deepspeech.pytorch_amd import dist as dsdist
    m = dsdist.StepModule(_StandIn())
    assert dsdist.wrap_data_parallel(m, torch.device("cpu"), 1) is m
    assert dsdist.aggregate(1.5, 7.0, torch.device("cpu"), 1) == (1.5, 7.0)


def test_bench_shards_distinct_minibatches_per_rank():
    import bench
    l0, _ = bench_lengths(bench, 0)
    l1, _ = bench_lengths(bench, 1)
    assert len(l0) == len(l1) == 32 and l0[0] == l1[0] == 1501          # weak scaling: 32 clips per rank, same Tmax
    assert not np.array_equal(l0, l1)
    assert all(np.diff(l0) <= 0)                                       # sorted by length, descending (collate contract)


def bench_lengths(bench, rank):
    from deepspeech.pytorch_amd import synth
    kind, H, L, bi, N, tmin, tmax, dtype = bench.CONFIGS["cfg3"]
    lengths = synth.synth_lengths(N, tmin, tmax, seed=3 * 1000 + rank)
    return lengths, None
