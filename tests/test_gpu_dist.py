"""The data-parallel wrappers around the REAL drop-in class on the device: a 1-rank RCCL process group (backend "nccl" is
RCCL on ROCm) exercises torch DDP's reducer and the early hand-off of dist.OverlappedGradSync with the actual autograd
nodes (_FrontFn / _RnnStackFn on two HIP streams), against the unwrapped step.  N > 1 ranks are covered on CPU (gloo) in
tests/test_dist.py; the 8-GPU scaling run is the driver's (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch

from fixtures import Fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    if dist.is_initialized():
        dist.destroy_process_group()


def _step(fx, wrap, overlap):
    from test_gpu_model import build
    from deepspeech.pytorch_amd import dist as dsdist, ops
    m = build(fx, "bf16").train()
    inputs, targets, pct, tsz = fx.batch()
    sm = dsdist.StepModule(m)
    trace = []
    if wrap:
        if overlap:
            sm = dsdist.OverlappedDataParallel(sm, 1)
            orig = sm.sync.layer_ready
            sm.sync.layer_ready = lambda tensors, params: (trace.append([tuple(p.shape) for p in params]), orig(tensors, params))[1]
        else:
            sm = torch.nn.parallel.DistributedDataParallel(sm, device_ids=[0], broadcast_buffers=False, bucket_cap_mb=64,
                                                           gradient_as_bucket_view=True)
    losses = []
    for _ in range(2):                                # two steps: per-step state of the wrappers is reset correctly
        m.zero_grad(set_to_none=True)
        loss = sm(torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
        loss.backward()
        if overlap and wrap:
            sm.finish_backward()
        losses.append(float(loss.item()))
    torch.cuda.synchronize()
    ops.check_persistent_kernels()
    return losses, {k: p.grad.detach().float().cpu().numpy() for k, p in m.named_parameters()}, trace


@pytest.mark.parametrize("name", ["gru_bi_1024", "lstm_uni_la"])
def test_ddp_and_overlapped_sync_reproduce_the_unwrapped_gradients(one_rank_group, name):
    fx = Fixture(name)
    l0, g0, _ = _step(fx, False, False)
    l1, g1, _ = _step(fx, True, False)
    l2, g2, trace = _step(fx, True, True)
    assert l0 == l1 == l2
    for k in g0:
        sc = max(np.abs(g0[k]).max(), 1e-12)
        assert np.abs(g1[k] - g0[k]).max() <= 1e-5 * sc, ("ddp", k)
        assert np.abs(g2[k] - g0[k]).max() <= 1e-5 * sc, ("overlap", k)
    # the composite backward node hands the recurrent layers over top layer first, once per layer and step
    L = fx.cfg["hidden_layers"]
    assert len(trace) == 2 * L
    widths = [t[0][1] for t in trace[:L]]             # weight_ih input widths: H for the upper layers, 1312 for layer 0 (last)
    assert widths[-1] == 1312 and all(w == fx.cfg["hidden_size"] for w in widths[:-1])
