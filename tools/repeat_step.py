#!/usr/bin/env python
"""Flake screen of a whole training step: N fresh models of the lstm_bi_1280 fixture (groups of 2, 2 and 1 clips in the round-4 general
sweeps), upstream gradient 1 / 65 536 alternating, device memory polluted with NaN between the steps; reports every step whose
gradients are not finite / differ from the first step's, and a raised sweep time-out.  (Found the lock-step violation of the
active-rows-only gather)

    gpurun -- 'python tools/repeat_step.py 60 > gpurun_out/repeat_step.txt'
"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fixtures import Fixture
from deepspeech.pytorch_amd import ops, _lib, model as M
import test_gpu_loop as T
fx = Fixture("lstm_bi_1280")
inputs, targets, pct, tsz = fx.batch()
batch = (torch.from_numpy(inputs).to("cuda"), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
if not os.environ.get("NOJUNK"):
    junk = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 20, 3 << 20, 1 << 24, 1 << 26)]
    del junk
ref = None
nbad = 0
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for it in range(n_it):
    m = T.build(fx, "bf16")
    try:
        l, g = T._grads_of_step(m, batch, 65536.0 if it % 2 else 1.0)
    except Exception as e:
        print("it", it, "raised", str(e)[:60], flush=True)
        nbad += 1
        ops._PERSIST_ERR.clear(); ops._ERR_MIRROR.clear()
        continue
    bad = [k for k, v in g.items() if not np.isfinite(v).all()]
    sc = 65536.0 if it % 2 else 1.0
    if ref is None:
        ref = {k: v / sc for k, v in g.items()}
    worst = max(np.abs(g[k] / sc - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-30) for k in g) if not bad else float("nan")
    if bad or worst > 2e-6:
        nbad += 1
        try:
            ops.check_persistent_kernels(); pk = "ok"
        except Exception as e:
            pk = "ERR " + str(e)[:80]
        print("it", it, "loss", l, "nan grads", len(bad), bad[:5], "worst", worst, pk, flush=True)
    if not os.environ.get("NOJUNK"):
        junk = torch.full((1 << 24,), float("nan"), device="cuda"); del junk
print("iterations", n_it, "bad", nbad, flush=True)
