#!/usr/bin/env python
"""Which ATen / runtime launches are left in a training step, and which line of this package issues them: one step of
`bench.py --config <cfg>` under torch.profiler with Python stacks; prints every aten:: op that launched a device kernel with its
count, device time and the innermost frames inside this repository.     python tools/trace_aten.py [cfg3]"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    from deepspeech.pytorch_amd import configs
    from deepspeech.pytorch_amd import dist as dsdist
    from deepspeech.pytorch_amd.model import DeepSpeech
    device = torch.device("cuda:0")
    kind, H, L, bi, N, tmin, tmax, dtype = bench.CONFIGS[cfg]
    torch.manual_seed(0)
    rt = getattr(configs.RNNType, kind)
    mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L) if bi else \
        configs.UniDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L, lookahead_context=20)
    model = DeepSpeech(configs.LABELS, mc, "bf16" if dtype == "bf16" else 32, configs.AdamConfig(), configs.SpectConfig()).to(device)
    model.train()
    lengths, batch = bench.build_batch(cfg, 0, device)
    opt = model.configure_optimizers()[0][0]
    opt.clip_grad_norm = 400.0
    step_mod = dsdist.wrap_data_parallel(dsdist.StepModule(model), device, 1)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = step_mod(batch[0], batch[1], batch[2].clone(), batch[3])
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith("aten::") and not ev.name.startswith("hipMem"):
            continue
        dev = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
        if not ev.kernels:
            continue
        frames = [f for f in (ev.stack or []) if "/repo/" in f or "deepspeech" in f or "bench" in f or "trace_aten" in f][:3]
        key = (ev.name, " <- ".join(f.replace(ROOT + "/", "") for f in frames) or "(autograd engine / no Python frame)")
        rows[key][0] += 1
        rows[key][1] += dev
    tot = 0.0
    print("| op | launches per step | device us | issued from |")
    print("|---|---|---|---|")
    for (name, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("| `%s` | %d | %.1f | %s |" % (name, n, us, where))
        tot += us
    print("\n%s: %.1f us of ATen / runtime device work per step" % (cfg, tot))


if __name__ == "__main__":
    main()
