#!/usr/bin/env python3
"""Full-size golden vectors: the REAL reference (``/root/reference/deepspeech_pytorch/model.py`` imported unmodified through
``ref_harness``) run on the CPU at a BASELINE.json configuration's FULL shape -- the same synthetic batch ``bench.py`` times
(seed cfg_id*1000) and parameters regenerated from a seed (``synth.synth_params``) -- dumping what fits in a small fixture:

    loss (fp32 run = the reference as shipped), output lengths, logits on every LOGIT_STRIDE-th frame, and for EVERY parameter
    gradient a strided sample (stride below; small tensors whole, as gradfull.<param>), its L2 norm and its sum;  with --autocast also the reference under
    ``torch.autocast(bfloat16)``: loss_ac and ``acnoise.<param>`` = relative L2 distance of its gradient from the fp32 run
    (the yard-stick of the bf16 bars in tests/test_gpu_model.py).

Build container only (minutes to tens of minutes of CPU).  Output: tests/golden/full/<config>.npz (+ the wall time of the fp32
step, which doubles as the reference-proper CPU baseline of SURVEY.md section 8d -> profiles/cpu_reference_<config>.json when
--baseline-json is given; note that this leg has no optimizer step).

    python tests/golden/make_fullsize_golden.py cfg3 --autocast
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import bench  # noqa: E402
import ref_harness  # noqa: E402
from deepspeech.pytorch_amd import synth  # noqa: E402

GRAD_STRIDE = 997       # prime: no resonance with the 1024 / 1280 / 3072 ... row lengths
FULL_BELOW = 16384      # gradients of at most this many elements (BatchNorm weights / biases, conv and RNN biases, the conv1 kernel)
                        # are stored whole as gradfull.<param>: a stride-997 sample of a 32-element tensor is ONE number
LOGIT_STRIDE = 16
PARAM_SEED = {"cfg2": 2202, "cfg3": 3303, "cfg5a": 5505, "cfg5b": 6606}
CFG_ID = {"cfg2": 2, "cfg3": 3, "cfg5a": 5, "cfg5b": 6}


def step(model, inputs, targets, pct, tsz, autocast):
    model.train()
    model.zero_grad()
    x = torch.from_numpy(inputs)
    sizes = torch.from_numpy(pct.copy()).mul_(int(inputs.shape[3])).int()           # model.py:243
    t0 = time.perf_counter()
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            logits, out_sizes, _ = model(x, sizes)                                  # model.py:244
            lp = logits.transpose(0, 1).log_softmax(-1)                             # model.py:245-246
            loss = model.criterion(lp, torch.from_numpy(targets), out_sizes, torch.from_numpy(tsz))   # model.py:248
    else:
        logits, out_sizes, _ = model(x, sizes)
        lp = logits.transpose(0, 1).log_softmax(-1)
        loss = model.criterion(lp, torch.from_numpy(targets), out_sizes, torch.from_numpy(tsz))
    loss.backward()
    return loss, logits, out_sizes, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--autocast", action="store_true")
    ap.add_argument("--threads", type=int, default=bench.usable_cores())
    a = ap.parse_args()
    kind, H, L, bi, N, tmin, tmax, dtype = bench.CONFIGS[a.config]
    torch.set_num_threads(a.threads)
    ns = ref_harness.load_reference()
    lengths = synth.synth_lengths(N, tmin, tmax, seed=CFG_ID[a.config] * 1000, linear=(a.config == "cfg2"))
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=CFG_ID[a.config] * 1000)
    model = ref_harness.build_reference_model(ns, kind, H, L, bi, 20)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    P = synth.synth_params(shapes, PARAM_SEED[a.config])
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    out = {}
    loss, logits, out_sizes, secs = step(model, inputs, targets, pct, tsz, False)
    print("fp32 step: %.1f s, loss %.4f" % (secs, loss.item()), flush=True)
    out["loss"] = np.float64(loss.item())
    out["output_lengths"] = out_sizes.numpy().astype(np.int32)
    out["logits_sub"] = logits.detach()[:, ::LOGIT_STRIDE].numpy().copy()               # (N, ceil(T'/16), C)
    g32 = {}
    for k, p in model.named_parameters():
        g = p.grad.detach().numpy().astype(np.float64)
        g32[k] = g
        out["gradsub." + k] = g.reshape(-1)[::GRAD_STRIDE].astype(np.float32)
        if g.size <= FULL_BELOW:
            out["gradfull." + k] = g.reshape(-1).astype(np.float32)
        out["gradl2." + k] = np.float64(np.sqrt((g ** 2).sum()))
        out["gradsum." + k] = np.float64(g.sum())
    secs_ac = None
    if a.autocast:
        del model
        mac = ref_harness.build_reference_model(ns, kind, H, L, bi, 20)
        mac.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
        loss_ac, _, _, secs_ac = step(mac, inputs, targets, pct, tsz, True)
        print("autocast(bf16) step: %.1f s, loss %.4f" % (secs_ac, loss_ac.item()), flush=True)
        out["loss_ac"] = np.float64(loss_ac.item())
        for k, p in mac.named_parameters():
            g = p.grad.detach().float().numpy().astype(np.float64)
            out["acnoise." + k] = np.float64(np.sqrt(((g - g32[k]) ** 2).sum()) / max(np.sqrt((g32[k] ** 2).sum()), 1e-30))
    meta = dict(config=a.config, rnn_type=kind, hidden_size=H, hidden_layers=L, bidirectional=bi, lookahead_context=20,
                lengths=[int(v) for v in lengths], data_seed=CFG_ID[a.config] * 1000, param_seed=PARAM_SEED[a.config],
                grad_stride=GRAD_STRIDE, logit_stride=LOGIT_STRIDE, shapes={k: list(v) for k, v in shapes.items()},
                torch_version=torch.__version__, threads=a.threads, fp32_forward_backward_seconds=secs,
                autocast_forward_backward_seconds=secs_ac)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    os.makedirs(os.path.join(HERE, "full"), exist_ok=True)
    path = os.path.join(HERE, "full", a.config + ".npz")
    np.savez_compressed(path, **out)
    print("%s: loss %.4f, %d gradients, file %.0f KB" % (path, loss.item(), len(g32), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
