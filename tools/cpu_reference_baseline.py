#!/usr/bin/env python
"""Times the REFERENCE's own model.py (imported unmodified through tests/golden/ref_harness.py) on the host cores at a bench
configuration's FULL shape: zero_grad -> training_step -> backward -> clip_grad_norm_(400) -> AdamW.step, fp32 -- the
"reference" CPU baseline of SURVEY.md section 8(d).  Build container only (/root/reference is not on the GPU box); the result
is committed as profiles/cpu_reference_<config>.json and quoted by bench.py next to the on-box "port" sample.

    python tools/cpu_reference_baseline.py cfg3 [--steps 1 --warmup 0]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import ref_harness  # noqa: E402
from deepspeech.pytorch_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    a = ap.parse_args()
    kind, H, L, bi, N, tmin, tmax, dtype = bench.CONFIGS[a.config]
    cores = bench.usable_cores()
    torch.set_num_threads(cores)
    ns = ref_harness.load_reference()
    model = ref_harness.build_reference_model(ns, kind, H, L, bi, 20)
    model.train()
    cfg_id = {"cfg2": 2, "cfg3": 3, "cfg5a": 5, "cfg5b": 6}[a.config]
    lengths = synth.synth_lengths(N, tmin, tmax, seed=cfg_id * 1000, linear=(a.config == "cfg2"))
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=cfg_id * 1000)
    opt = model.configure_optimizers()[0][0]

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.training_step((torch.from_numpy(inputs), torch.from_numpy(targets), torch.from_numpy(pct.copy()),
                                    torch.from_numpy(tsz)), 0)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 400.0)
        opt.step()
        return float(loss.detach())

    for _ in range(a.warmup):
        step()
    times, loss = [], None
    for _ in range(a.steps):
        t0 = time.perf_counter()
        loss = step()
        times.append(time.perf_counter() - t0)
        print("step %.1f s loss %.3f" % (times[-1], loss), flush=True)
    best = min(times)
    secs = synth.audio_seconds(lengths)
    cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
    out = {"value": round(secs / best, 3), "unit": "audio-seconds/sec", "cores": cores, "kind": "reference",
           "sample": "/root/reference deepspeech_pytorch/model.py unmodified (Lightning/OmegaConf stubs), fp32, FULL %s shape (%s H=%d L=%d %s, "
                     "%d clips of %.2f-%.2f s), best of %d step(s) after %d warm-up (%.1f s/step), torch %s, %d threads, %s; build container" % (
                         a.config, kind, H, L, "bi" if bi else "uni", N, lengths.min() * 0.01, lengths.max() * 0.01, len(times), a.warmup,
                         best, torch.__version__, cores, cpu[0] if cpu else "?"),
           "ctc_loss_first_step": loss}
    path = os.path.join(ROOT, "profiles", "cpu_reference_%s.json" % a.config)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
