#!/usr/bin/env python
"""Where does the spread of the per-step times come from?  (2 of 20 driver steps took 26-27 ms against a
median of 22.7.)  Runs the bench's own training step on cfg3 and logs, per step and IN ORDER: the host time to enqueue the step, the
device time between consecutive step starts, the duration of each of the 10 recurrent sweeps (HIP events on the launch stream), and
the time the device spent outside the sweeps.  Prints the steps as a table plus a summary of slow steps vs the rest.

    gpurun -- 'python tools/step_jitter.py --steps 200 > gpurun_out/step_jitter.txt'"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sync-every", type=int, default=0, help="synchronise the device every N steps (0: never inside the timed region)")
    ap.add_argument("--sleep-ms", type=float, default=0.0, help="host sleep after every step's enqueue (lets the device drain: idle gaps on purpose)")
    ap.add_argument("--lag-sync", type=int, default=0, help="before enqueueing step i wait for the START of step i - K + 1 (the host stays at most K steps ahead)")
    ap.add_argument("--prime", type=int, default=0, help="before the warm-up: N tiny launches queued behind a ~40 ms spin kernel (the host runs far ahead)")
    ap.add_argument("--gc-off", action="store_true", help="gc.disable() during the measured steps")
    args = ap.parse_args()
    from deepspeech.pytorch_amd import configs, ops
    from deepspeech.pytorch_amd.model import DeepSpeech
    dev = torch.device("cuda", 0)
    kind, H, L, bi, N, tmin, tmax, dtype = bench.CONFIGS[args.config]
    rt = getattr(configs.RNNType, kind)
    mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L) if bi else \
        configs.UniDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L, lookahead_context=20)
    torch.manual_seed(0)
    model = DeepSpeech(configs.LABELS, mc, "bf16" if dtype == "bf16" else 32, configs.AdamConfig(), configs.SpectConfig())
    bench.load_reference_loss(args.config, model)
    model = model.to(dev).train()
    lengths, batch = bench.build_batch(args.config, 0, dev)
    opt = model.configure_optimizers()[0][0]
    opt.clip_grad_norm = 400.0

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.training_step((batch[0], batch[1], batch[2].clone(), batch[3]), 0)
        loss.backward()
        opt.step()
        return loss

    if args.prime:
        x = torch.zeros(64, device=dev)
        for rep in range(2):
            torch.cuda._sleep(int(40e-3 * 2.0e9))
            evs = []
            for i in range(args.prime):
                x.add_(1.0)
                if i % 8 == 0:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    evs.append(e)
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.gc_off:
        import gc
        gc.collect()
        gc.disable()
    marks, host, sweeps, allocs = [], [], [], []
    a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    for i in range(args.steps):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append(ev)
        if args.lag_sync and i + 1 >= args.lag_sync:
            marks[i + 1 - args.lag_sync].synchronize()
        ops.SWEEP_EVENTS = []
        t0 = time.perf_counter()
        step()
        host.append((time.perf_counter() - t0) * 1e3)
        ms_ = torch.cuda.memory_stats()
        allocs.append((ms_.get("num_device_alloc", 0), ms_.get("reserved_bytes.all.current", 0) >> 20))
        sweeps.append(ops.SWEEP_EVENTS)
        ops.SWEEP_EVENTS = None
        if args.sync_every and (i + 1) % args.sync_every == 0:
            torch.cuda.synchronize()
        if args.sleep_ms:
            time.sleep(args.sleep_ms / 1e3)
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append(ev)
    torch.cuda.synchronize()
    ops.check_persistent_kernels()
    devms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
    sw = np.array([[e0.elapsed_time(e1) for (_, _, e0, e1) in s] for s in sweeps])      # [steps][2L]: L forward then L BPTT sweeps
    tags = [t for (t, _, _, _) in sweeps[0]]
    host = np.array(host)
    print("# %s, %d steps; columns: step, device ms (start to next start), host enqueue ms, sweeps total ms, rest ms, then each sweep in launch order" % (args.config, args.steps))
    print("# sweep order:", " ".join("%s%d" % ("F" if "fwd" in t else "B", i) for i, t in enumerate(tags)))
    for i in range(args.steps):
        print("%4d %7.3f %7.3f %7.3f %7.3f  %s  | device mallocs so far %d, reserved %d MB" % (
            i, devms[i], host[i], sw[i].sum(), devms[i] - sw[i].sum(), " ".join("%.3f" % v for v in sw[i]), allocs[i][0] - a0, allocs[i][1]))
    med = np.median(devms)
    slow = devms > 1.05 * med
    print("# median %.3f ms, mean %.3f, min %.3f, max %.3f, p90 %.3f; %d of %d steps above 1.05 x median" % (
        med, devms.mean(), devms.min(), devms.max(), np.percentile(devms, 90), int(slow.sum()), args.steps))
    print("# host enqueue: median %.3f ms, max %.3f; steps where the host took longer than the device: %d" % (np.median(host), host.max(), int((host > devms).sum())))
    for name, m in (("slow", slow), ("other", ~slow)):
        if m.any():
            print("# %-5s steps: device %.3f ms, sweeps %.3f (fwd %.3f, bwd %.3f), rest %.3f, host %.3f" % (
                name, devms[m].mean(), sw[m].sum(1).mean(), sw[m][:, :len(tags) // 2].sum(1).mean(), sw[m][:, len(tags) // 2:].sum(1).mean(),
                (devms[m] - sw[m].sum(1)).mean(), host[m].mean()))
    print("# per-sweep mean ms over all steps (launch order):", " ".join("%.3f" % v for v in sw.mean(0)))


if __name__ == "__main__":
    main()
