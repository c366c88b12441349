#!/usr/bin/env python
"""bench.py -- training-step throughput of the MI355X-native DeepSpeech2 hot path.

    python bench.py [--gpus N --steps K --warmup W]                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                            # N GPUs, one rank per GPU over RCCL

Metric (BASELINE.json / SURVEY.md section 8d): audio-seconds/sec of one training step
    zero_grad -> training_step (forward + CTC) -> backward [-> gradient all-reduce] -> clip_grad_norm_(400) -> AdamW.step
on the LibriSpeech-shaped configuration cfg3: 5 x BiGRU hidden=1024, batch 32 clips/GPU of 12-15 s, bf16 activations /
MFMA operands with fp32 master weights, statistics, gate math and CTC; synthetic 16 kHz log-spectrograms already
resident in HBM when the timed region starts; weak scaling (32 clips per GPU).  Audio seconds = true (unpadded) frames
x 10 ms.  One JSON line is printed by rank 0; it also carries

  "roofline":     the dominant kernel (the persistent recurrent sweep k_rnn_persist_bwd / _fwd: ONE launch per layer and
                  sweep): algorithmic h2h FLOPs of a sweep over valid frames / average launch duration measured live with
                  HIP events on the launch stream, against the dense MFMA peak of the compute dtype (bf16 2.5 PFLOP/s, fp32
                  157.3 TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md);
  "cpu_baseline": oracle/ds2_torch_port.py (a port of the reference's model.py:214-249: the same torch calls in the same order)
                  timed on the host cores of THIS box (rank 0, N=1 only), fp32, all usable cores.  For cfg3 -- the metric's
                  configuration -- `value` is ONE training step at the bench's own full batch (32 clips of 12-15 s; minutes of CPU
                  time, no same-shape warm-up); `short_sample` is the bounded sample (same model and batch size, 3-s clips) and
                  `reference_offline` the reference's own model.py timed offline in the build container
                  (tools/cpu_reference_baseline.py, profiles/cpu_reference_<config>.json).  A reported baseline, not the target.

  "stock_baseline" / "vs_baseline": stock PyTorch-ROCm (the `--stock` leg below) timed by THIS run in a subprocess on the same
                  batch, the same parameters and the same device (N=1, rank 0; --no-stock-baseline skips it), and value / its value;
  "ctc_loss_ref" / "ctc_loss_rel_diff": the reference's fp32 CPU loss of this very batch and parameters (generated offline from the
                  real reference, tests/golden/) against the first timed-path loss -- the line proves it computed the same thing;
  "ms_per_step_median", "step_time_distribution": per-step HIP-event times of the timed steps (the mean is `ms_per_step`).

With `--gpus N` and no launcher environment (WORLD_SIZE unset) the script re-executes itself under torch.distributed.run
with N ranks on 127.0.0.1.

`--stock` instead times stock PyTorch-ROCm (oracle/ds2_torch_port.py on the GPU under bf16 autocast: MIOpen conv / BN /
RNN + ATen CTC) on the same batch: the denominator of the north star's ">= 3x over stock PyTorch-ROCm".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md chip table
CONFIGS = {
    # name: (rnn_type, H, L, bidirectional, N per GPU, Tmin, Tmax, dtype)
    "cfg3": ("gru", 1024, 5, True, 32, 1201, 1501, "bf16"),      # LibriSpeech-shaped (the headline configuration)
    "cfg2": ("gru", 800, 5, True, 8, 101, 201, "f32"),           # AN4-shaped, fp32 parity mode
    "cfg5a": ("lstm", 1280, 7, True, 64, 501, 1501, "bf16"),
    "cfg5b": ("lstm", 1280, 7, False, 64, 501, 1501, "bf16"),
}
GATES = {"gru": 3, "lstm": 4, "rnn": 1}


def build_batch(cfg_name, rank, device):
    from deepspeech.pytorch_amd import synth
    kind, H, L, bi, N, tmin, tmax, dtype = CONFIGS[cfg_name]
    cfg_id = {"cfg2": 2, "cfg3": 3, "cfg5a": 5, "cfg5b": 6}[cfg_name]
    lengths = synth.synth_lengths(N, tmin, tmax, seed=cfg_id * 1000 + rank, linear=(cfg_name == "cfg2"))
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=cfg_id * 1000 + rank)
    batch = (torch.from_numpy(inputs).to(device), torch.from_numpy(targets), torch.from_numpy(pct), torch.from_numpy(tsz))
    return lengths, batch


def out_frames(lengths):
    """Output frames per sample after the conv stack (reference model.py:299-310), incl. the float32 percentage round
    trip of model.py:243."""
    lengths = np.asarray(lengths, dtype=np.int64)
    tmax = int(lengths.max())
    rt = ((lengths / float(tmax)).astype(np.float32) * np.float32(tmax)).astype(np.int32).astype(np.int64)
    return (rt + 2 * 5 - 10 - 1) // 2 + 1


def train_flops(cfg_name, lengths):
    kind, H, L, bi, N, tmin, tmax, dtype = CONFIGS[cfg_name]
    D, G = (2 if bi else 1), GATES[kind]
    frames = int(out_frames(lengths).sum())
    macs = 32 * 81 * 41 * 11 + 32 * 41 * 21 * 11 * 32 + D * G * H * 1312 + (L - 1) * D * G * H * H + L * D * G * H * H + 29 * H
    if not bi:
        macs += H * 20
    return 3 * 2.0 * macs * frames


PARAM_SEED = {"cfg2": 2202, "cfg3": 3303, "cfg5a": 5505, "cfg5b": 6606}     # tests/golden/make_fullsize_golden.py


def load_reference_loss(cfg_name, model):
    """Random-init weights of the architecture, drawn the way the full-size reference fixture draws them (synth.synth_params, the
    fixture's seed), so that the FIRST step of the bench on rank 0's batch is exactly the step the reference's own model.py was run
    on when tests/golden/full/<config>.npz was generated (make_fullsize_golden.py: same batch seed, same parameters): its CTC loss
    is the `ctc_loss_ref` of the JSON line.  Returns {"fp32": loss of the reference as shipped, "autocast": under bf16 autocast}."""
    from deepspeech.pytorch_amd import synth
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    P = synth.synth_params(shapes, PARAM_SEED[cfg_name])
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    path = os.path.join(ROOT, "tests", "golden", "full", cfg_name + ".npz")
    if not os.path.exists(path):
        return None
    z = np.load(path)
    return {"fp32": float(z["loss"]), "autocast": float(z["loss_ac"]) if "loss_ac" in z.files else None}


def run_native(args, rank, world, device):
    from deepspeech.pytorch_amd import configs, ops
    from deepspeech.pytorch_amd import dist as dsdist
    from deepspeech.pytorch_amd.model import DeepSpeech
    kind, H, L, bi, N, tmin, tmax, dtype = CONFIGS[args.config]
    torch.manual_seed(0)
    rt = getattr(configs.RNNType, kind)
    mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L) if bi else \
        configs.UniDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L, lookahead_context=20)
    model = DeepSpeech(configs.LABELS, mc, "bf16" if dtype == "bf16" else 32, configs.AdamConfig(), configs.SpectConfig())
    ref = load_reference_loss(args.config, model)          # parameters of the full-size reference fixture (same on every rank)
    model = model.to(device)
    model.train()
    lengths, batch = build_batch(args.config, rank, device)
    opt = model.configure_optimizers()[0][0]       # FusedAdamW: clip + AdamW + next step's bf16 weight layouts on HIP kernels
    fused_clip = hasattr(opt, "clip_grad_norm") and not args.torch_optimizer
    if args.torch_optimizer:
        opt = torch.optim.AdamW(model.parameters(), lr=model.optim_cfg.learning_rate, betas=tuple(model.optim_cfg.betas),
                                eps=model.optim_cfg.eps, weight_decay=model.optim_cfg.weight_decay, fused=True)
    if fused_clip:
        opt.clip_grad_norm = 400.0                 # Lightning's gradient_clip_val: 400 (reference configs/*.yaml)
    step_mod = dsdist.wrap_data_parallel(dsdist.StepModule(model), device, world)
    params = [p for p in model.parameters()]

    marks = []                                     # one event per timed step start (+ one at the end): per-step durations, no host sync

    def step():
        if marks is not None and ops.SWEEP_EVENTS is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
        opt.zero_grad(set_to_none=True)
        loss = step_mod(batch[0], batch[1], batch[2].clone(), batch[3])
        loss.backward()
        if finish_backward is not None:
            finish_backward()
        if not fused_clip:
            torch.nn.utils.clip_grad_norm_(params, 400.0)
        opt.step()
        return loss

    finish_backward = getattr(step_mod, "finish_backward", None)     # opt-in OverlappedDataParallel (no-op under DDP)
    first_loss = None
    log("model + batch ready; warm-up")
    for i in range(args.warmup):
        l = step()
        if i == 0:
            first_loss = float(l.detach().item())
            log("first step done, loss %.4f" % first_loss)
    if first_loss is None:
        first_loss = float("nan")
    log("timed region")
    ops.SWEEP_EVENTS = []
    dt, last = dsdist.timed_steps(step, args.steps, device, world)
    events, ops.SWEEP_EVENTS = ops.SWEEP_EVENTS, None
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    end.synchronize()
    marks.append(end)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1))
    step_stats = {"median_ms": round(per_step[len(per_step) // 2], 3), "min_ms": round(per_step[0], 3), "max_ms": round(per_step[-1], 3),
                  "p90_ms": round(per_step[min(len(per_step) - 1, int(0.9 * len(per_step)))], 3),
                  "how": "HIP events at every step start on the step's stream (device time between consecutive steps)"} if per_step else None
    log("timed region done: %.1f ms/step" % (dt / args.steps * 1e3))
    last_loss = float(last.detach().item())
    # ---- data-parallel evidence (world > 1, or DS2_FORCE_DDP=1 on one GPU): what the exchange costs, and that the ranks are N GPUs
    dp_info = None
    is_ddp = isinstance(step_mod, torch.nn.parallel.DistributedDataParallel)
    if world > 1 or is_ddp:
        import torch.distributed as dist_
        dp_info = {"wrapper": type(step_mod).__name__}
        if is_ddp:
            # the same step with the gradient exchange switched off (DDP.no_sync): step time with - without = what the all-reduce
            # costs the step after overlap (exposed tail + contention under the sweeps), measured, not modelled
            k2 = max(3, min(args.steps, 10))
            ns_marks = []

            def step_marked():
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                ns_marks.append(ev)
                return step()
            with step_mod.no_sync():              # (ops.SWEEP_EVENTS is None again: step() itself records no events in this leg)
                for _ in range(2):
                    step()
                dt_ns, _ = dsdist.timed_steps(step_marked, k2, device, world)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            ev.synchronize()
            ns_marks.append(ev)
            ns_steps = sorted(ns_marks[i].elapsed_time(ns_marks[i + 1]) for i in range(k2))
            med_with = step_stats["median_ms"] if step_stats else dt / args.steps * 1e3
            t = torch.tensor([dt / args.steps * 1e3, dt_ns / k2 * 1e3, med_with, ns_steps[k2 // 2]], dtype=torch.float64, device=device)
            if dist_.is_initialized():
                dist_.all_reduce(t, op=dist_.ReduceOp.MAX)
            dp_info.update(ms_per_step_with_allreduce=round(float(t[0]), 3), ms_per_step_no_sync=round(float(t[1]), 3),
                           median_ms_with_allreduce=round(float(t[2]), 3), median_ms_no_sync=round(float(t[3]), 3),
                           comm_exposed_ms=round(float(t[2] - t[3]), 3), comm_exposed_ms_from_means=round(float(t[0] - t[1]), 3),
                           no_sync_steps=k2,
                           comm_exposed_how="max-over-ranks MEDIAN per-step device time of the timed region minus the same step under "
                                            "DDP.no_sync() (%d steps right after it): exposed all-reduce tail + its contention with the "
                                            "sweeps; the means are given too (one host-side stall in a 20-step window moves them by ms)" % k2)
        mine = {"rank": rank, "device": torch.cuda.get_device_name(device), "uuid": str(getattr(torch.cuda.get_device_properties(device), "uuid", "?")),
                "pci_bus_id": getattr(torch.cuda.get_device_properties(device), "pci_bus_id", None),
                "step_ms_min": step_stats["min_ms"] if step_stats else None, "step_ms_median": step_stats["median_ms"] if step_stats else None,
                "step_ms_max": step_stats["max_ms"] if step_stats else None, "seconds_timed_region": round(dt, 4)}
        if dist_.is_initialized() and world > 1:
            gathered = [None] * world
            dist_.all_gather_object(gathered, mine)
        else:
            gathered = [mine]
        dp_info["ranks"] = gathered
        dp_info["distinct_devices"] = len({g["uuid"] for g in gathered})

    ops.check_persistent_kernels()
    # ---- roofline of the dominant kernel: the recurrent sweep (persistent kernel: ONE launch per layer and sweep; the
    #      per-time-step kernels of ds2_rnn.hip when the persistent path does not cover the shape).  Durations come from
    #      HIP events recorded around every sweep launch on the launch stream, inside the timed region.
    D, G = (2 if bi else 1), GATES[kind]
    frames = int(out_frames(lengths).sum())
    tp = int(out_frames(lengths).max())
    flops_per_sweep = 2.0 * frames * D * G * H * H          # algorithmic: valid frames only
    peak = PEAK_TFLOPS[dtype]
    per_kind = {}
    for tag, launches, e0, e1 in events:
        d = per_kind.setdefault(tag, [0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
    kernels = {}
    for tag, (n, ms) in per_kind.items():
        persistent = tag.endswith("persistent")
        avg_s = ms / 1e3 / n / (1 if persistent else tp)       # per launch
        fl = flops_per_sweep if persistent else flops_per_sweep / tp
        kernels[tag] = {"launches_per_step": (n if persistent else n * tp) // max(1, args.steps), "avg_launch_us": round(avg_s * 1e6, 2),
                        "us_per_time_step": round(ms * 1e3 / n / tp, 3), "achieved_tflops": round(fl / avg_s / 1e12, 2),
                        "ms_per_train_step": round(ms / max(1, args.steps), 3)}
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_train_step"]) if kernels else None
    # the kernel symbols behind the tags (what rocprofv3 lists), from the library's own dispatch decision
    from deepspeech.pytorch_amd._lib import query
    fam = ops.persist_kind(torch.bfloat16 if dtype == "bf16" else torch.float32, kind, D, N, H)
    stem = {1: ("k_rnn_persist_fwd4", "k_rnn_persist_bwd4"), 2: ("k_rnn_persist_fwd", "k_rnn_persist_bwd"),
            3: ("k_rnn_persist3_fwd", "k_rnn_persist3_bwd"), 4: ("k_rnn_persist2_fwd", "k_rnn_persist2_bwd")}.get(fam, ("?", "?"))
    names = {"rnn_fwd_persistent": stem[0], "rnn_bwd_persistent": stem[1], "rnn_fwd": "k_rnn_step_fwd", "rnn_bwd": "k_rnn_step_bwd"}
    traffic, traffic_src = None, None
    try:   # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/pmc_traffic.json)
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pm.get("config") == args.config and names.get(dom, dom) in pm["kernels"]:
            k = pm["kernels"][names.get(dom, dom)]
            traffic, traffic_src = int(k["fetch_bytes"] + k["write_bytes"]), pm["source"]
    except Exception:
        pass
    roofline = {"bound": "mfma", "kernel": names.get(dom, dom), "achieved": kernels[dom]["achieved_tflops"] if dom else 0.0,
                "peak": peak, "unit": "TFLOP/s", "frac": round((kernels[dom]["achieved_tflops"] if dom else 0.0) / peak, 6),
                "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": kernels[dom]["avg_launch_us"] if dom else None,
                "us_per_time_step": kernels[dom]["us_per_time_step"] if dom else None,
                "note": "latency-bound serial recurrence: 2*L*T' dependent steps per train step; flops are the algorithmic "
                        "h2h products over valid frames",
                "recurrent_kernels": {names.get(k, k): v for k, v in kernels.items()},
                "whole_step_frac_of_mfma_roofline": None}
    roofline["_step_stats"] = step_stats
    roofline["_loss_ref"] = ref
    roofline["_dp_info"] = dp_info
    return lengths, dt, first_loss, last_loss, roofline


def run_stock(args, rank, world, device):
    """Stock PyTorch-ROCm on the same batch (bf16 autocast for cfg3, as the reference's precision=16 does)."""
    from oracle import ds2_torch_port as TP
    kind, H, L, bi, N, tmin, tmax, dtype = CONFIGS[args.config]
    cfg = dict(rnn_type=kind, hidden_size=H, hidden_layers=L, bidirectional=bi, lookahead_context=20)
    from deepspeech.pytorch_amd import synth
    st0 = TP.random_state(cfg, 0)
    st0 = synth.synth_params({k: tuple(np.asarray(v).shape) for k, v in st0.items()}, PARAM_SEED[args.config])   # the native leg's parameters
    port = TP.Port(cfg, st0, device)
    lengths, batch = build_batch(args.config, rank, device)
    opt = port.make_optimizer()
    ac = torch.bfloat16 if dtype == "bf16" else None
    first = None
    for i in range(args.warmup):
        l = port.train_step(batch, opt, ac)
        if i == 0:
            first = float(l.detach().item())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = port.train_step(batch, opt, ac)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return lengths, dt, first if first is not None else float("nan"), float(last.detach().item()), None


def usable_cores():
    """Cores this process may actually use: affinity mask and cgroup CPU quota (os.cpu_count() reports the host's)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 64))


def log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def other_config_subprocess(args, cfg):
    """`bench.py --config <cfg>` (native path, a few steps, no baselines) in a child process under a time limit: the configurations
    the driver does not time itself (cfg2: the fp32 parity configuration; cfg5a / cfg5b: 7 x LSTM-1280, batch 64) ride along with the
    headline line as `other_configs`, timed after it and outside `value`."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", str(args.other_steps), "--warmup", "3",
           "--no-cpu-baseline", "--no-stock-baseline", "--no-other-configs"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.other_timeout, env=env)
        j = _last_json_line(r.stdout)
        if j is None:
            return {"error": (r.stderr.strip().splitlines() or ["?"])[-1][:200]}
        rf = j.get("roofline") or {}
        return {"ms_per_step": j["ms_per_step"], "ms_per_step_median": j.get("ms_per_step_median"), "value": j["value"], "unit": j["unit"],
                "steps": j["steps"], "dtype": j["dtype"], "workload": j["config"]["workload"],
                "ctc_loss_first_step": j.get("ctc_loss_first_step"), "ctc_loss_ref": j.get("ctc_loss_ref"),
                "ctc_loss_rel_diff": j.get("ctc_loss_rel_diff"),
                "roofline_kernel": rf.get("kernel"), "roofline_frac": rf.get("frac"), "us_per_time_step": rf.get("us_per_time_step"),
                "recurrent_kernels": rf.get("recurrent_kernels"), "whole_step_frac_of_mfma_roofline": rf.get("whole_step_frac_of_mfma_roofline")}
    except subprocess.TimeoutExpired:
        return {"error": "exceeded its %d s limit" % args.other_timeout}


def _last_json_line(text):
    for line in (text or "").splitlines()[::-1]:
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def cpu_baseline_subprocess(args):
    """Runs the CPU leg in a child process under a hard time limit so that it can never stall the GPU bench.  The child prints a
    line per leg (short sample, then the full shape); the last complete one counts, also when the limit cuts the child off."""
    import subprocess
    full = args.cpu_full
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--config", args.config,
           "--cpu-clips", str(args.cpu_clips), "--cpu-frames", str(args.cpu_frames), "--cpu-full", "1" if full else "0"]
    limit = args.cpu_timeout if args.cpu_timeout > 0 else (1200 if full else 170)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit)
        j = _last_json_line(r.stdout)
        if j is not None:
            return j
        return {"value": None, "unit": "audio-seconds/sec", "cores": usable_cores(), "kind": "port",
                "sample": "cpu leg failed: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200]}
    except subprocess.TimeoutExpired as e:
        so = e.stdout.decode() if isinstance(e.stdout, bytes) else e.stdout
        j = _last_json_line(so)
        if j is not None:
            j["note"] = "the full-shape leg exceeded the %d s limit: this is the short sample" % limit
            return j
        return {"value": None, "unit": "audio-seconds/sec", "cores": usable_cores(), "kind": "port",
                "sample": "cpu leg exceeded its %d s limit" % limit}


def stock_baseline_subprocess(args):
    """`bench.py --stock` (stock PyTorch-ROCm on this GPU, same configuration) in a child process under a time limit: the
    denominator of the north star's ">= 3x over stock PyTorch-ROCm", measured in the same driver run as the native number."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--stock", "--config", args.config, "--steps", str(args.stock_steps),
           "--warmup", str(args.stock_warmup), "--no-cpu-baseline"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    base = {"impl": "stock-pytorch-rocm", "unit": "audio-seconds/sec", "steps": args.stock_steps, "warmup": args.stock_warmup}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.stock_timeout, env=env)
        for line in r.stdout.splitlines()[::-1]:
            if line.startswith("{"):
                j = json.loads(line)
                base.update(value=j["value"], ms_per_step=j["ms_per_step"], dtype=j["dtype"], ctc_loss_first_step=j.get("ctc_loss_first_step"),
                            torch=torch.__version__)
                return base
        base.update(value=None, note="stock leg failed: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200])
    except subprocess.TimeoutExpired:
        base.update(value=None, note="stock leg exceeded its %d s limit" % args.stock_timeout)
    return base


def cpu_baseline(args):
    """oracle/ds2_torch_port.py -- a port of the reference's forward / training_step (model.py:214-249: the same torch calls in the
    same order) -- on the host cores of THIS box, fp32, all usable cores.  Two legs, each printed as its own JSON line (the parent
    takes the last one it got, so a full-shape leg that outlives the time limit still leaves the short sample):
      1. a bounded sample: the configuration's model and batch size on short clips (--cpu-frames), best of <= 4 steps;
      2. (--cpu-full, default for cfg3) the metric's own configuration: rank 0's FULL batch of the bench (the same clips the GPU
         leg trains on: 32 clips of 12-15 s for cfg3), ONE timed step, no same-shape warm-up (a step is minutes of CPU time; the
         thread pool and allocator are warm from leg 1).  This is `value`; leg 1 moves to `short_sample`."""
    from deepspeech.pytorch_amd import synth
    from oracle import ds2_torch_port as TP
    kind, H, L, bi, N, tmin, tmax, dtype = CONFIGS[args.config]
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = dict(rnn_type=kind, hidden_size=H, hidden_layers=L, bidirectional=bi, lookahead_context=20)
    st0 = TP.random_state(cfg, 0)
    st0 = synth.synth_params({k: tuple(np.asarray(v).shape) for k, v in st0.items()}, PARAM_SEED[args.config])   # the native leg's parameters
    port = TP.Port(cfg, st0, "cpu")
    n = args.cpu_clips if args.cpu_clips > 0 else N          # default: the configuration's own batch size (the CPU path is
    t = args.cpu_frames                                      # weight-bandwidth-bound per time step, so the batch size matters)
    opt = port.make_optimizer()
    what = "oracle/ds2_torch_port.py (port of the reference's model.py:214-249: the same torch op sequence) fp32, same model (%s H=%d L=%d %s)" % (
        kind, H, L, "bi" if bi else "uni")

    def make(n_, t_, seed):
        lengths = synth.synth_lengths(n_, max(41, t_ - t_ // 5), t_, seed=seed)
        inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=seed)
        return lengths, (torch.from_numpy(inputs), torch.from_numpy(targets), torch.from_numpy(pct), torch.from_numpy(tsz))
    port.train_step(make(2, 61, 76)[1], opt)          # warm-up (thread pool, allocator) on a tiny batch
    lengths, batch = make(n, t, 77)
    times = []
    t_all = time.perf_counter()
    while len(times) < 1 or (time.perf_counter() - t_all < 14.0 and len(times) < 4):
        t0 = time.perf_counter()
        port.train_step(batch, opt)
        times.append(time.perf_counter() - t0)
    best = min(times)
    secs = synth.audio_seconds(lengths)
    out = {"value": round(secs / best, 3), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
           "sample": "%s, %d clips of %.2f-%.2f s, best of %d steps (%.2f s/step), torch %s, %d threads" % (
               what, n, lengths.min() * 0.01, lengths.max() * 0.01, len(times), best, torch.__version__, cores)}
    try:   # the reference's own model.py at the FULL shape, timed offline in the build container (no reference on this box)
        ref = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference_%s.json" % args.config)))
        out["reference_offline"] = {k: ref[k] for k in ("value", "unit", "cores", "kind", "sample") if k in ref}
    except Exception:
        pass
    if not args.cpu_full:
        return out
    print(json.dumps(out), flush=True)                 # what the parent falls back to if the full-shape leg runs out of time
    short = {k: out[k] for k in ("value", "unit", "sample")}
    # ---- the metric's own configuration: rank 0's batch of the bench, fresh parameters and optimizer state (the fixture's)
    port = TP.Port(cfg, st0, "cpu")
    opt = port.make_optimizer()
    lengths, batch = build_batch(args.config, 0, "cpu")
    t0 = time.perf_counter()
    loss = port.train_step(batch, opt)
    dt = time.perf_counter() - t0
    secs = synth.audio_seconds(lengths)
    out.update(value=round(secs / dt, 4), short_sample=short, seconds_per_step=round(dt, 2), ctc_loss_first_step=float(loss.detach().item()),
               timed_steps=1, same_shape_warmup_steps=0,     # "n = 1, cold": SURVEY 8(d) asks >= 1 warm-up + >= 3 timed; a step is ~6 min here
               sample="n = 1, cold (1 timed step, 0 same-shape warm-ups): %s, the bench's own rank-0 batch: %d clips of %.2f-%.2f s (%.1f s of audio, the metric's configuration %s), ONE timed "
                      "training step (forward + CTC + backward + clip_grad_norm(400) + AdamW) of %.1f s with 0 same-shape warm-up steps "
                      "(thread pool / allocator warm from the short sample), torch %s, %d threads on this box's host cores" % (
                          what, len(lengths), lengths.min() * 0.01, lengths.max() * 0.01, secs, args.config, dt, torch.__version__, cores))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default: a >= 5 s timed region on cfg3: 200 x ~30 ms)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--stock", action="store_true", help="time stock PyTorch-ROCm instead of the native path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--torch-optimizer", action="store_true", help="clip_grad_norm_ + torch AdamW(fused=True) instead of the HIP optimizer kernels")
    ap.add_argument("--cpu-clips", type=int, default=0, help="clips of the CPU sample (0 = the configuration's batch size)")
    ap.add_argument("--cpu-frames", type=int, default=301)
    ap.add_argument("--cpu-timeout", type=int, default=0, help="limit of the CPU leg in seconds (0: 1200 with the full-shape step, 170 without)")
    ap.add_argument("--cpu-full", type=int, default=-1, help="1: time ONE training step of the port at the configuration's full batch on the "
                    "host cores (minutes); 0: the short sample only; default: 1 for cfg3 (the metric's configuration), 0 otherwise")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--no-stock-baseline", action="store_true", help="skip the stock PyTorch-ROCm leg of the N=1 run")
    ap.add_argument("--stock-steps", type=int, default=3)
    ap.add_argument("--stock-warmup", type=int, default=2)
    ap.add_argument("--stock-timeout", type=int, default=150)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the cfg2 / cfg5a / cfg5b legs of the default N=1 run")
    ap.add_argument("--other-configs", default="cfg5a,cfg5b,cfg2", help="configurations timed after the headline (N=1, --config cfg3 only)")
    ap.add_argument("--other-steps", type=int, default=10)
    ap.add_argument("--other-timeout", type=int, default=120)
    args = ap.parse_args()
    if args.cpu_full < 0:
        args.cpu_full = 1 if args.config == "cfg3" else 0
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start N ranks (one per GPU, RCCL over 127.0.0.1) of this same command and relay rank 0's JSON line
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    from deepspeech.pytorch_amd import dist as dsdist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path for the product)")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, world, local_rank = dsdist.init_from_env("nccl")
    device = torch.device("cuda", local_rank)
    if args.gpus != world:
        # a launcher decided the rank count (e.g. `torchrun --nproc-per-node 8 bench.py` with the default --gpus 1): follow it
        if rank == 0:
            log("--gpus %d but the launcher started WORLD_SIZE=%d ranks: reporting n_gpus = %d" % (args.gpus, world, world))
        args.gpus = world

    runner = run_stock if args.stock else run_native
    log("start %s on %d rank(s), config %s" % ("stock" if args.stock else "native", world, args.config))
    lengths, dt, first_loss, last_loss, roofline = runner(args, rank, world, device)

    from deepspeech.pytorch_amd import synth
    secs_local = synth.audio_seconds(lengths)
    dt_max, secs_total = dsdist.aggregate(dt, secs_local, device, world)
    kind, H, L, bi, N, tmin, tmax_, dtype = CONFIGS[args.config]
    value = secs_total * args.steps / dt_max
    ms_per_step = dt_max / args.steps * 1e3
    if roofline is not None:
        # whole-step view: algorithmic train FLOPs over valid frames / step time / MFMA peak
        roofline["whole_step_frac_of_mfma_roofline"] = round(train_flops(args.config, lengths) / (ms_per_step / 1e3) / 1e12 /
                                                             PEAK_TFLOPS[dtype], 6)
    step_stats = roofline.pop("_step_stats", None) if roofline is not None else None
    loss_ref = roofline.pop("_loss_ref", None) if roofline is not None else None
    dp_info = roofline.pop("_dp_info", None) if roofline is not None else None
    out = {
        "metric": "audio-seconds/sec (train step)", "value": round(value, 2), "unit": "audio-seconds/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic", "rccl_ranks": world,
        "impl": "stock-pytorch-rocm" if args.stock else "ds2hip",
        "config": {"workload": "%s: %dx %s%s hidden=%d, %d clips/GPU of %.1f-%.1f s, train step incl. clip_grad_norm(400)+AdamW" % (
            args.config, L, "Bi" if bi else "Uni", kind.upper(), H, N, tmin * 0.01, tmax_ * 0.01),
            "global_batch": N * world, "frames_max": int(tmax_), "parallelism": "dp%d" % world,
            "audio_seconds_per_step": round(secs_total, 2)},
        "ctc_loss_first_step": first_loss, "ctc_loss_last_step": last_loss,
    }
    if step_stats is not None:
        out["ms_per_step_median"] = step_stats["median_ms"]
        out["step_time_distribution"] = step_stats
    if loss_ref is not None and first_loss == first_loss:
        # BASELINE.json's metric is "audio-s/s + CTC loss vs ref": the first step runs on the full-size fixture's parameters and
        # (on rank 0) its batch, so its loss is directly comparable with the reference's own model.py on the CPU
        ref_mode = loss_ref["autocast"] if (dtype == "bf16" and loss_ref.get("autocast") is not None) else loss_ref["fp32"]
        out["ctc_loss_ref"] = loss_ref["fp32"]
        out["ctc_loss_ref_autocast_bf16"] = loss_ref.get("autocast")
        out["ctc_loss_rel_diff"] = round(abs(first_loss - loss_ref["fp32"]) / abs(loss_ref["fp32"]), 8)
        out["ctc_loss_rel_diff_vs_same_precision_ref"] = round(abs(first_loss - ref_mode) / abs(ref_mode), 8)
        out["ctc_loss_ref_source"] = ("tests/golden/full/%s.npz: the reference's model.py (imported unmodified) on the CPU, same parameters "
                                      "(synth.synth_params seed %d), rank 0's batch" % (args.config, PARAM_SEED[args.config]))
    if roofline is not None:
        out["roofline"] = roofline
    if dp_info is not None:
        out["data_parallel"] = dp_info
    if rank == 0 and world == 1 and not args.stock and not args.no_stock_baseline:
        log("stock PyTorch-ROCm leg (subprocess, <= %d s)" % args.stock_timeout)
        sb = stock_baseline_subprocess(args)
        out["stock_baseline"] = sb
        if sb.get("value"):
            out["vs_baseline"] = round(value / sb["value"], 3)
            out["vs_baseline_kind"] = ("this run's audio-s/s over stock PyTorch-ROCm's (MIOpen conv / BatchNorm / RNN + ATen CTC, same "
                                       "model, batch and precision) measured on the same GPU right after the timed region; BASELINE.md "
                                       "holds no published number for this metric")
    if args.no_cpu_baseline and args.no_stock_baseline:
        args.no_other_configs = True          # "just the number" runs (tools/*.sh, A/B scripts) skip every extra leg
    if rank == 0 and world == 1 and not args.stock and not args.no_other_configs and args.config == "cfg3":
        out["other_configs"] = {}
        for cfg in [c for c in args.other_configs.split(",") if c in CONFIGS and c != args.config]:
            log("other configuration %s (subprocess, <= %d s)" % (cfg, args.other_timeout))
            out["other_configs"][cfg] = other_config_subprocess(args, cfg)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.stock:
        log("cpu baseline leg (subprocess%s)" % (": short sample, then ONE full-shape step on the host cores -- minutes" if args.cpu_full else ""))
        out["cpu_baseline"] = cpu_baseline_subprocess(args)
    if rank == 0:
        print(json.dumps(out))
    dsdist.shutdown(world)


if __name__ == "__main__":
    main()
