#!/usr/bin/env python
"""Where a half-step of the round-4 general persistent sweeps (ds2_rnn_persist3_impl.h) goes: the -DDS2_PROBE build's cycle counters
(gather + products | partial sums + barrier | gate math + publish, re-polls) of workgroup 0 of every group, and the wall time per time
step, with work switched off piece by piece (DS2_PERSIST_DBG bits: 1 no prefetch loads of the gate operands, 2 no output stores, 8 no
gather and no products, 32 no products with the LDS-resident fragments, 64 plain-store publishes whatever the placement, 128 gather but no
products).  Timing only: with any bit set the results are wrong.

    gpurun -- 'python tools/probe_persist3.py > gpurun_out/probe_persist3.txt'
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepspeech.pytorch_amd import _lib, build  # noqa: E402

_lib.LIB_PATH = build.build(probe=True, verbose=False)
from deepspeech.pytorch_amd import ops  # noqa: E402

ops._OPTS["variant"] = int(os.environ.get("DS2_PROBE_VARIANT", "64"))     # 64: dense 16-row tiles for <= 16 clips (the counters' nset assumes it)

CASES = [("lstm", 2, 64, 1280, 751, "cfg5a"), ("lstm", 1, 64, 1280, 751, "cfg5b"), ("gru", 2, 8, 800, 401, "bf16 GRU-800 bi (XCD-local)"),
         ("lstm", 2, 128, 1024, 301, "LSTM-1024 bi, 128 clips (XCD-local, 2 sets)")]
MASKS = [0, 3, 8, 32, 128]     # (64: plain-store publishes -- only meaningful for XCD-local groups, stale forever otherwise)
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[5].split()[0] in sys.argv[1:]] or CASES
dev = "cuda"


def counters(Tp, nset):
    ws = ops.LAST_PERSIST_WS
    tail = ws[:1024].view(torch.int64).cpu().numpy().reshape(-1, 8)
    out = []
    for g in (0, 1):
        for wv, o in (("w0", 0), ("w3", 4)):
            c = tail[g][o:o + 4]
            out.append("g%d %s: gather+mma %5.0f  store+barrier %5.0f  gate+publish %5.0f = %5.0f cyc per half-step, %.2f re-polls" % (
                g, wv, c[0] / Tp / nset, c[1] / Tp / nset, c[2] / Tp / nset, (c[0] + c[1] + c[2]) / Tp / nset, c[3] / Tp / nset))
    return "\n        ".join(out)


for kind, D, N, H, Tp, label in CASES:
    G = ops.GATES[kind]
    torch.manual_seed(0)
    GI = torch.randn(Tp * N, D * G * H, device=dev).to(torch.bfloat16)
    Whh = ((torch.rand(D, G * H, H, device=dev) * 2 - 1) / H ** 0.5).to(torch.bfloat16)
    WhhT = Whh.transpose(1, 2).contiguous()
    bhh = torch.zeros(D, G * H, device=dev)
    lens = torch.from_numpy(np.sort(np.random.RandomState(0).randint(Tp // 3, Tp + 1, N))[::-1].copy().astype(np.int32)).to(dev)
    lens[0] = Tp
    dout = torch.randn(Tp, N, H, device=dev).to(torch.bfloat16)
    fam = ops.persist_kind(torch.bfloat16, kind, D, N, H)
    slots = (8 * (32 // (H // 32))) if H // 32 <= 32 else 256 // (H // 32)
    gpd = min(slots // D, N)
    ns = -(-N // gpd)
    nset = 1 if ns <= 16 else 2
    print("==== %s: %s D=%d N=%d H=%d T'=%d: kernel family %d, %d groups of %d workgroups, %d samples per group in %d set(s)" % (
        label, kind, D, N, H, Tp, fam, gpd * D, H // 32, ns, nset), flush=True)
    for mask in MASKS:
        os.environ["DS2_PERSIST_DBG"] = str(mask)
        res = {}
        for which in ("fwd", "bwd"):
            best = None
            for it in range(2):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if which == "fwd":
                    hext, Sv, hn, cn = ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
                else:
                    ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp)
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1)
                if best is None or t < best[0]:
                    best = (t, counters(Tp, nset))
            res[which] = best
        for which in ("fwd", "bwd"):
            t, c = res[which]
            print("  mask %3d %s: %.3f ms = %.3f us per time step\n        %s" % (mask, which, t, t * 1e3 / Tp, c), flush=True)
    os.environ["DS2_PERSIST_DBG"] = "0"
    try:
        ops.check_persistent_kernels()
    except Exception as e:  # noqa: BLE001
        print("  !! time-out:", e)
