#!/usr/bin/env python
"""Timeline of a few steps of the 8-clip recurrent sweeps (cfg3 shape) from the probe build: s_memtime stamps of wave 0 of all 32
workgroups of group 0 -- step top, gather issued, gather + products done, barrier passed, publish issued, step end -- printed relative
to the earliest publish of the step before (= when the first piece of the step's input left its producer).

    gpurun -- 'python tools/probe_timeline.py > gpurun_out/timeline.txt'"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepspeech.pytorch_amd import _lib, build  # noqa: E402

_lib.LIB_PATH = build.build(probe=True, verbose=False)
from deepspeech.pytorch_amd import ops  # noqa: E402

TL_N, TL_K = 8, 6
kind = sys.argv[1] if len(sys.argv) > 1 else "gru"
D, N, H, Tp = 2, 32, 1024, 751
G = ops.GATES[kind]
dev = "cuda"
torch.manual_seed(0)
GI = torch.randn(Tp * N, D * G * H, device=dev).to(torch.bfloat16)
Whh = ((torch.rand(D, G * H, H, device=dev) * 2 - 1) / H ** 0.5).to(torch.bfloat16)
WhhT = Whh.transpose(1, 2).contiguous()
bhh = torch.zeros(D, G * H, device=dev)
lens = torch.from_numpy(np.sort(np.random.RandomState(0).randint(600, Tp + 1, N))[::-1].copy().astype(np.int32)).to(dev)
lens[0] = Tp
dout = torch.randn(Tp, N, H, device=dev).to(torch.bfloat16)
NAMES = ["top", "gather issued", "gather+products done", "barrier passed", "publish issued", "step end"]


def show(name):
    ws = ops.LAST_PERSIST_WS
    tl = ws[-32 * TL_N * TL_K * 8:].view(torch.int64).cpu().numpy().reshape(32, TL_N, TL_K).astype(np.float64)
    print("== %s: per step (rows) the mean / min / max over the 32 workgroups of group 0, cycles relative to the EARLIEST publish of the step before" % name)
    print("   %-8s" % "step" + "".join("%-28s" % n for n in NAMES))
    for i in range(1, TL_N):
        base = tl[:, i - 1, 4].min()
        cells = []
        for k in range(TL_K):
            v = tl[:, i, k] - base
            cells.append("%6.0f [%5.0f..%5.0f]" % (v.mean(), v.min(), v.max()))
        print("   %-8d" % i + "".join("%-28s" % c for c in cells))
    spread = tl[:, 1:, 4].max(0) - tl[:, 1:, 4].min(0)
    period = np.diff(tl[:, :, 4].mean(0))
    print("   publish spread over the group (max - min), per step: %s" % " ".join("%.0f" % v for v in spread))
    print("   step period (mean publish to mean publish): %s" % " ".join("%.0f" % v for v in period))
    seg = np.diff(tl[:, 1:, :], axis=2).mean((0, 1))
    print("   mean segment lengths: " + ", ".join("%s -> %s %.0f" % (NAMES[k], NAMES[k + 1], seg[k]) for k in range(TL_K - 1)))
    wrap = (tl[:, 1:, 0] - tl[:, :-1, 5]).mean()
    print("   step end -> next top %.0f; own publish -> own next gather issued %.0f; LAST peer publish -> own gather done: mean %.0f" % (
        wrap, (tl[:, 1:, 1] - tl[:, :-1, 4]).mean(), (tl[:, 1:, 2] - tl[:, :-1, 4].max(0)[None, :]).mean()))


for it in range(3):
    hext, Sv, hn, cn = ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
    if it == 2:
        torch.cuda.synchronize()
        show("forward")
    ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp)
torch.cuda.synchronize()
show("BPTT")
ops.check_persistent_kernels()
