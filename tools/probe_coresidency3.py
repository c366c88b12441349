#!/usr/bin/env python
"""Third co-residency probe: WHICH phase of a BPTT / forward sweep step gets longer next to the co-resident weight-gradient GEMM,
in shader CYCLES (the -DDS2_PROBE build's s_memtime counters) next to wall time -- cycles that stay put while the wall time grows
mean a lower clock (DVFS), cycles that grow mean contention, and the phase says for what.

    gpurun -- 'python tools/probe_coresidency3.py > gpurun_out/coresidency3.txt'
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepspeech.pytorch_amd import _lib, build  # noqa: E402

_lib.LIB_PATH = build.build(probe=True, verbose=False)
from deepspeech.pytorch_amd import ops  # noqa: E402

kind, D, N, H, Tp = "gru", 2, 32, 1024, 751
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
Kp = (Tp * N + 63) // 64 * 64
At = torch.randn(D * G * H, Kp, device=dev).to(torch.bfloat16)
Bt = torch.randn(H, Kp, device=dev).to(torch.bfloat16)
Bsmall = torch.randn(256, Kp, device=dev).to(torch.bfloat16)
side = torch.cuda.Stream()
os.environ["DS2_PERSIST_DBG"] = "0"


def counters():
    ws = ops.LAST_PERSIST_WS
    tail = ws[:1024].view(torch.int64).cpu().numpy().reshape(-1, 8)[:8]
    out = []
    for g in (0, 3, 7):
        c = tail[g][0:4]
        out.append("g%d w0: gather+mma %.0f store+barrier %.0f gate+publish %.0f = %.0f cyc/step, %.2f re-polls" % (
            g, c[0] / Tp, c[1] / Tp, c[2] / Tp, (c[0] + c[1] + c[2]) / Tp, c[3] / Tp))
    return "\n      ".join(out)


def interferers():
    yield "nothing", None
    yield "4 x co-resident low-register GEMM [6144 x 1024 x T'N]", lambda: [ops.gemm_nt(At, Bt, out_dtype=torch.float32, coresident=True) for _ in range(4)]
    yield "16 x co-resident GEMM with an L2-resident B panel [6144 x 256 x T'N]", lambda: [ops.gemm_nt(At, Bsmall, out_dtype=torch.float32, coresident=True) for _ in range(16)]


hext, Sv, hn, cn = ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
for name, run in interferers():
    for which in ("bwd", "fwd"):
        best = None
        for it in range(3):
            torch.cuda.synchronize()
            if run is not None:
                with torch.cuda.stream(side):
                    run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if which == "bwd":
                ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp)
            else:
                ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            if best is None or t < best[0]:
                best = (t, counters())
        cyc = float(best[1].split("= ")[1].split(" cyc")[0])
        print("%s next to %s: %.3f ms = %.2f us per time step -> %.2f GHz implied by group 0's cycle count\n      %s" % (
            which, name, best[0], best[0] * 1e3 / Tp, cyc / (best[0] * 1e3 / Tp) / 1e3, best[1]))
ops.check_persistent_kernels()
