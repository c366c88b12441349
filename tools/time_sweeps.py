#!/usr/bin/env python
"""Wall time per time step of the persistent recurrent sweeps (shipping build) for a list of shapes -- kernel level, synthetic operands.

    gpurun -- 'python tools/time_sweeps.py lstm,1,88,1024,401 lstm,1,66,1280,401 > gpurun_out/time_sweeps.txt'

Shape = kind,D,N,H,T'[,ragged]: `ragged` = lengths uniform in [T'/3, T'] sorted descending (the loader's order), else every clip T' long.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepspeech.pytorch_amd import _lib, ops  # noqa: E402

if os.environ.get("DS2_AB_LIB"):          # tools/ab_sweeps.py: another build of the library (A/B runs only)
    _lib.LIB_PATH = os.environ["DS2_AB_LIB"]

dev = "cuda"
shapes = sys.argv[1:] or ["lstm,1,88,1024,401", "lstm,1,66,1280,401", "lstm,2,128,1024,401", "lstm,2,64,1280,401", "lstm,2,64,1280,401,ragged"]
for spec in shapes:
    if spec.startswith("variant="):        # ds2_persist_opts.variant bits for the shapes that follow (this tool's process only)
        ops._OPTS["variant"] = int(spec.split("=")[1])
        print("-- variant bits", spec.split("=")[1], flush=True)
        continue
    f = spec.split(",")
    kind, D, N, H, Tp = f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4])
    ragged = len(f) > 5
    G = ops.GATES[kind]
    torch.manual_seed(0)
    GI = torch.randn(Tp * N, D * G * H, device=dev).to(torch.bfloat16)
    Whh = ((torch.rand(D, G * H, H, device=dev) * 2 - 1) / H ** 0.5).to(torch.bfloat16)
    WhhT = Whh.transpose(1, 2).contiguous()
    bhh = torch.zeros(D, G * H, device=dev)
    if ragged:
        lens_np = np.sort(np.random.RandomState(0).randint(Tp // 3, Tp + 1, N))[::-1].copy().astype(np.int32)
        lens_np[0] = Tp
    else:
        lens_np = np.full(N, Tp, dtype=np.int32)
    lens = torch.from_numpy(lens_np).to(dev)
    dout = torch.randn(Tp, N, H, device=dev).to(torch.bfloat16)
    fam = ops.persist_kind(torch.bfloat16, kind, D, N, H)
    best = {"fwd": 1e9, "bwd": 1e9}
    for it in range(4):
        for which in ("fwd", "bwd"):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if which == "fwd":
                hext, Sv, hn, cn = ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
            else:
                ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp)
            e1.record()
            torch.cuda.synchronize()
            best[which] = min(best[which], e0.elapsed_time(e1))
    ops.check_persistent_kernels()
    print("%-28s family %d  valid frames %.2f  fwd %.3f ms = %.3f us/step   bwd %.3f ms = %.3f us/step" % (
        spec, fam, lens_np.sum() / (N * Tp), best["fwd"], best["fwd"] * 1e3 / Tp, best["bwd"], best["bwd"] * 1e3 / Tp), flush=True)
