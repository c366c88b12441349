#!/usr/bin/env python
"""Race screen of the persistent sweeps at kernel level: twelve forward + BPTT launches per shape on the same operands (every third
with NaN in the padding rows of the input projection, as a row-list product leaves them), each compared with the launch-per-step
kernels and bit for bit with the first launch.  Found the handshake-word overflow of H = 512 (profiles/r04i_*, r04j_*).

    gpurun -- 'DS2_ALLOW_LAUNCH_PER_STEP=1 python tools/repeat_sweeps.py [gru1 lstm2 ...] > gpurun_out/repeat_sweeps.txt'
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepspeech.pytorch_amd import ops, _lib
dev = "cuda"
CASES = [("lstm", 1, 16, 512, 451), ("gru", 1, 16, 512, 451), ("lstm", 2, 16, 512, 451), ("gru", 2, 16, 512, 451), ("lstm", 1, 64, 1280, 451)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if "%s%d" % (c[0], c[1]) in sys.argv[1:]]
for kind, D, N, H, Tp in CASES:
    G = ops.GATES[kind]
    torch.manual_seed(0)
    GI = torch.randn(Tp * N, D * G * H, device=dev).to(torch.bfloat16)
    Whh = ((torch.rand(D, G * H, H, device=dev) * 2 - 1) / H ** 0.5).to(torch.bfloat16)
    WhhT = Whh.transpose(1, 2).contiguous()
    bhh = torch.zeros(D, G * H, device=dev)
    lens_np = np.sort(np.random.RandomState(0).randint(Tp // 3, Tp + 1, N))[::-1].copy().astype(np.int32)
    lens_np[0] = Tp
    lens = torch.from_numpy(lens_np).to(dev)
    dout = torch.randn(Tp, N, H, device=dev).to(torch.bfloat16)
    fam = ops.persist_kind(torch.bfloat16, kind, D, N, H)
    print("====", kind, D, N, H, Tp, "family", fam, flush=True)
    ops.PERSIST_ENABLED = False
    hext_r, Sv_r, hn_r, cn_r = ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
    rg_r = ops.rnn_bwd(kind, dout, WhhT, hext_r, Sv_r, lens, D, N, H, Tp)
    ops.PERSIST_ENABLED = True
    hr = hext_r.float().clone(); dr = rg_r.dGI.float().clone()
    first = None
    for it in range(12):
        if it % 3 == 2:      # garbage in the padding rows of GI (what a row-list product leaves)
            GI2 = GI.clone()
            mask = (torch.arange(Tp, device=dev)[:, None] >= lens[None, :]).reshape(-1)
            GI2[mask] = float("nan")
        else:
            GI2 = GI
        hext, Sv, hn, cn = ops.rnn_fwd(kind, GI2, Whh, bhh, lens, D, N, H, Tp)
        rg = ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp)
        torch.cuda.synchronize()
        ops.check_persistent_kernels()
        h = hext.float(); d = rg.dGI.float()
        eh = (h - hr).abs(); ed = (d - dr).abs()
        msg = "it %2d: fwd vs launch-per-step max %.4f (%d elems > 0.05)  bwd max %.4f (%d > 0.05 x max %.3f)" % (
            it, float(eh.max()), int((eh > 0.05).sum()), float(ed.max()), int((ed > 0.05 * dr.abs().max()).sum()), float(dr.abs().max()))
        if first is None:
            first = (hext.clone(), rg.dGI.clone())
        else:
            nh = int((hext != first[0]).sum()); nd = int((rg.dGI != first[1]).sum())
            msg += "   vs run 0: %d h elems differ, %d dGI elems differ" % (nh, nd)
            if nh:
                idx = (hext != first[0]).nonzero()
                msg += "  first at (d,t+1,n,j)=%s last %s" % (idx[0].tolist(), idx[-1].tolist())
        print(msg, flush=True)
