#!/usr/bin/env python
"""Times one forward and one BPTT persistent sweep at the cfg3 shape and prints the in-kernel cycle split
(gather+MFMA / partial-store+barrier / gate+publish) of workgroup 0 of every group plus the number of poll retries."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepspeech.pytorch_amd import _lib, build  # noqa: E402

_lib.LIB_PATH = build.build(probe=True, verbose=False)      # the instrumented (-DDS2_PROBE) library, not the shipping one
from deepspeech.pytorch_amd import ops  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "gru"
D, N, H, Tp = 2, int(os.environ.get("N", 32)), 1024, int(os.environ.get("TP", 751))
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


def dbg(name):
    ws = ops.LAST_PERSIST_WS
    tail = ws[:1024].view(torch.int64).cpu().numpy().reshape(-1, 8)[:8]
    for g in (0, 7):
        for o, who in ((0, "wave0"), (4, "wave3")):
            c = tail[g][o:o + 4]
            tot = c[0] + c[1] + c[2]
            print("  %s group %d %s: gather+mma %.0f  store+barrier %.0f  gate+publish %.0f  cycles/step (total %.0f) poll retries/step %.2f" % (
                name, g, who, c[0] / Tp, c[1] / Tp, c[2] / Tp, tot / Tp, c[3] / Tp))


masks = [int(x) for x in os.environ.get("MASKS", "0").split(",")]
for mask in masks:
  os.environ["DS2_PERSIST_DBG"] = str(mask)
  print("=== dbgmask %d (1 no prefetch loads, 2 no output stores, 4 no MFMA, 8 no gather)" % mask)
  for it in range(3):
      e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
      e[0].record()
      hext, Sv, hn, cn = ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
      e[1].record()
      if it == 2:
          torch.cuda.synchronize()
          dbg("fwd")
      rg = ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp)
      e[2].record()
      torch.cuda.synchronize()
      print("iter %d: fwd %.3f ms (%.2f us/step)  bwd %.3f ms (%.2f us/step)" % (
          it, e[0].elapsed_time(e[1]), e[0].elapsed_time(e[1]) * 1e3 / Tp, e[1].elapsed_time(e[2]), e[1].elapsed_time(e[2]) * 1e3 / Tp))
  dbg("bwd")
ops.check_persistent_kernels()
