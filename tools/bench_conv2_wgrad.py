#!/usr/bin/env python
"""Times ds2_conv2_wgrad, ds2_conv2_fwd and ds2_conv2_dgrad (bf16, 161 bins) on the cfg3 / cfg5a shapes and, with DS2_AB_LIB=<other build>, runs that build in a child
process on the same seeded operands and compares the results BIT FOR BIT (the pipelined kernel keeps the summation order).

    python tools/bench_conv2_wgrad.py                 # this build
    DS2_AB_LIB=deepspeech/pytorch_amd/libds2hip_base.so python tools/bench_conv2_wgrad.py --dump /tmp/base.pt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepspeech.pytorch_amd import _lib  # noqa: E402

if os.environ.get("DS2_AB_LIB"):
    _lib.LIB_PATH = os.environ["DS2_AB_LIB"]
from deepspeech.pytorch_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dump")
ap.add_argument("--compare")
a = ap.parse_args()
dev = "cuda"
out = {}
for name, N, Tp in (("cfg3", 32, 751), ("cfg5a", 64, 751), ("ragged", 3, 77)):
    g = torch.Generator(device="cpu").manual_seed(1234 + N)
    dy = (torch.randn(N, 41, Tp, 32, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    a1 = torch.randn(N, 81, Tp, 32, generator=g).clamp_(0, 20).to(torch.bfloat16).to(dev)
    ts = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dw = ops.conv2_wgrad(dy, a1, 161)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts[2:])[len(ts[2:]) // 2]
    fl = 2.0 * 32 * 32 * 231 * N * 41 * Tp
    print("%-6s N=%2d T'=%4d: %.3f ms (kernel + reduction)  %.0f TFLOP/s  [%s]" % (name, N, Tp, t, fl / t / 1e9, os.path.basename(_lib.LIB_PATH)))
    out[name] = dw.cpu()
    w = (torch.randn(32, 32, 21, 11, generator=g) * 0.02)
    w2t = w.permute(2, 3, 0, 1).contiguous().to(torch.bfloat16).to(dev)
    w2d = [w[:, :, q::2, :].flip(2, 3).permute(2, 3, 1, 0).contiguous().to(torch.bfloat16).to(dev) for q in (0, 1)]
    b2 = (torch.randn(32, generator=g) * 0.1).to(dev)
    lens = torch.tensor([max(Tp - 7 * i, 1) for i in range(N)], dtype=torch.int32, device=dev)
    for what, fn in (("fwd", lambda: ops.conv2_fwd(a1, w2t, b2, lens, 161)), ("dgrad", lambda: ops.conv2_dgrad(dy, w2d[0], w2d[1], 161))):
        ts = []
        for it in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts[2:])[len(ts[2:]) // 2]
        print("%-6s conv2 %-5s: %.3f ms  %.0f TFLOP/s" % (name, what, t, fl / t / 1e9))
        out[name + "." + what] = r.float().cpu()
if a.dump:
    torch.save(out, a.dump)
if a.compare:
    ref = torch.load(a.compare)
    for k in out:
        same = torch.equal(out[k], ref[k])
        print("%-6s bit-identical to %s: %s (max abs diff %.3e, max |ref| %.3e)" % (
            k, a.compare, same, (out[k] - ref[k]).abs().max().item(), ref[k].abs().max().item()))
