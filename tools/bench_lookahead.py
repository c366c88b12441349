#!/usr/bin/env python
"""Lookahead forward / backward at the config-5b shape (T' = 751, N = 64, H = 1280, bf16), context 20 (sliding-window kernels) and
21 (the per-tap kernels), microseconds per call.   python tools/bench_lookahead.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepspeech.pytorch_amd import ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    Tp, N, H = 751, 64, 1280
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(Tp * N, H, device="cuda", generator=g).bfloat16()
    dy = torch.randn(Tp * N, H, device="cuda", generator=g).bfloat16()
    for ctx in (20, 21):
        w = torch.rand(H, ctx, device="cuda", generator=g) - 0.4
        y, pre = ops.lookahead_fwd(x, w, Tp, N, H)
        print("ctx %d: fwd %.0f us, bwd (dx + dw + column sum) %.0f us" % (
            ctx, timed(lambda: ops.lookahead_fwd(x, w, Tp, N, H)), timed(lambda: ops.lookahead_bwd(x, w, pre, dy, Tp, N, H))))


if __name__ == "__main__":
    main()
