#!/usr/bin/env python
"""Times ds2_gemm_nt on the GEMM shapes of the cfg3 training step (HIP events, median of 10)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepspeech.pytorch_amd import ops  # noqa: E402

dev = "cuda"
shapes = [  # name, M, N, K, out f32, splitk
    ("i2h fwd", 24032, 6144, 1024, False, 1),
    ("i2h fwd l0", 24032, 6144, 1344, False, 1),
    ("dgrad", 24032, 1024, 6144, False, 1),
    ("wgrad ih", 6144, 1024, 24064, True, 1),
    ("wgrad ih sk2", 6144, 1024, 24064, True, 2),
    ("wgrad hh", 3072, 1024, 24064, True, 1),
    ("wgrad hh sk4", 3072, 1024, 24064, True, 4),
    ("head", 24032, 32, 1024, True, 1),
]
for name, M, N, K, f32, sk in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    ts = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        C = ops.gemm_nt(A, B, out_dtype=torch.float32 if f32 else None, splitk=sk)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts[2:])[len(ts[2:]) // 2]
    print("%-14s M=%5d N=%5d K=%5d  %.3f ms  %.0f TFLOP/s" % (name, M, N, K, t, 2.0 * M * N * K / t / 1e9))
    rows = torch.cat([torch.arange(0, 130), torch.arange(M - 130, M)]).to(dev)
    ref = (A[rows].float() @ B.float().t())
    err = (C[rows].float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (1e-5 if f32 else 6e-3), (name, err)
