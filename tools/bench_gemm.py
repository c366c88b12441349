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
    if not f32:   # yard-stick only (never on the product path): the vendor library (hipBLASLt behind torch.matmul) on the same operands
        ts = []
        for it in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            Cl = torch.matmul(A, B.t())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts[2:])[len(ts[2:]) // 2]
        print("   vendor library (torch.matmul, bf16):   %.3f ms  %.0f TFLOP/s" % (t, 2.0 * M * N * K / t / 1e9))
    if K % 64 == 0 and N >= 256:
        ts = []
        for it in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            C8 = ops.gemm8_nt(A, B, out_dtype=torch.float32 if f32 else None)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts[2:])[len(ts[2:]) // 2]
        err = (C8[rows].float() - ref).abs().max().item() / ref.abs().max().item()
        print("   gemm8 256x256 phase-split:            %.3f ms  %.0f TFLOP/s  (err %.1e)" % (t, 2.0 * M * N * K / t / 1e9, err))
# grouped TN: one layer's weight gradients of cfg3 (dW_ih + 2 x dW_hh with the [dr, dz | dQ] split) in one launch
R, H = 24064, 1024
dGI = torch.randn(R, 6 * H, device=dev).to(torch.bfloat16)
X = torch.randn(R, H, device=dev).to(torch.bfloat16)
dQ = torch.randn(2, R, H, device=dev).to(torch.bfloat16)
Hp = torch.randn(2, R, H, device=dev).to(torch.bfloat16)
probs = [dict(At=dGI, Bt=X, M=6 * H, N=H, lda=6 * H, ldb=H)]
for d in range(2):
    probs.append(dict(At=dGI[:, d * 3 * H:], At2=dQ[d], lda2=H, m_split=2 * H, Bt=Hp[d], M=3 * H, N=H, lda=6 * H, ldb=H))
ts = []
for it in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    outs = ops.gemm8_tn_grouped(probs, R)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = sorted(ts[2:])[len(ts[2:]) // 2]
fl = 2.0 * R * (6 * H * H + 2 * 3 * H * H)
ref = dGI[:, :256].float().t() @ X.float()
print("grouped TN weight gradients of one cfg3 layer (192 tiles, K = %d): %.3f ms  %.0f TFLOP/s  (err %.1e)" % (
    R, t, fl / t / 1e9, (outs[0][:256] - ref).abs().max().item() / ref.abs().max().item()))
# leading-dimension sensitivity of the K = 1024 input projection (row stride 2048 B = a power of two: channel camping?)
for ld in (1024, 1032, 1088, 1152):
    M, N, K = 24032, 6144, 1024
    Aw = torch.randn(M, ld, device=dev).to(torch.bfloat16)
    Bw = torch.randn(N, ld, device=dev).to(torch.bfloat16)
    ts = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        C8 = ops.gemm8_nt(Aw, Bw, M=M, N=N, K=K, lda=ld, ldb=ld)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts[2:])[len(ts[2:]) // 2]
    print("i2h K=1024 with lda = ldb = %d: %.3f ms  %.0f TFLOP/s" % (ld, t, 2.0 * M * N * K / t / 1e9))
for Kx in (1024, 2048, 4096):
    M, N = 24032, 6144
    Aw = torch.randn(M, Kx + 64, device=dev).to(torch.bfloat16)
    Bw = torch.randn(N, Kx + 64, device=dev).to(torch.bfloat16)
    ts = []
    for it in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        C8 = ops.gemm8_nt(Aw, Bw, M=M, N=N, K=Kx, lda=Kx + 64, ldb=Kx + 64)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts[2:])[len(ts[2:]) // 2]
    print("M=24032 N=6144 K=%d (ld K+64): %.3f ms  %.0f TFLOP/s" % (Kx, t, 2.0 * M * N * Kx / t / 1e9))

# (the staging-schedule A/B of rounds 3-5 -- ds2_gemm8_set_variant -- is gone with the hook: profiles/r03d_gemm8_ab.txt keeps its result)
from deepspeech.pytorch_amd import _lib  # noqa: E402,F401
# yard-stick: the vendor library on the TN weight-gradient product and on config 5a's input projection (bf16 results; the kernels of
# this repository accumulate and store the weight gradients in fp32)
def _time(fn, n=10):
    ts = []
    for it in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts[2:])[len(ts[2:]) // 2]


t = _time(lambda: torch.matmul(dGI.t(), X))
print("vendor library, TN dW_ih of one cfg3 layer (M=6144 N=1024 K=%d): %.3f ms  %.0f TFLOP/s" % (R, t, 2.0 * R * 6 * H * H / t / 1e9))
M5, N5, K5 = 64 * 751, 2 * 4 * 1280, 1280
A5 = torch.randn(M5, K5, device=dev).to(torch.bfloat16)
B5 = torch.randn(N5, K5, device=dev).to(torch.bfloat16)
t = _time(lambda: torch.matmul(A5, B5.t()))
print("cfg5a i2h M=%d N=%d K=%d vendor library: %.3f ms  %.0f TFLOP/s" % (M5, N5, K5, t, 2.0 * M5 * N5 * K5 / t / 1e9))
t = _time(lambda: ops.gemm8_nt(A5, B5))
print("cfg5a i2h M=%d N=%d K=%d gemm8:          %.3f ms  %.0f TFLOP/s" % (M5, N5, K5, t, 2.0 * M5 * N5 * K5 / t / 1e9))
