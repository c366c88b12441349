#!/usr/bin/env python
"""Round 6 experiment: a 128 x 128 wave tile (four waves, one per SIMD, 256 accumulator registers) against ds2_gemm8's 128 x 64 (eight
waves) on the NT shapes of the training step -- tools/gemm9_probe.hip, compiled here, never part of libds2hip.so.  Checks the result
against torch (fp32 reference on the same bf16 operands), then times it next to ds2_gemm8_nt and the vendor library (torch.matmul).

    gpurun -- 'python tools/probe_gemm9.py > gpurun_out/gemm9.txt'"""
import ctypes
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepspeech.pytorch_amd import ops  # noqa: E402

d = tempfile.mkdtemp()
lib = os.path.join(d, "libg9.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tools", "gemm9_probe.hip"),
                       "-o", lib] + os.environ.get("G9_FLAGS", "").split())
L = ctypes.CDLL(lib)
L.gemm9_nt.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_long] * 3 + [ctypes.c_void_p]
L.gemm9r_nt.argtypes = L.gemm9_nt.argtypes
dev = "cuda"


def g9(A, B, M, N, K, lda, ldb, bias=None, fn=None):
    C = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    rc = (fn or L.gemm9_nt)(A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr() if bias is not None else None, M, N, K, lda, ldb, N,
                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return C


def timeit(fn, n=12):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts[2:])[len(ts[2:]) // 2]


# correctness
for M, N, K in ((300, 520, 256), (1000, 256, 1024), (24032, 1024, 1024)):
    torch.manual_seed(M)
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    ref = A.float() @ B.float().t() + bias
    for name, fn in (("gemm9 (LDS-DMA)", L.gemm9_nt), ("gemm9r (register staging)", L.gemm9r_nt)):
        C = g9(A, B, M, N, K, K, K, bias, fn=fn)
        err = (C.float() - ref).abs().max().item() / ref.abs().max().item()
        print("check %s M=%d N=%d K=%d: max rel err %.2e %s" % (name, M, N, K, err, "ok" if err < 8e-3 else "WRONG"), flush=True)

shapes = [("i2h cfg3 K=1024 ld 1088", 24032, 6144, 1024, 1088), ("i2h cfg5a K=1280 ld 1344", 48064, 10240, 1280, 1344), ("dX cfg3 K=6144", 24032, 1024, 6144, 6144),
          ("square 8192", 8192, 8192, 8192, 8192), ("square 4096", 4096, 4096, 4096, 4096)]
for name, M, N, K, ld in shapes:
    A = torch.randn(M, ld, device=dev).to(torch.bfloat16)
    B = torch.randn(N, ld, device=dev).to(torch.bfloat16)
    fl = 2.0 * M * N * K
    t9 = timeit(lambda: g9(A, B, M, N, K, ld, ld))
    t9r = timeit(lambda: g9(A, B, M, N, K, ld, ld, fn=L.gemm9r_nt))
    t8 = timeit(lambda: ops.gemm8_nt(A, B, M=M, N=N, K=K, lda=ld, ldb=ld))
    Av, Bv = A[:, :K], B[:, :K]
    tv = timeit(lambda: torch.matmul(Av, Bv.t()))
    c9, c8 = g9(A, B, M, N, K, ld, ld, fn=L.gemm9r_nt), ops.gemm8_nt(A, B, M=M, N=N, K=K, lda=ld, ldb=ld)
    same = (c9.float() - c8.float()).abs().max().item()
    print("%-28s gemm9 %.3f ms %5.0f TFLOP/s | gemm9r %.3f ms %5.0f | gemm8 %.3f ms %5.0f | vendor library %.3f ms %5.0f | max |gemm9r - gemm8| %.3g" % (
        name, t9, fl / t9 / 1e9, t9r, fl / t9r / 1e9, t8, fl / t8 / 1e9, tv, fl / tv / 1e9, same), flush=True)
