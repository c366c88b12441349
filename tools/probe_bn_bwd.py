"""Times ds2_bn_bwd (reduce + finalize + apply) at the shapes of cfg3 and cfg5a; checks two builds against each other bit for bit.
    gpurun -- 'python tools/probe_bn_bwd.py [other_lib.so] > gpurun_out/bn_bwd.txt'"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepspeech.pytorch_amd import _lib
if os.environ.get("DS2_LIB"):
    _lib.LIB_PATH = os.environ["DS2_LIB"]
from deepspeech.pytorch_amd import ops
from collections import namedtuple

SV = namedtuple("SV", "mean rstd scale shift")


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    g = torch.Generator().manual_seed(5)
    out = {}
    for name, mode, N, F, Tp, C in (("cfg3 BatchNorm1d H=1024", 0, 32, 0, 751, 1024), ("cfg5a BatchNorm1d H=1280", 0, 64, 0, 751, 1280),
                                    ("cfg3 conv BN1 (81 rows)", 1, 32, 81, 751, 32), ("cfg3 conv BN2 -> sequence (41 rows)", 2, 32, 41, 751, 32),
                                    ("cfg5 conv BN1", 1, 64, 81, 751, 32)):
        R = Tp * N if mode == 0 else N * F * Tp
        X = torch.randn((R, C), generator=g).cuda().to(torch.bfloat16)
        if mode == 2:
            G = torch.randn((Tp * N, F * C), generator=g).cuda().to(torch.bfloat16)
            ldg = F * C
        else:
            G = torch.randn((R, C), generator=g).cuda().to(torch.bfloat16)
            ldg = C
        DX = torch.empty((R, C), dtype=torch.bfloat16, device="cuda")
        sv = SV(torch.randn(C, generator=g).cuda() * 0.1, torch.rand(C, generator=g).cuda() + 0.5, torch.rand(C, generator=g).cuda() + 0.5,
                torch.rand(C, generator=g).cuda() * 4)
        lens = (torch.arange(N, dtype=torch.int32) * 3 + Tp - 3 * N).clamp(1, Tp).cuda() if mode else None
        fn = lambda: ops.bn_bwd(G, X, DX, mode, sv, R, C, ldg, C, C, F=F, Tp=Tp, N=N, lens=lens)
        us = t_us(fn)
        dg, db = fn()
        torch.cuda.synchronize()
        byts = 3.0 * R * C * 2 + 2.0 * R * C * 2
        print("%-40s %8.1f us for reduce + finalize + apply (%.2f TB/s over 5 passes of the tensor)" % (name, us, byts / us / 1e6), flush=True)
        out[name] = (DX.clone().cpu(), dg.cpu(), db.cpu())
    return out


if __name__ == "__main__":
    res = main()
    if len(sys.argv) > 1 and not os.environ.get("DS2_LIB"):
        torch.save(res, "/tmp/bn_a.pt")
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DS2_LIB=sys.argv[1]), capture_output=True, text=True)
        print("---- " + sys.argv[1] + "\n" + r.stdout + r.stderr[-500:])
    elif os.environ.get("DS2_LIB") and os.path.exists("/tmp/bn_a.pt"):
        a = torch.load("/tmp/bn_a.pt")
        same = all(torch.equal(x, y) for k in a for x, y in zip(a[k], res[k]))
        print("outputs of the two builds bit-identical:", same)
