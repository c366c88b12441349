"""Times the bf16 conv1 kernels (forward, weight gradient) at the cfg3 and cfg5 shapes.
    gpurun -- 'python tools/probe_conv1.py > gpurun_out/conv1.txt'        env DS2_LIB: another build of the library"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepspeech.pytorch_amd import _lib
if os.environ.get("DS2_LIB"):
    _lib.LIB_PATH = os.environ["DS2_LIB"]
from deepspeech.pytorch_amd import ops


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


for name, N, T in (("cfg3 (32 clips, 1501 frames)", 32, 1501), ("cfg5 (64 clips, 1501 frames)", 64, 1501)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, 1, 161, T), generator=g).cuda()
    w1k = (torch.rand((451, 32), generator=g) * 0.1 - 0.05).cuda()
    b = torch.zeros(32).cuda()
    Tp = (T - 1) // 2 + 1
    lens = torch.full((N,), Tp, dtype=torch.int32).cuda()
    y = ops.conv1_fwd(x, w1k, b, lens, Tp, torch.bfloat16)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2)).cuda().to(torch.bfloat16)
    fl = 2.0 * N * 81 * Tp * 32 * 451
    tf = t_us(lambda: ops.conv1_fwd(x, w1k, b, lens, Tp, torch.bfloat16))
    tw = t_us(lambda: ops.conv1_wgrad(x, dy, Tp))
    print("%-30s forward %7.1f us (%4.0f TFLOP/s)   weight gradient %7.1f us (%4.0f TFLOP/s)" % (name, tf, fl / tf / 1e6, tw, fl / tw / 1e6))
