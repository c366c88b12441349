"""Times the Lookahead kernels (forward, data + weight gradient) at cfg5b's shape (64 clips, 751 frames, H = 1280, context 20).
    gpurun -- 'python tools/probe_lookahead.py'        env DS2_LIB: another build of the library"""
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


Tp, N, H = 751, 64, 1280
g = torch.Generator().manual_seed(3)
x = (torch.randn((Tp * N, H), generator=g) * 3).cuda().to(torch.bfloat16)
w = (torch.rand((H, 20), generator=g) * 1.5 - 0.6).cuda()
dy = torch.randn((Tp * N, H), generator=g).cuda().to(torch.bfloat16)
y, pre = ops.lookahead_fwd(x, w, Tp, N, H)
dx, dw = ops.lookahead_bwd(x, w, pre, dy, Tp, N, H)
mb = Tp * N * H * 2 / 1e6
tf = t_us(lambda: ops.lookahead_fwd(x, w, Tp, N, H))
tb = t_us(lambda: ops.lookahead_bwd(x, w, pre, dy, Tp, N, H))
print("forward %.1f us (%.2f TB/s over x + y + pre), backward (dx + dw + column sums) %.1f us (%.2f TB/s over dy, pre, x, dy, pre, dx)" % (
    tf, 3 * mb / tf, tb, 6 * mb / tb))
print("checksums: y %.6e pre %.6e dx %.6e dw %.6e" % (y.float().sum().item(), pre.float().sum().item(), dx.float().sum().item(), dw.sum().item()))
