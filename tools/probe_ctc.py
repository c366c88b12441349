"""Times the fused log_softmax + CTC loss/gradient kernel at the cfg3 shape (32 clips, T' up to 751, bench targets)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepspeech.pytorch_amd import _lib
if os.environ.get("DS2_LIB"):                   # A/B: another build of the library (tools/ab_variants.py --build-only ...)
    _lib.LIB_PATH = os.environ["DS2_LIB"]
from deepspeech.pytorch_amd import ops, synth

if os.environ.get("DS2_CTC_RECURSION"):         # A/B: 1 = four-wave recursion kernel, 2 = one-wave kernel wherever it applies
    ops.CTC_RECURSION = int(os.environ["DS2_CTC_RECURSION"])
lengths = synth.synth_lengths(32, 1201, 1501, seed=3000)
inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=3000)
N = 32
Tp = (int(inputs.shape[3]) - 1) // 2 + 1
out = torch.tensor([(int(l) - 1) // 2 + 1 for l in lengths], dtype=torch.int32, device="cuda")
tsz_t = torch.from_numpy(tsz).to(torch.int64)
offs = torch.zeros(N, dtype=torch.int64)
offs[1:] = torch.cumsum(tsz_t, 0)[:-1]
g = torch.Generator().manual_seed(0)
logits = torch.randn((Tp * N, 32), generator=g).cuda()
tg = torch.from_numpy(targets).to("cuda", torch.int32)


def run(name, out_lens, tlens):
    tl = tlens.to(torch.int64)
    o = torch.zeros(N, dtype=torch.int64)
    o[1:] = torch.cumsum(tsz_t, 0)[:-1]           # offsets of the full targets; shorter targets use a prefix
    args = (logits, tg, o.to("cuda", torch.int32), out_lens, tl.to("cuda", torch.int32), Tp, N, 29, 0, int(tl.max()))
    for _ in range(3):
        loss, nll, dl = ops.ctc_loss_grad(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        loss, nll, dl = ops.ctc_loss_grad(*args)
    e1.record()
    torch.cuda.synchronize()
    print("%-28s T' max %4d, target max %3d: %7.1f us per call, loss %.4f"
          % (name, int(out_lens.max()), int(tl.max()), e0.elapsed_time(e1) / 20 * 1e3, float(loss)))


run("cfg3 batch", out, tsz_t)
run("half the frames", (out // 2).to(torch.int32), tsz_t)
run("quarter of the labels", out, tsz_t // 4)
run("half frames, quarter labels", (out // 2).to(torch.int32), tsz_t // 4)
