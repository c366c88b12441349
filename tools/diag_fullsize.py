#!/usr/bin/env python
"""Per-parameter distance of the full-size bf16 step's gradients from the reference's own fp32 CPU gradients
(tests/golden/full/<config>.npz), for this repo's kernels AND for stock PyTorch-ROCm under bf16 autocast on the same device --
the table behind the bars of tests/test_gpu_model.py::test_full_size_step_matches_the_reference_itself.

    gpurun -- 'python tools/diag_fullsize.py cfg3 > gpurun_out/diag_fullsize_cfg3.txt'"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepspeech.pytorch_amd import configs, ops, synth  # noqa: E402
from deepspeech.pytorch_amd.model import DeepSpeech  # noqa: E402
from oracle import ds2_torch_port as TP  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
z = np.load(os.path.join(ROOT, "tests", "golden", "full", name + ".npz"))
meta = json.loads(bytes(z["meta_json"]).decode())
kind, H, L, bi = meta["rnn_type"], meta["hidden_size"], meta["hidden_layers"], meta["bidirectional"]
lengths = np.asarray(meta["lengths"], dtype=np.int64)
inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=meta["data_seed"])
P = synth.synth_params({k: tuple(v) for k, v in meta["shapes"].items()}, meta["param_seed"])
DEV = "cuda"
mk = lambda: (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
rt = getattr(configs.RNNType, kind)
mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L) if bi else \
    configs.UniDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L, lookahead_context=meta["lookahead_context"])
m = DeepSpeech(configs.LABELS, mc, "bf16", configs.AdamConfig(), configs.SpectConfig())
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
m = m.to(DEV).train()
loss = m.training_step(mk(), 0)
loss.backward()
ops.check_persistent_kernels()
own = {k: p.grad.detach().float().cpu().numpy().astype(np.float64).reshape(-1) for k, p in m.named_parameters()}
own_loss = float(loss.item())
del m
torch.cuda.empty_cache()
cfg = dict(rnn_type=kind, hidden_size=H, hidden_layers=L, bidirectional=bi, lookahead_context=meta["lookahead_context"])
state = {k: torch.from_numpy(v.copy()) for k, v in P.items()}
res = {}
for label, ac in (("stock bf16", True), ("stock fp32", False)):
    port = TP.Port(cfg, state, DEV)
    if ac:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ref = port.training_loss(mk())
    else:
        ref = port.training_loss(mk())
    ref.backward()
    res[label] = (float(ref.item()), {k: p.grad.detach().float().cpu().numpy().astype(np.float64).reshape(-1) for k, p in port.P.items() if p.grad is not None})
    del port, ref
    torch.cuda.empty_cache()
print("loss: reference fp32 (CPU) %.4f, reference autocast (CPU) %.4f, ours %.4f, stock bf16 %.4f, stock fp32 (GPU) %.4f" % (
    float(z["loss"]), float(z["loss_ac"]) if "loss_ac" in z.files else float("nan"), own_loss, res["stock bf16"][0], res["stock fp32"][0]))
st = meta["grad_stride"]
print("%-44s %10s %10s %10s %10s" % ("relative L2 distance from the reference fp32", "ours", "stock bf16", "stock fp32", "ref autocast"))
for k in own:
    sub = z["gradsub." + k].astype(np.float64)
    den = max(np.sqrt((sub ** 2).sum()), 1e-30)
    d = lambda g: np.sqrt(((g[::st] - sub) ** 2).sum()) / den
    print("%-44s %10.3e %10.3e %10.3e %10.3e" % (k, d(own[k]), d(res["stock bf16"][1][k]), d(res["stock fp32"][1][k]),
                                                  float(z["acnoise." + k]) if "acnoise." + k in z.files else float("nan")))
