#!/usr/bin/env python
"""cfg2 (fp32 mode) at full size: per-element distance of the small gradient tensors (stored whole in the fixture since round 5) from the
reference's, in units of the tensor's largest element -- which elements sit near the 1e-3 bar, and do they move with the Hardtanh
decisions (DS2 fp32 BatchNorm output within rounding of a clamp boundary)?

    gpurun -- 'python tools/diag_cfg2_small_tensors.py > gpurun_out/diag_cfg2_small.txt'"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepspeech.pytorch_amd import configs, ops, synth  # noqa: E402
from deepspeech.pytorch_amd.model import DeepSpeech  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "full", "cfg2.npz"))
meta = json.loads(bytes(z["meta_json"]).decode())
lengths = np.asarray(meta["lengths"], dtype=np.int64)
inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=meta["data_seed"])
P = synth.synth_params({k: tuple(v) for k, v in meta["shapes"].items()}, meta["param_seed"])
mc = configs.BiDirectionalConfig(rnn_type=configs.RNNType.gru, hidden_size=meta["hidden_size"], hidden_layers=meta["hidden_layers"])
# capture the inputs of the two conv-block BatchNorms (X = conv output, NFTC) and their saved scale / shift
captured = []
_bn_fwd = ops.bn_fwd


def bn_fwd_spy(X, mode, *a, **k):
    sv = _bn_fwd(X, mode, *a, **k)
    if mode in (1, 2):
        captured.append((mode, X, sv, k.get("lens")))
    return sv


ops.bn_fwd = bn_fwd_spy
for gemm in (os.environ.get("DS2_FP32_GEMM", "wgrad"),):
    m = DeepSpeech(configs.LABELS, mc, 32, configs.AdamConfig(), configs.SpectConfig())
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    m = m.to("cuda").train()
    loss = m.training_step((torch.from_numpy(inputs).cuda(), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz)), 0)
    loss.backward()
    ops.check_persistent_kernels()
    print("DS2_FP32_GEMM=%s loss %.6f (reference %.6f)" % (gemm, float(loss.item()), float(z["loss"])))
    rows = []
    for k, p in m.named_parameters():
        if "gradfull." + k not in z.files:
            continue
        g = p.grad.detach().double().cpu().numpy().reshape(-1)
        r = z["gradfull." + k].astype(np.float64)
        e = np.abs(g - r) / max(np.abs(r).max(), 1e-30)
        rows.append((e.max(), k, int(e.argmax()), g.size, float(np.sqrt(((g - r) ** 2).sum() / max((r ** 2).sum(), 1e-60)))))
    for emax, k, i, n, l2 in sorted(rows, reverse=True)[:12]:
        print("%-40s worst element %.3e of max (index %d of %d), relative L2 %.3e" % (k, emax, i, n, l2))

# Hardtanh decisions of the conv block within fp32 rounding of a clamp boundary: y = x * scale + shift (what the kernels compute in fp32),
# evaluated in float64 from the same stored x; a |y| or |y - 20| below ~1e-6 * |x * scale| means the sign of the decision is rounding
for mode, X, sv, lens in captured:
    x = X.detach().double().cpu().numpy()                      # [N][F][Tp][32]
    sc, sh = sv.scale.double().cpu().numpy(), sv.shift.double().cpu().numpy()
    ln = lens.cpu().numpy()
    y = x * sc + sh
    mag = np.abs(x * sc) + np.abs(sh) + 1e-30
    valid = (np.arange(x.shape[2])[None, None, :, None] < ln[:, None, None, None])
    for name, d in (("0", np.abs(y)), ("20", np.abs(y - 20.0))):
        rel = np.where(valid, d / mag, np.inf)
        idx = np.unravel_index(np.argsort(rel, axis=None)[:4], rel.shape)
        print("BatchNorm of conv block %d, closest Hardtanh inputs to the boundary %s: %s" % (
            mode, name, ", ".join("channel %d: |y - b| = %.2e (%.1e of the operands' magnitude)" % (idx[3][i], d[idx[0][i], idx[1][i], idx[2][i], idx[3][i]],
                                                                                                 rel[idx[0][i], idx[1][i], idx[2][i], idx[3][i]]) for i in range(4))))
