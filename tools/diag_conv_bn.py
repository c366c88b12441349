#!/usr/bin/env python
"""Is the conv-block BatchNorm gradient of this repo's bf16 step (conv.seq_module.1.weight: 0.345 from the reference's fp32 gradient,
stock PyTorch-ROCm bf16 0.262, the reference under CPU autocast 0.084 -- profiles/r03c_diag_fullsize_cfg3.txt) systematically worse
than stock's, or is that one draw of a sum of ~2e6 cancelling terms per channel?  For several synthetic batches of cfg3's shape the
script takes stock PyTorch-ROCm in fp32 ON THIS DEVICE as the truth (it sits 1.8e-2 from the reference's CPU fp32 on that tensor) and
prints the relative L2 distance of ours (bf16) and of stock bf16 for the conv-block tensors.

    gpurun -- 'python tools/diag_conv_bn.py cfg3 11 12 13 14 > gpurun_out/diag_conv_bn.txt'"""
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
seeds = [int(a) for a in sys.argv[2:]] or [11, 12, 13, 14]
z = np.load(os.path.join(ROOT, "tests", "golden", "full", name + ".npz"))
meta = json.loads(bytes(z["meta_json"]).decode())
kind, H, L, bi = meta["rnn_type"], meta["hidden_size"], meta["hidden_layers"], meta["bidirectional"]
P = synth.synth_params({k: tuple(v) for k, v in meta["shapes"].items()}, meta["param_seed"])
DEV = "cuda"
rt = getattr(configs.RNNType, kind)
mc = configs.BiDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L) if bi else \
    configs.UniDirectionalConfig(rnn_type=rt, hidden_size=H, hidden_layers=L, lookahead_context=meta["lookahead_context"])
cfg = dict(rnn_type=kind, hidden_size=H, hidden_layers=L, bidirectional=bi, lookahead_context=meta["lookahead_context"])
KEYS = ["conv.seq_module.0.weight", "conv.seq_module.1.weight", "conv.seq_module.1.bias", "conv.seq_module.3.weight",
        "conv.seq_module.4.weight", "conv.seq_module.4.bias", "rnns.0.rnn.weight_ih_l0", "rnns.1.batch_norm.module.weight"]
print("relative L2 distance from stock PyTorch-ROCm fp32 on this device: ours bf16 | stock bf16   (variant env: %s)" % {
    k: v for k, v in os.environ.items() if k.startswith("DS2_")})
print("%-8s %s" % ("seed", " ".join("%-27s" % k.replace("conv.seq_module.", "conv.").replace(".module", "")[:27] for k in KEYS)))
tot = {k: [0.0, 0.0] for k in KEYS}
for seed in seeds:
    lengths = np.asarray(meta["lengths"], dtype=np.int64) if seed == meta["data_seed"] else \
        synth.synth_lengths(len(meta["lengths"]), int(min(meta["lengths"])), int(max(meta["lengths"])), seed=seed)
    inputs, targets, pct, tsz = synth.synth_batch(lengths, seed=seed)
    mk = lambda: (torch.from_numpy(inputs).to(DEV), torch.from_numpy(targets), torch.from_numpy(pct.copy()), torch.from_numpy(tsz))
    m = DeepSpeech(configs.LABELS, mc, "bf16", configs.AdamConfig(), configs.SpectConfig())
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    m = m.to(DEV).train()
    loss = m.training_step(mk(), 0)
    loss.backward()
    ops.check_persistent_kernels()
    own = {k: p.grad.detach().double().cpu().numpy().reshape(-1) for k, p in m.named_parameters() if k in KEYS}
    del m, loss
    torch.cuda.empty_cache()
    state = {k: torch.from_numpy(v.copy()) for k, v in P.items()}
    res = {}
    for label, ac in (("bf16", True), ("fp32", False)):
        port = TP.Port(cfg, state, DEV)
        if ac:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                ref = port.training_loss(mk())
        else:
            ref = port.training_loss(mk())
        ref.backward()
        res[label] = {k: p.grad.detach().double().cpu().numpy().reshape(-1) for k, p in port.P.items() if p.grad is not None and k in KEYS}
        del port, ref
        torch.cuda.empty_cache()
    cells = []
    for k in KEYS:
        t = res["fp32"][k]
        den = max(np.sqrt((t ** 2).sum()), 1e-30)
        a, b = np.sqrt(((own[k] - t) ** 2).sum()) / den, np.sqrt(((res["bf16"][k] - t) ** 2).sum()) / den
        tot[k][0] += a * a
        tot[k][1] += b * b
        cells.append("%-27s" % ("%.3e | %.3e" % (a, b)))
    print("%-8d %s" % (seed, " ".join(cells)), flush=True)
n = len(seeds)
print("%-8s %s" % ("rms", " ".join("%-27s" % ("%.3e | %.3e" % ((tot[k][0] / n) ** 0.5, (tot[k][1] / n) ** 0.5)) for k in KEYS)))
