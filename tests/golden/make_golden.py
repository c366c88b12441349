#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (``/root/reference/deepspeech_pytorch/model.py``, imported
unmodified through ``ref_harness``) on CPU fp32.  Run in the build container only:

    python tests/golden/make_golden.py [case ...]   # rewrites tests/golden/*.npz (all cases, or the named ones)

A fixture stores: the config, the seeds that regenerate inputs/parameters (``deepspeech.pytorch_amd.synth`` with
numpy's frozen RandomState stream), and the reference's OUTPUTS: training-step loss, logits, output lengths
(fp32 run = the reference as shipped; plus the same reference code run under ``model.double()`` as the precise
ground truth), every parameter gradient (from the float64 run; big tensors strided-subsampled + sum + L2 norm;
``noise.*`` = the fp32 run's own deviation), BatchNorm running stats after the step, eval-mode softmax output,
greedy transcripts and the hidden-state carry of reference inference.py:86-96.  Nothing here is used by the product path.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from deepspeech.pytorch_amd import synth  # noqa: E402
import ref_harness  # noqa: E402

BIG = 50000
STRIDE = 7

CASES = [
    dict(name="gru_bi_tiny", rnn_type="gru", hidden_size=32, hidden_layers=2, bidirectional=True,
         lengths=[61, 50, 37], data_seed=11, param_seed=101),
    dict(name="lstm_bi_tiny", rnn_type="lstm", hidden_size=32, hidden_layers=2, bidirectional=True,
         lengths=[58, 58, 41, 22], data_seed=12, param_seed=102),
    dict(name="rnn_bi_tiny", rnn_type="rnn", hidden_size=32, hidden_layers=2, bidirectional=True,
         lengths=[47, 33], data_seed=13, param_seed=103),
    dict(name="gru_uni_la", rnn_type="gru", hidden_size=32, hidden_layers=2, bidirectional=False,
         lookahead_context=20, lengths=[90, 71, 64], data_seed=14, param_seed=104),
    dict(name="lstm_uni_la", rnn_type="lstm", hidden_size=48, hidden_layers=3, bidirectional=False,
         lookahead_context=7, lengths=[66, 40], data_seed=15, param_seed=105),
    # BN gains large enough that Hardtanh's upper clamp (20) is hit; one sample with more labels than frames
    # (infeasible -> zero_infinity path, loss and grad contributions are 0, model.py:203)
    dict(name="gru_bi_clamp_inf", rnn_type="gru", hidden_size=48, hidden_layers=3, bidirectional=True,
         lengths=[80, 77, 30, 21], data_seed=16, param_seed=106, bn_gain=14.0, chars_per_second=12.0,
         long_target_sample=3),
    dict(name="gru_bi_mid", rnn_type="gru", hidden_size=64, hidden_layers=3, bidirectional=True,
         lengths=[141, 130, 122, 101, 90], data_seed=17, param_seed=107),
    dict(name="single_sample", rnn_type="lstm", hidden_size=32, hidden_layers=2, bidirectional=True,
         lengths=[73], data_seed=18, param_seed=108),
    # the headline width (BASELINE.json config 3: hidden 1024, bi-directional GRU): the shape the persistent recurrent
    # kernels and the DMA-staged GEMMs cover, pinned to the real reference (2 layers / 4 short clips keep the CPU run and
    # the fixture small; big gradients are subsampled with a large stride)
    dict(name="gru_bi_1024", rnn_type="gru", hidden_size=1024, hidden_layers=2, bidirectional=True,
         lengths=[121, 101, 90, 77], data_seed=19, param_seed=109, stride=1009),
    # uni-directional LSTM + Lookahead at the same width (the BASELINE config-5b direction/cell type)
    dict(name="lstm_uni_1024_la", rnn_type="lstm", hidden_size=1024, hidden_layers=2, bidirectional=False,
         lookahead_context=20, lengths=[111, 96, 83], data_seed=20, param_seed=110, stride=1009),
    dict(name="rnn_bi_1024", rnn_type="rnn", hidden_size=1024, hidden_layers=2, bidirectional=True,
         lengths=[101, 88, 61, 45, 33], data_seed=21, param_seed=111, stride=1009),
    # the reference's DEFAULT model family (configs/train_config.py:46-50: bidirectional LSTM, hidden 1024) at its own width
    dict(name="lstm_bi_1024", rnn_type="lstm", hidden_size=1024, hidden_layers=2, bidirectional=True,
         lengths=[127, 111, 94, 80, 52], data_seed=26, param_seed=116, stride=1009),
    # ---- BASELINE.json configurations at their own size / width (round 2) -------------------------------------
    # configs[1] (AN4 shape) at FULL size: 5 x BiGRU-800, 8 clips of 2.01 ... 1.01 s, fp32
    dict(name="cfg2_full", rnn_type="gru", hidden_size=800, hidden_layers=5, bidirectional=True,
         lengths=[201, 187, 172, 158, 144, 130, 115, 101], data_seed=22, param_seed=112, stride=1009),
    # configs[4] family: LSTM hidden 1280, bi-directional (5a) and uni-directional + Lookahead (5b)
    dict(name="lstm_bi_1280", rnn_type="lstm", hidden_size=1280, hidden_layers=2, bidirectional=True,
         lengths=[131, 120, 101, 77, 60], data_seed=23, param_seed=113, stride=1009),
    dict(name="lstm_uni_1280_la", rnn_type="lstm", hidden_size=1280, hidden_layers=2, bidirectional=False,
         lookahead_context=20, lengths=[125, 110, 96, 71], data_seed=24, param_seed=114, stride=1009),
    # configs[2] at full DEPTH and BATCH (5 x BiGRU-1024, 32 clips), clips shortened to 2.4-3 s so that the reference's
    # fp32 + fp64 CPU runs stay in minutes
    dict(name="gru_bi_1024_l5_n32", rnn_type="gru", hidden_size=1024, hidden_layers=5, bidirectional=True,
         lengths=[301, 300, 298, 297, 295, 293, 291, 290, 288, 286, 284, 282, 281, 279, 277, 275, 273, 272, 270, 268, 266, 264,
                  262, 260, 258, 256, 254, 252, 249, 247, 244, 241], data_seed=25, param_seed=115, stride=4001),
    # ---- round 3 --------------------------------------------------------------------------------------------------------
    # the reference's OWN test model (tests/smoke_test.py:216-219: BiDirectionalConfig(hidden_size=10, hidden_layers=1), default
    # cell = LSTM, batch_size=10): a hidden size that is not a multiple of the 16-unit MFMA tile
    dict(name="lstm_bi_h10_n10", rnn_type="lstm", hidden_size=10, hidden_layers=1, bidirectional=True,
         lengths=[151, 143, 131, 120, 111, 97, 84, 70, 61, 45], data_seed=31, param_seed=131),
    # what else the reference accepts and the kernels used to refuse: 45 output classes (> 32), lookahead context 40 (> 32;
    # train_config.py:55 is a free int), hidden size 50
    dict(name="gru_uni_h50_la40_c45", rnn_type="gru", hidden_size=50, hidden_layers=2, bidirectional=False,
         lookahead_context=40, n_labels=45, lengths=[141, 120, 93, 66], data_seed=32, param_seed=132),
    # a label set beyond 256 classes (round 5: the CTC kernels no longer stage class rows in LDS for such sets)
    dict(name="gru_bi_h32_c300", rnn_type="gru", hidden_size=32, hidden_layers=2, bidirectional=True, n_labels=300,
         lengths=[121, 100, 77], data_seed=34, param_seed=134),
    # another spectrogram geometry (round 5): 8 kHz audio = 81 frequency bins -> 41 -> 21 rows, 672 RNN input features
    # (SpectConfig.sample_rate, model.py:166-169); and 129 bins (an odd row count after conv1: 65 -> 33)
    dict(name="gru_bi_8khz", rnn_type="gru", hidden_size=32, hidden_layers=2, bidirectional=True, sample_rate=8000,
         lengths=[131, 101, 88, 60], data_seed=36, param_seed=136),
    dict(name="lstm_uni_12k8", rnn_type="lstm", hidden_size=32, hidden_layers=2, bidirectional=False, lookahead_context=20,
         sample_rate=12800, lengths=[97, 80, 41], data_seed=37, param_seed=137),
    dict(name="rnn_bi_h24_c40", rnn_type="rnn", hidden_size=24, hidden_layers=2, bidirectional=True, n_labels=40,
         lengths=[99, 80, 55], data_seed=33, param_seed=133),
    # configs[4] at its own DEPTH (7 layers) and width (LSTM-1280), bi-directional and uni-directional + Lookahead, 18 clips (two
    # 16-sample m-tiles per group in the general persistent kernels), clips shortened so that the float64 CPU run stays in minutes
    dict(name="lstm_bi_1280_l7_n18", rnn_type="lstm", hidden_size=1280, hidden_layers=7, bidirectional=True,
         lengths=[81, 80, 78, 77, 75, 73, 71, 70, 68, 66, 64, 62, 61, 59, 57, 55, 53, 51], data_seed=34, param_seed=134, stride=8009),
    dict(name="lstm_uni_1280_la_l7_n18", rnn_type="lstm", hidden_size=1280, hidden_layers=7, bidirectional=False, lookahead_context=20,
         lengths=[83, 81, 80, 78, 75, 74, 71, 69, 68, 65, 64, 61, 60, 58, 56, 54, 52, 50], data_seed=35, param_seed=135, stride=8009),
]


def make_batch(case):
    lengths = np.asarray(case["lengths"], dtype=np.int64)
    inputs, targets, pct, tsz = synth.synth_batch(lengths, case["data_seed"],
                                                  chars_per_second=case.get("chars_per_second", 12.0), n_labels=case.get("n_labels", 29),
                                                  n_freq=case.get("sample_rate", 16000) // 100 + 1)
    if "long_target_sample" in case:   # make one sample infeasible: more labels than output frames
        i = case["long_target_sample"]
        rs = np.random.RandomState(case["data_seed"] + 999)
        tsz = tsz.copy()
        parts, off = [], 0
        for j, s in enumerate(tsz):
            if j == i:
                parts.append(rs.randint(1, 29, size=int(lengths[i])).astype(np.int64))  # S = T > T'
            else:
                parts.append(targets[off:off + s])
            off += s
        tsz[i] = int(lengths[i])
        targets = np.concatenate(parts)
    return inputs, targets, pct, tsz


def make_params(case, shapes):
    P = synth.synth_params(shapes, case["param_seed"])
    g = case.get("bn_gain")
    if g:
        for k in ("conv.seq_module.1.weight", "conv.seq_module.4.weight"):
            P[k] = (P[k] * g).astype(np.float32)
    return P


def case_labels(ns, case):
    return ref_harness.extended_labels(case["n_labels"]) if "n_labels" in case else ns.labels


def fresh_model(ns, case, P, double=False):
    model = ref_harness.build_reference_model(ns, case["rnn_type"], case["hidden_size"], case["hidden_layers"],
                                              case["bidirectional"], case.get("lookahead_context", 20), labels=case_labels(ns, case),
                                              sample_rate=case.get("sample_rate", 16000))
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.items()}, strict=True)
    return model.double() if double else model


def run_train_step(model, inputs, targets, pct, tsz, double):
    """DeepSpeech.training_step (model.py:241-249) restated only to keep the logits; asserted equal to the
    reference's own training_step below."""
    model.train()
    model.zero_grad()
    x = torch.from_numpy(inputs)
    input_sizes = torch.from_numpy(pct.copy()).mul_(int(inputs.shape[3])).int()
    logits, out_sizes, _ = model(x.double() if double else x, input_sizes)
    lp = logits.transpose(0, 1).log_softmax(-1)
    loss = model.criterion(lp, torch.from_numpy(targets), out_sizes, torch.from_numpy(tsz))
    loss.backward()
    return loss, logits, out_sizes, input_sizes


def main():
    ns = ref_harness.load_reference()
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    for case in CASES:
        if only and case["name"] not in only:
            continue
        stride = case.get("stride", STRIDE)
        probe = ref_harness.build_reference_model(ns, case["rnn_type"], case["hidden_size"], case["hidden_layers"],
                                                  case["bidirectional"], case.get("lookahead_context", 20), labels=case_labels(ns, case),
                                              sample_rate=case.get("sample_rate", 16000))
        shapes = {k: tuple(v.shape) for k, v in probe.state_dict().items()}
        P = make_params(case, shapes)
        inputs, targets, pct, tsz = make_batch(case)
        out = {}
        # ---- (A) the reference exactly as shipped: fp32 on CPU
        m32 = fresh_model(ns, case, P)
        loss32, logits32, out_sizes, input_sizes = run_train_step(m32, inputs, targets, pct, tsz, False)
        out["loss"] = np.float64(loss32.item())
        out["logits"] = logits32.detach().numpy()
        out["output_lengths"] = out_sizes.numpy().astype(np.int32)
        out["input_sizes"] = input_sizes.numpy().astype(np.int32)
        for k, b in m32.named_buffers():
            if k.endswith("running_mean") or k.endswith("running_var"):
                out["running." + k] = b.detach().numpy().copy()
        # the reference's own training_step must give the same loss (sanity of the restated step above)
        chk = fresh_model(ns, case, P)
        chk.train()
        l2 = chk.training_step((torch.from_numpy(inputs), torch.from_numpy(targets), torch.from_numpy(pct.copy()),
                                torch.from_numpy(tsz)), 0)
        assert abs(l2.item() - loss32.item()) <= 1e-5 * max(1.0, abs(loss32.item())), (l2.item(), loss32.item())
        # ---- (B) the same reference code run in float64 (model.double()): the precise ground truth.  Gradients
        # are stored from this run (as float32); "noise.<param>" records how far the fp32 run (A) is from it, i.e.
        # the reference's own fp32 rounding noise (up to ~1e-3 relative on the conv weight gradients).
        m64 = fresh_model(ns, case, P, double=True)
        loss64, logits64, _, _ = run_train_step(m64, inputs, targets, pct, tsz, True)
        out["loss64"] = np.float64(loss64.item())
        out["logits64"] = logits64.detach().numpy().astype(np.float32)
        g32 = {k: p.grad.detach().numpy() for k, p in m32.named_parameters()}
        for k, p in m64.named_parameters():
            g = p.grad.detach().numpy()
            sc = max(np.abs(g).max(), 1e-30)
            out["noise." + k] = np.float64(np.abs(g32[k].astype(np.float64) - g).max() / sc)
            if g.size > BIG:
                out["gradsub." + k] = g.reshape(-1)[::stride].astype(np.float32)
                out["gradsum." + k] = np.float64(g.sum())
                out["gradl2." + k] = np.float64(np.sqrt((g ** 2).sum()))
            else:
                out["grad." + k] = g.astype(np.float32)
        # ---- (C) the reference under torch.autocast(bfloat16) (what Lightning's precision=16/bf16 plugin wraps the step
        # in; CPU autocast here -- the GPU box has no reference): loss, logits and every gradient, plus "acnoise.<param>" =
        # relative L2 distance of its gradient from the float64 run (B).  The bf16 mode of the HIP path is held against
        # these (tests/test_gpu_model.py::test_bf16_train_step_vs_reference_autocast).
        mac = fresh_model(ns, case, P)
        mac.train()
        mac.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            xin = torch.from_numpy(inputs)
            isz = torch.from_numpy(pct.copy()).mul_(int(inputs.shape[3])).int()
            lg_ac, osz_ac, _ = mac(xin, isz)
            lp_ac = lg_ac.transpose(0, 1).log_softmax(-1)
            loss_ac = mac.criterion(lp_ac, torch.from_numpy(targets), osz_ac, torch.from_numpy(tsz))
        loss_ac.backward()
        out["loss_ac"] = np.float64(loss_ac.item())
        out["logits_ac"] = lg_ac.detach().float().numpy()
        g64 = {k: p.grad.detach().numpy() for k, p in m64.named_parameters()}
        for k, p in mac.named_parameters():
            g = p.grad.detach().float().numpy().astype(np.float64)
            out["acnoise." + k] = np.float64(np.sqrt(((g - g64[k]) ** 2).sum()) / max(np.sqrt((g64[k] ** 2).sum()), 1e-30))
            if g.size > BIG:
                out["gradsub_ac." + k] = g.reshape(-1)[::stride].astype(np.float32)
            else:
                out["grad_ac." + k] = g.astype(np.float32)
        # ---- eval forward (fresh weights = the fixture's running stats), softmax probs + transcripts
        mev = fresh_model(ns, case, P)
        mev.eval()
        x = torch.from_numpy(inputs)
        with torch.no_grad():
            probs, sizes, hs = mev(x, input_sizes)
        out["eval_probs"] = probs.numpy()
        dec = ns.GreedyDecoder(case_labels(ns, case))
        strings, _ = dec.decode(probs, sizes)
        transcripts = [s[0] for s in strings]
        # ---- hidden-state carry (inference.py:86-96): batch 1, feed hs back in
        with torch.no_grad():
            x1 = x[:1, :, :, :int(case["lengths"][0])]
            l1 = torch.tensor([int(case["lengths"][0])], dtype=torch.int)
            _, _, hs1 = mev(x1, l1)
            probs2, _, hs2 = mev(x1, l1, hs1)
        out["carry_probs"] = probs2.numpy()
        if case["rnn_type"] == "lstm":
            out["carry_h_last"] = hs2[-1][0].numpy()
            out["carry_c_last"] = hs2[-1][1].numpy()
        else:
            out["carry_h_last"] = hs2[-1].numpy()
        meta = dict(case)
        meta["transcripts"] = transcripts
        meta["labels"] = case_labels(ns, case)
        meta["shapes"] = {k: list(v) for k, v in shapes.items()}
        meta["torch_version"] = torch.__version__
        meta["big"] = BIG
        meta["stride"] = stride
        out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **out)
        worst = max(float(out[k]) for k in out if k.startswith("noise."))
        print("%-20s loss32=%.6f loss64=%.6f T'=%s ref-fp32-noise(max rel grad)=%.1e file=%.0f KB" % (
            case["name"], loss32.item(), loss64.item(), out["output_lengths"].tolist(), worst,
            os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
