#!/usr/bin/env python
"""Per-kernel roofline table of a bench configuration's training step from a rocprofv3 kernel-stats summary
(profiles/rNN_kernel_stats_<config>.md, tools/rocpd_stats.py): algorithmic FLOPs (or bytes) per launch from the step's shapes
 / average launch duration, against the MI355X peaks (bf16 MFMA 2.5 PFLOP/s dense, fp32 MFMA / vector 157.3
TFLOP/s, HBM 8 TB/s; /opt/skills/guides/MI355X_MICROARCH.md).

    python tools/roofline_table.py profiles/r03e_kernel_stats_cfg3.md cfg3 [steps in the trace = 3] > profiles/r03e_roofline_cfg3.md
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepspeech.pytorch_amd import synth  # noqa: E402


def main():
    stats, cfg = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "cfg3")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    kind, H, L, bi, N, tmin, tmax, dtype = bench.CONFIGS[cfg]
    G, D = bench.GATES[kind], 2 if bi else 1
    cfg_id = {"cfg2": 2, "cfg3": 3, "cfg5a": 5, "cfg5b": 6}[cfg]
    lengths = synth.synth_lengths(N, tmin, tmax, seed=cfg_id * 1000, linear=(cfg == "cfg2"))
    of = bench.out_frames(lengths)
    frames, TP = int(of.sum()), int(of.max())
    R, GH = N * TP, G * H
    peak = 2.5e15 if dtype == "bf16" else 157.3e12
    pos2, pos1 = N * 41 * TP, N * 81 * TP
    Rv = frames if frames < 0.97 * R else R      # row lists: the dense products visit only the real frames
    i2h = [2.0 * Rv * D * GH * (1344 if l == 0 else H) for l in range(L)]         # = dX = dW_ih per layer
    whh = 2.0 * D * GH * H * Rv                                                    # dW_hh of a layer (all directions)
    conv2, conv1 = 2.0 * 32 * 32 * 231 * pos2, 2.0 * 451 * 32 * pos1
    sweep = 2.0 * frames * D * GH * H
    esz = 2 if dtype == "bf16" else 4
    # kernel-name substring -> (role, bound, work per launch (flop or bytes), peak, unit)
    rows = [
        ("k_rnn_persist_bwd", "BPTT sweep (W_hh^T resident), one launch per layer", "mfma (latency-bound)", sweep, peak, "TFLOP/s"),
        ("k_rnn_persist_fwd", "forward sweep (W_hh resident)", "mfma (latency-bound)", sweep, peak, "TFLOP/s"),
        ("k_rnn_persist3_bwd", "BPTT sweep, round-4 general kernel (32 units per workgroup, two sample sets with their own step schedules)",
         "mfma (latency / exchange-volume bound)", sweep, peak, "TFLOP/s"),
        ("k_rnn_persist3_fwd", "forward sweep, round-4 general kernel", "mfma (latency / exchange-volume bound)", sweep, peak, "TFLOP/s"),
        ("k_rnn_persist2_bwd", "BPTT sweep, general kernel", "mfma (latency / exchange-volume bound)", sweep, peak, "TFLOP/s"),
        ("k_rnn_persist2_fwd", "forward sweep, general kernel", "mfma (latency / exchange-volume bound)", sweep, peak, "TFLOP/s"),
        ("k_gemm8<0", "input projections X*W_ih^T (256x256 phase-split, NT)", "mfma", sum(i2h) / L, peak, "TFLOP/s"),
        ("k_gemm8<2", "weight gradients dW_ih + dW_hh (grouped TN) + dX (NT) of a layer, one launch", "mfma", (2 * sum(i2h) + L * whh) / L, peak, "TFLOP/s"),
        ("k_gemm8<1", "weight gradients dW_ih + dW_hh of a layer (grouped TN)", "mfma", (sum(i2h) + L * whh) / L, peak, "TFLOP/s"),
        ("k_gemm_nt_bf16_big", "GEMMs on the 256x128 tile", "mfma", sum(i2h) / L, peak, "TFLOP/s"),
        ("k_gemm_nt<float>", "fp32 GEMMs (input projections, dX, weight gradients: 32x32x2 f32 MFMA), average of the step's shapes", "mfma fp32",
         (3 * sum(i2h) + L * whh) / (3 * L + D * L), 157.3e12, "TFLOP/s"),
        ("k_conv_rtap", "conv2 forward and data gradient (two 11x11 / 10x11 tap correlations each, taps in registers)", "mfma", conv2 / 2, peak, "TFLOP/s"),
        ("k_conv_tap<float", "conv2 forward / data gradient, fp32 storage (tap GEMM)", "mfma fp32", conv2 * 2 / 3, 157.3e12, "TFLOP/s"),
        ("k_conv2_wgrad", "conv2 weight gradient", "mfma", conv2, peak, "TFLOP/s"),
        ("k_conv1_fwd_mfma", "conv1 forward (MFMA, time taps padded 11 -> 16)", "mfma", conv1, peak, "TFLOP/s"),
        ("k_conv1_wgrad_mfma", "conv1 weight gradient (MFMA)", "mfma", conv1, peak, "TFLOP/s"),
        ("k_conv1_fwd<", "conv1 forward (VALU, fp32 storage)", "valu fp32", conv1, 157.3e12, "TFLOP/s"),
        ("k_conv1_wgrad<", "conv1 weight gradient (VALU, fp32 storage)", "valu fp32", conv1, 157.3e12, "TFLOP/s"),
        ("k_opt_matrix4_multi", "AdamW + bf16 layouts of every un-permuted recurrent weight matrix, one launch (32 B per parameter)", "hbm",
         D * GH * H * (2 * L - 1) * 32.0, 8e12, "TB/s"),
        ("k_opt_matrix4(", "AdamW + bf16 layouts of a recurrent weight matrix (32 B per parameter)", "hbm", GH * H * 32.0, 8e12, "TB/s"),
        ("k_add2", "direction sum (2 reads + 1 write of [T'N][H])", "hbm", 3.0 * R * H * esz, 8e12, "TB/s"),
        ("k_bn_apply<bf16_t, false", "SequenceWise BatchNorm apply (read + write of [T'N][H])", "hbm", 2.0 * R * H * esz, 8e12, "TB/s"),
        ("k_bn_bwd_apply<bf16_t, false", "SequenceWise BatchNorm backward apply (2 reads + 1 write)", "hbm", 3.0 * R * H * esz, 8e12, "TB/s"),
    ]
    got = {}
    for line in open(stats):
        m = re.match(r"\| `(.*?)`? \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            got[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    print("| kernel | role | bound | launches / step | avg us | achieved | % of peak |")
    print("|---|---|---|---|---|---|---|")
    for key, what, bound, work, pk, unit in rows:
        hit = [v for k, v in got.items() if key in k]
        if not hit:
            continue
        calls, tot = sum(v[0] for v in hit), sum(v[1] for v in hit)
        avg = tot / calls
        print("| `%s` | %s | %s | %.1f | %.1f | %.1f %s | %.1f |" % (key, what, bound, calls / steps, avg, work / (avg * 1e-6) / 1e12, unit,
                                                                     100 * work / (avg * 1e-6) / pk))
    # the general persistent kernels gather the whole exchanged vector of their group's samples every step: bytes per workgroup and
    # step over the step's duration = what one CU's vector-memory path delivers (the large-batch regime is
    # bound by this volume)
    for key, K_ in (("k_rnn_persist2_fwd", H), ("k_rnn_persist2_bwd", G * H)):
        hit = [(k, v) for k, v in got.items() if key in k]
        if not hit:
            continue
        P = H // 16
        groups_per_dir = max(1, (256 // P) // D)
        ns = -(-N // groups_per_dir)
        m = re.search(r"<\d+, (\w+), \d+, (\d+), (\d+)>", hit[0][0])
        mt = int(m.group(2)) if m else 1
        tagged = not (dtype == "bf16" and mt >= 2)
        byts = ns * K_ * esz * (2 if tagged else 1)
        calls, tot = sum(v[0] for _, v in hit), sum(v[1] for _, v in hit)
        us_step = tot / calls / TP
        print("| `%s` exchange | every workgroup gathers %d samples x %d values%s = %.0f KB per time step | vector-memory path of one CU | - | %.2f per step | %.1f GB/s per CU | - |" % (
            key, ns, K_, " (tagged granules: x2)" if tagged else "", byts / 1e3, us_step, byts / (us_step * 1e-6) / 1e9))
    total = sum(v[1] for v in got.values())
    print("\nShapes: %s (N = %d, T' = %d, H = %d, %d x %s%s, %s); %d valid frames; %d train steps in the trace; kernel time per step %.2f ms; "
          "whole step %.1f %% of the MFMA roofline (%.2f TFLOP algorithmic per step)." % (
              cfg, N, TP, H, L, "Bi" if bi else "Uni", kind.upper(), dtype, frames, steps, total / steps / 1e3,
              100 * bench.train_flops(cfg, lengths) / (total / steps * 1e-6) / peak, bench.train_flops(cfg, lengths) / 1e12))


if __name__ == "__main__":
    main()
