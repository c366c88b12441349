#!/usr/bin/env python
"""Per-kernel roofline table of the cfg3 training step from a rocprofv3 kernel-stats summary (profiles/rNN_kernel_stats_cfg3.md):
algorithmic FLOPs (or bytes) per launch from the step's shapes / average launch duration, against the MI355X peaks
(bf16 MFMA 2.5 PFLOP/s dense, fp32 vector 157.3 TFLOP/s, HBM 8 TB/s; /opt/skills/guides/MI355X_MICROARCH.md).

    python tools/roofline_table.py profiles/r02c_kernel_stats_cfg3.md [valid_frames] > profiles/r02c_roofline_cfg3.md
"""
import re
import sys

N, TP, H, G, D, L = 32, 751, 1024, 3, 2, 5
R = N * TP
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 21502          # valid output frames of the bench batch (bench.out_frames)
GH = G * H
pos2 = N * 41 * TP
pos1 = N * 81 * TP
i2h = [2.0 * R * D * GH * (1344 if l == 0 else H) for l in range(L)]
wih = [2.0 * D * GH * (1344 if l == 0 else H) * R for l in range(L)]
whh = 2.0 * GH * H * R
conv2 = 2.0 * 32 * 32 * 231 * pos2
conv1 = 2.0 * 451 * 32 * pos1
# kernel-name substring -> (what, bound, work per launch (flop or bytes), peak (flop/s or B/s))
ROWS = [
    ("k_rnn_persist_bwd", "BPTT sweep (W_hh^T resident), one launch per layer", "mfma (latency-bound)", 2.0 * frames * D * GH * H, 2.5e15),
    ("k_rnn_persist_fwd", "forward sweep (W_hh resident)", "mfma (latency-bound)", 2.0 * frames * D * GH * H, 2.5e15),
    ("k_gemm_nt_bf16_big", "input projections X*W_ih^T (256x128 tile)", "mfma", sum(i2h) / L, 2.5e15),
    ("k_gemm_nt_bf16_glds<true>", "weight gradients dW_ih, dW_hh (co-resident with the BPTT sweeps)", "mfma", (sum(wih) + 2 * L * whh) / (3 * L), 2.5e15),
    ("k_gemm_nt_bf16_glds<false>", "dX = dGI*W_ih (5 of 6 launches per step; + head)", "mfma", sum(i2h) / L * 5 / 6, 2.5e15),
    ("k_conv_rtap", "conv2 forward and data gradient (two 11x11 / 10x11 tap correlations each, taps in registers)", "mfma", conv2 / 2, 2.5e15),
    ("k_conv_tap<bf16_t, 1>", "conv2 forward (tap GEMM, one position tile per wave; before round 2e)", "mfma", conv2, 2.5e15),
    ("k_conv_tap<bf16_t, 2>", "conv2 data gradient (two row-parity launches; before round 2e)", "mfma", conv2 / 2, 2.5e15),
    ("k_conv2_wgrad_bf16", "conv2 weight gradient", "mfma", conv2, 2.5e15),
    ("k_conv1_fwd_mfma", "conv1 forward (MFMA, time taps padded 11 -> 16: 69 % useful flops, counted as algorithmic)", "mfma", conv1, 2.5e15),
    ("k_conv1_wgrad_mfma", "conv1 weight gradient (MFMA)", "mfma", conv1, 2.5e15),
    ("k_conv1_fwd<", "conv1 forward (Cin = 1: VALU; fp32 storage and before round 2e)", "valu fp32", conv1, 157.3e12),
    ("k_conv1_wgrad<", "conv1 weight gradient (VALU; fp32 storage and before round 2e)", "valu fp32", conv1, 157.3e12),
    ("k_transpose", "operand transposes of the weight-gradient GEMMs", "hbm", None, 8e12),
    ("k_opt_matrix", "AdamW + bf16 layouts of a recurrent weight matrix", "hbm", None, 8e12),
]
BYTES = {"k_transpose": (R * D * GH * 2 * 2 + R * H * 2 * 2 + 2 * 2 * (R * H * 2 * 2)) * 1.0 / 6,   # dGI, Xh, 2 x (h_prev, dQ): read + write
         "k_opt_matrix": GH * H * 32.0}                                                             # 32 B per parameter of a [3H][H] matrix


def main():
    rows = {}
    for line in open(sys.argv[1]):
        m = re.match(r"\| `(.*?)`? \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            rows[m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    print("| kernel | role | bound | launches / step | avg us | achieved | % of peak |")
    print("|---|---|---|---|---|---|---|")
    for key, what, bound, work, peak in ROWS:
        hit = [(k, v) for k, v in rows.items() if key in k]
        if not hit:
            continue
        calls = sum(v[0] for _, v in hit)
        tot = sum(v[1] for _, v in hit)
        avg = tot / calls
        if work is None:
            work = BYTES[key]
            rate, unit = work / (avg * 1e-6) / 1e12, "TB/s"
        else:
            rate, unit = work / (avg * 1e-6) / 1e12, "TFLOP/s"
        print("| `%s` | %s | %s | %d | %.1f | %.1f %s | %.1f |" % (key, what, bound, calls // 3, avg, rate, unit, 100 * work / (avg * 1e-6) / peak))
    print("\nShapes: cfg3 (N = 32, T' = 751, H = 1024, 5 x BiGRU); %d valid frames.  3 train steps in the trace." % frames)


if __name__ == "__main__":
    main()
