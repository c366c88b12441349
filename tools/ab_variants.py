#!/usr/bin/env python
"""A/B compile-time variants of the kernels on ONE GPU box (box-to-box spread is +-3 %, more than most variants are worth).

Here (no GPU): build the variant libraries next to the shipping one -- they travel with the gpurun snapshot:
    python tools/ab_variants.py --build-only ahead2=-DDS2_L2_AHEAD=2 ahead4=-DDS2_L2_AHEAD=4 chunk8="-DDS2_CHUNK=8"
On the box: the shipping library and every libds2hip_<name>.so found, `repeat` interleaved passes of bench.py each:
    gpurun -- 'python tools/ab_variants.py --config cfg3 --steps 40 --repeat 2 > gpurun_out/ab.txt'
Prints ms per step and the sweeps' microseconds per time step per variant (best and mean over the passes)."""
import argparse
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "deepspeech", "pytorch_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--only", default="", help="comma-separated variant names to run (default: every library found); 'shipping' always runs")
    ap.add_argument("variants", nargs="*", help="name=flags (flags separated by spaces inside the quotes)")
    args = ap.parse_args()
    from deepspeech.pytorch_amd import build
    for v in args.variants:
        name, flags = v.split("=", 1)
        print("building %s: %s" % (name, build.build_variant(name, flags.split(), verbose=True)), file=sys.stderr)
    if args.build_only:
        return
    libs = [("shipping", os.path.join(PKG, "libds2hip.so"))]
    for f in sorted(glob.glob(os.path.join(PKG, "libds2hip_*.so"))):
        name = os.path.basename(f)[len("libds2hip_"):-3]
        if name != "probe" and (not args.only or name in args.only.split(",")):
            libs.append((name, f))
    res = {n: [] for n, _ in libs}
    for rep in range(args.repeat):
        for name, lib in libs:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_with_lib.py"), lib, "--config", args.config, "--steps",
                                str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-stock-baseline"], capture_output=True, text=True,
                               cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT))
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                print("%s: FAILED (rc %d)\n%s" % (name, r.returncode, r.stderr[-800:]))
                continue
            d = json.loads(line[-1])
            res[name].append((d["ms_per_step"], {k: v["us_per_time_step"] for k, v in d["roofline"]["recurrent_kernels"].items()}))
    print("| variant | ms per step (best / mean) | sweeps, us per time step (best pass) |")
    print("|---|---|---|")
    for name, _ in libs:
        if res[name]:
            ms = [m for m, _ in res[name]]
            best = min(res[name], key=lambda t: t[0])
            print("| %s | %.2f / %.2f | %s |" % (name, min(ms), sum(ms) / len(ms), ", ".join("%s %.3f" % kv for kv in sorted(best[1].items()))))


if __name__ == "__main__":
    main()
