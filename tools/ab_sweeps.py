#!/usr/bin/env python
"""Kernel-level A/B of compile-time variants of the recurrent sweeps on ONE box: tools/time_sweeps.py for the shipping library and
every libds2hip_<name>.so next to it (built here with `python tools/ab_variants.py --build-only name=flags ...`).

    gpurun -- 'python tools/ab_sweeps.py gru,2,32,1024,751,ragged > gpurun_out/ab_sweeps.txt'      (--only a,b to select)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deepspeech", "pytorch_amd")
args = sys.argv[1:]
only = None
if args and args[0] == "--only":
    only, args = args[1].split(","), args[2:]
shapes = args or ["gru,2,32,1024,751,ragged"]
libs = [("shipping", os.path.join(PKG, "libds2hip.so"))]
for f in sorted(glob.glob(os.path.join(PKG, "libds2hip_*.so"))):
    name = os.path.basename(f)[len("libds2hip_"):-3]
    if name != "probe" and (only is None or name in only):
        libs.append((name, f))
for rep in range(2):
    for name, lib in libs:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_sweeps.py")] + shapes, capture_output=True, text=True, cwd=ROOT,
                           env=dict(os.environ, DS2_AB_LIB=lib))
        for line in r.stdout.splitlines():
            m = re.search(r"^(\S+).*fwd\s+\S+ ms = (\S+) us/step\s+bwd\s+\S+ ms = (\S+) us/step", line)
            if m:
                print("%-14s %-30s fwd %s  bwd %s us/step" % (name, m.group(1), m.group(2), m.group(3)), flush=True)
        if r.returncode != 0:
            print("%-14s FAILED: %s" % (name, r.stderr.strip().splitlines()[-1:] ))
