"""Builds libds2hip.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python -m deepspeech.pytorch_amd.build            # or __graft_entry__.build()

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot (it is git-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libds2hip.so")
SOURCES = ["ds2_norm.hip", "ds2_gemm.hip", "ds2_gemm8.hip", "ds2_rnn.hip", "ds2_rnn_persist.hip", "ds2_rnn_persist_gru.hip", "ds2_rnn_persist_lstm.hip",
           "ds2_rnn_persist_rnn.hip", "ds2_rnn_persist2_bf16_800.hip", "ds2_rnn_persist2_bf16_1280.hip", "ds2_rnn_persist2_f32_800.hip",
           "ds2_rnn_persist2_f32_1024.hip", "ds2_conv.hip", "ds2_ctc.hip", "ds2_seqops.hip", "ds2_decode.hip", "ds2_optim.hip", "ds2_spect.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-pass-failed"] + \
    os.environ.get("DS2_EXTRA_HIPCC_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, probe=False):
    """probe=True: the instrumented library for tools/probe_rnn_persist.py (-DDS2_PROBE: in-kernel cycle counters and the
    DS2_PERSIST_DBG work-skipping masks) as libds2hip_probe.so -- never loaded by the product."""
    global OBJ, LIB
    obj_dir, lib_path, flags = OBJ, LIB, FLAGS
    if probe:
        obj_dir, lib_path, flags = OBJ + "_probe", LIB.replace("libds2hip.so", "libds2hip_probe.so"), FLAGS + ["-DDS2_PROBE"]
    return _build(obj_dir, lib_path, flags, force, verbose)


def _build(OBJ, LIB, FLAGS, force, verbose):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "ds2_common.h"), os.path.join(CSRC, "ds2_rnn_persist_impl.h"), os.path.join(CSRC, "ds2_rnn_persist2_impl.h"),
               os.path.join(HERE, "..", "..", "include", "ds2hip.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        r = subprocess.run([HIPCC] + FLAGS + ["-c", s, "-o", o], capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    with ThreadPoolExecutor(max_workers=8) as ex:
        for s, rc, out in ex.map(cc, jobs):
            if verbose and out.strip():
                print(out, file=sys.stderr)
            if rc != 0:
                raise RuntimeError("hipcc failed on %s\n%s" % (s, out))
    objs = [os.path.join(OBJ, src.replace(".hip", ".o")) for src in SOURCES]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, probe="--probe" in sys.argv))
