"""Builds libds2hip.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python -m deepspeech.pytorch_amd.build            # or __graft_entry__.build()

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot (it is git-ignored).
"""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libds2hip.so")
SOURCES = ["ds2_norm.hip", "ds2_gemm.hip", "ds2_gemm8.hip", "ds2_rnn.hip", "ds2_rnn_persist.hip", "ds2_rnn_persist_gru.hip", "ds2_rnn_persist_lstm.hip",
           "ds2_rnn_persist_rnn.hip", "ds2_rnn_persist2_bf16_800.hip", "ds2_rnn_persist2_bf16_1280.hip", "ds2_rnn_persist2_f32_800.hip",
           "ds2_rnn_persist2_f32_1024.hip", "ds2_rnn_persist2_f32_1280.hip", "ds2_rnn_persist3_384.hip", "ds2_rnn_persist3_640.hip", "ds2_rnn_persist3_896.hip", "ds2_rnn_persist3_1152.hip", "ds2_rnn_persist3_1408.hip", "ds2_rnn_persist3_512.hip", "ds2_rnn_persist3_768.hip", "ds2_rnn_persist3_800.hip",
           "ds2_rnn_persist3_1024.hip", "ds2_rnn_persist3_1280.hip", "ds2_rnn_persist3_1536.hip", "ds2_conv.hip", "ds2_ctc.hip", "ds2_seqops.hip", "ds2_decode.hip", "ds2_optim.hip", "ds2_spect.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-pass-failed"] + \
    os.environ.get("DS2_EXTRA_HIPCC_FLAGS", "").split()


# ds2_rnn_persist_impl.h lands its fire-and-forget scalar loads (l2_touch) in one fixed SGPR; nothing else may name that register,
# because the write arrives asynchronously.  The device assembly of these sources (kept by -save-temps) is checked after every compile.
L2_SINK_SOURCES = ("ds2_rnn_persist_gru.hip", "ds2_rnn_persist_lstm.hip", "ds2_rnn_persist_rnn.hip")
L2_SINK = 101


def _l2_sink_misuse(obj):
    stem = os.path.splitext(obj)[0]
    asm = [f for f in glob.glob(stem + "*gfx950*.s")]
    if not asm:
        return ["no device assembly found next to %s (-save-temps=obj)" % obj]
    bad = []
    single = re.compile(r"\bs%d\b" % L2_SINK)
    rng = re.compile(r"\bs\[(\d+):(\d+)\]")
    for f in asm:
        for ln, line in enumerate(open(f), 1):
            code = line.split(";")[0]
            if not code.strip() or code.lstrip().startswith("."):
                continue
            hit = bool(single.search(code)) or any(int(a) <= L2_SINK <= int(b) for a, b in rng.findall(code))
            if hit and not re.match(r"\s*s_load_dword s%d, s\[\d+:\d+\], 0x0\s*$" % L2_SINK, code):
                bad.append("%s:%d: %s" % (os.path.basename(f), ln, code.strip()))
    return bad


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


def build_variant(name, extra_flags, force=False, verbose=False):
    """A/B builds (tools/ab_variants.py): the same sources with extra compiler flags (-DDS2_L2_AHEAD=4, -DDS2_CHUNK=8, ...) as
    libds2hip_<name>.so next to the shipping library -- never loaded by the product."""
    assert name.isidentifier() and name not in ("probe",), name
    return _build(OBJ + "_" + name, LIB.replace("libds2hip.so", "libds2hip_%s.so" % name), FLAGS + list(extra_flags), force, verbose)


def _build(OBJ, LIB, FLAGS, force, verbose):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "ds2_common.h"), os.path.join(CSRC, "ds2_rnn_persist_impl.h"), os.path.join(CSRC, "ds2_rnn_persist2_impl.h"), os.path.join(CSRC, "ds2_rnn_persist3_impl.h"),
               os.path.join(HERE, "..", "..", "include", "ds2hip.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        extra = ["-save-temps=obj"] if os.path.basename(s) in L2_SINK_SOURCES else []
        r = subprocess.run([HIPCC] + FLAGS + extra + ["-c", s, "-o", o], capture_output=True, text=True)
        out = r.stdout + r.stderr
        if extra:
            bad = _l2_sink_misuse(o) if r.returncode == 0 else []
            stem = os.path.splitext(o)[0]
            for f in glob.glob(stem + "-hip-*") + glob.glob(stem + "-host-*") + glob.glob(stem + ".hip-hip-*"):   # -save-temps by-products
                os.remove(f)
            if bad:
                os.remove(o)
                return s, 1, "%s: the L2-touch sink register is used outside the touches:\n%s" % (s, "\n".join(bad[:10]))
        return s, r.returncode, out

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
