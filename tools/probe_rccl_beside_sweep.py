#!/usr/bin/env python
"""What a data-parallel gradient all-reduce costs the persistent sweeps when it runs BESIDE them on one GPU.

A 1-rank RCCL all-reduce moves nothing, and this box has one GPU, so the collective's kernel is stood in for by a synthetic kernel of
the same SHAPE: `wgs` workgroups of `threads` threads that stream HBM -> HBM (read two buffers, add, write one: the reduce-copy a ring
step does) in bursts with a sleep in between, so that its aggregate rate sits where a ring over xGMI sits (one link: ~150 GB/s on the
wire = ~450 GB/s of local HBM traffic), holding `lds` bytes of LDS and a register footprint set by the unroll depth.  The shape of
RCCL's own kernel on this box is read first (rocprofv3 kernel trace of a 2-buffer all_reduce on a 1-rank group, if it launches one).

For each co-runner shape: the cfg3 BPTT / forward sweep alone, the co-runner alone, and the sweep launched (a) while the co-runner is
already resident and (b) with the co-runner launched right behind it -- wall time per time step, the sweep's own cycle counters
(-DDS2_PROBE build: cycles that stay put while wall time grows = a lower clock; cycles that grow = contention), time-outs.

    gpurun -- 'python tools/probe_rccl_beside_sweep.py > gpurun_out/rccl_beside_sweep.txt'
"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SRC = r"""
#include <hip/hip_runtime.h>
#include <stdint.h>
// reduce-copy in bursts: every workgroup owns a contiguous slab; per burst each thread moves B x 16 bytes (B loads of each input in
// flight = the register footprint), then sleeps.  lds: dynamic LDS the workgroup holds (touched once).
template <int NT, int B>
__global__ void __launch_bounds__(NT) k_rccl_like(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ dst,
                                                  long n16_per_wg, int rounds, int sleep_units) {
  extern __shared__ uint4 hold[];
  if (threadIdx.x == 0) hold[0] = make_uint4(1, 2, 3, 4);
  const long base = (long)blockIdx.x * n16_per_wg;
  for (int r = 0; r < rounds; ++r) {
    for (long i = threadIdx.x; i + (long)(B - 1) * NT < n16_per_wg; i += (long)B * NT) {
      uint4 x[B], y[B];
#pragma unroll
      for (int k = 0; k < B; ++k) {
        x[k] = a[base + i + (long)k * NT];
        y[k] = b[base + i + (long)k * NT];
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        uint4 z;
        z.x = x[k].x + y[k].x; z.y = x[k].y + y[k].y; z.z = x[k].z + y[k].z; z.w = x[k].w + y[k].w;
        dst[base + i + (long)k * NT] = z;
      }
      for (int s = 0; s < sleep_units; ++s) __builtin_amdgcn_s_sleep(127);
    }
  }
}
#define INST(NT, B) \
  extern "C" int rccl_like_##NT##_##B(const void* a, const void* b, void* d, long n16, int wgs, int rounds, int sleep_units, int lds, void* st) { \
    static bool attr = false; \
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_rccl_like<NT, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr = true; } \
    hipLaunchKernelGGL((k_rccl_like<NT, B>), dim3(wgs), dim3(NT), lds < 16 ? 16 : lds, (hipStream_t)st, (const uint4*)a, (const uint4*)b, (uint4*)d, n16, rounds, sleep_units); \
    return (int)hipGetLastError(); }
INST(256, 2) INST(256, 8) INST(256, 24) INST(512, 4) INST(512, 16) INST(1024, 2)
"""
SHAPES = [(256, 2), (256, 8), (256, 24), (512, 4), (512, 16), (1024, 2)]


def build_helper():
    d = os.path.join(ROOT, "tools", "_build")
    os.makedirs(d, exist_ok=True)
    src, lib = os.path.join(d, "rccl_like.hip"), os.path.join(d, "librccl_like.so")
    stale = not os.path.exists(lib) or not os.path.exists(src) or open(src).read() != SRC
    if stale:
        open(src, "w").write(SRC)
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                            src, "-o", lib], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        regs, cur = {}, None
        for line in r.stderr.splitlines():
            if "Function Name" in line:
                cur = line.split("Function Name:")[1].split()[0]
            if "VGPRs:" in line and cur:
                regs[cur] = int(line.split("VGPRs:")[1].split()[0])
        open(os.path.join(d, "rccl_like_regs.txt"), "w").write("\n".join("%s %d" % kv for kv in sorted(regs.items())))
    return ctypes.CDLL(lib)


def rccl_kernel_shape():
    """rocprofv3 kernel trace of a 2-buffer all_reduce on a 1-rank RCCL group: grid / workgroup / VGPR / LDS of its kernel, if any."""
    code = ("import os, torch, torch.distributed as d\n"
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')\n"
            "d.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
            "x = torch.ones(16 << 20, device='cuda'); y = torch.ones(16 << 20, device='cuda')\n"
            "for _ in range(3): d.all_reduce(x); d.all_reduce(y)\n"
            "torch.cuda.synchronize(); d.destroy_process_group()\n")
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        open(os.path.join(td, "ar.py"), "w").write(code)
        r = subprocess.run("cd %s && TMPDIR=/tmp timeout 240 rocprofv3 --kernel-trace --output-format csv -d %s/out -- python ar.py" % (td, td),
                           shell=True, capture_output=True, text=True)
        rows = []
        for dp, _, fs in os.walk(os.path.join(td, "out")):
            for f in fs:
                if f.endswith("kernel_trace.csv"):
                    import csv
                    for row in csv.DictReader(open(os.path.join(dp, f))):
                        rows.append(row)
        seen = {}
        for row in rows:
            name = row.get("Kernel_Name", "")
            key = name[:60]
            if key not in seen:
                seen[key] = {k: row.get(k) for k in ("Grid_Size_X", "Workgroup_Size_X", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size")}
        return seen, (r.stderr or "")[-300:] if not rows else ""


def main():
    from deepspeech.pytorch_amd import _lib, build
    _lib.LIB_PATH = build.build(probe=True, verbose=False)
    from deepspeech.pytorch_amd import ops
    helper = build_helper()
    print("co-runner register footprints (hipcc):", open(os.path.join(ROOT, "tools", "_build", "rccl_like_regs.txt")).read().replace("\n", "; "))
    if os.environ.get("SKIP_RCCL_TRACE", "0") in ("", "0"):
        try:
            seen, err = rccl_kernel_shape()
            print("kernels launched by a 1-rank RCCL all_reduce (rocprofv3 kernel trace):")
            for k, v in seen.items():
                print("   ", k, v)
            if not seen:
                print("    none recorded", err)
        except Exception as e:  # noqa: BLE001
            print("RCCL kernel trace failed:", e)

    kind, D, N, H, Tp = "gru", 2, 32, 1024, 751
    G = ops.GATES[kind]
    dev = "cuda"
    torch.manual_seed(0)
    GI = torch.randn(Tp * N, D * G * H, device=dev).to(torch.bfloat16)
    Whh = ((torch.rand(D, G * H, H, device=dev) * 2 - 1) / H ** 0.5).to(torch.bfloat16)
    WhhT = Whh.transpose(1, 2).contiguous()
    bhh = torch.zeros(D, G * H, device=dev)
    lens = torch.from_numpy(np.sort(np.random.RandomState(0).randint(600, Tp + 1, N))[::-1].copy().astype(np.int32)).to(dev)
    lens[0] = Tp
    dout = torch.randn(Tp, N, H, device=dev).to(torch.bfloat16)
    side = torch.cuda.Stream()
    os.environ["DS2_PERSIST_DBG"] = "0"
    nbytes = 64 << 20                                   # per buffer: one recurrent layer's gradients (~50 MB fp32 at cfg3)
    A = torch.ones(nbytes // 4, device=dev)
    B_ = torch.ones(nbytes // 4, device=dev)
    Dst = torch.empty(nbytes // 4, device=dev)

    def counters():
        ws = ops.LAST_PERSIST_WS
        tail = ws[:1024].view(torch.int64).cpu().numpy().reshape(-1, 8)[:8]
        c = tail[0][0:4]
        return (c[0] + c[1] + c[2]) / Tp, c[3] / Tp

    hext, Sv, hn, cn = ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)

    def sweep(which):
        if which == "bwd":
            ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp)
        else:
            ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)

    def corun(nt, b, wgs, rounds, sleep_units, lds, stream):
        fn = getattr(helper, "rccl_like_%d_%d" % (nt, b))
        n16 = nbytes // 16 // wgs
        rc = fn(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B_.data_ptr()), ctypes.c_void_p(Dst.data_ptr()), ctypes.c_long(n16),
                wgs, rounds, sleep_units, lds, ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0, rc

    def timed(fn):
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None else min(best, t)
        return best

    base = {}
    for which in ("bwd", "fwd"):
        t = timed(lambda: sweep(which))
        cyc, rep = counters()
        base[which] = t
        print("%s sweep alone: %.3f ms = %.3f us per time step, %.0f cycles per step (%.2f GHz), %.2f re-polls" % (
            which, t, t * 1e3 / Tp, cyc, cyc / (t * 1e3 / Tp) / 1e3, rep))

    print("\nco-runner: reduce-copy of 2 x 64 MB -> 64 MB (192 MB of HBM traffic per pass)")
    main_stream = torch.cuda.current_stream()
    K = 1024
    # (threads, loads in flight, workgroups, LDS bytes): shapes that fit beside a sweep workgroup on its CU (the tuned cfg3 sweeps hold
    # ~344 of 512 registers per lane and ~41 KB of LDS) and, last two, shapes that do NOT (128 KB of LDS): their CUs cannot host a
    # sweep workgroup until the co-runner leaves, or the co-runner cannot start until the sweep ends
    CASES = [(256, 2, 16, 0), (256, 8, 32, 0), (256, 8, 32, 48 * K), (256, 24, 32, 48 * K), (512, 4, 32, 48 * K), (512, 16, 32, 48 * K),
             (512, 16, 64, 48 * K), (1024, 2, 32, 48 * K), (512, 16, 32, 128 * K), (512, 16, 16, 128 * K)]
    for (nt, b, wgs, lds) in CASES:
        for _once in (0,):
            for _once2 in (0,):
                # calibrate the sleep so that the co-runner alone moves ~450 GB/s of HBM traffic (a ring step's local share at one link's rate)
                chosen = None
                for sleep_units in (0, 1, 2, 4, 8, 16, 32):
                    t = timed(lambda: corun(nt, b, wgs, 1, sleep_units, lds, main_stream))
                    gbs = 3 * nbytes / (t * 1e-3) / 1e9
                    chosen = (sleep_units, t, gbs)
                    if gbs <= 500:
                        break
                sleep_units, t_alone, gbs = chosen
                rounds = max(1, int(2.5 * base["bwd"] / t_alone + 0.5))          # long enough to cover a whole sweep
                t_co = timed(lambda: corun(nt, b, wgs, rounds, sleep_units, lds, main_stream))
                line = "%4d thr x %2d loads, %2d WGs, LDS %2d KB, sleep %2d: alone %.0f GB/s (%.2f ms x %d rounds = %.2f ms)" % (
                    nt, b, wgs, lds // 1024, sleep_units, gbs, t_alone, rounds, t_co)
                for which in ("bwd", "fwd"):
                    for order in ("corunner first", "sweep first"):
                        def both():
                            if order == "corunner first":
                                with torch.cuda.stream(side):
                                    corun(nt, b, wgs, rounds, sleep_units, lds, side)
                                sweep(which)
                            else:
                                sweep(which)
                                with torch.cuda.stream(side):
                                    corun(nt, b, wgs, rounds, sleep_units, lds, side)
                        best = None
                        for _ in range(2):
                            torch.cuda.synchronize()
                            side.wait_stream(main_stream)
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            if order == "corunner first":
                                with torch.cuda.stream(side):
                                    corun(nt, b, wgs, rounds, sleep_units, lds, side)
                                e0.record()
                                sweep(which)
                                e1.record()
                            else:
                                e0.record()
                                sweep(which)
                                e1.record()
                                with torch.cuda.stream(side):
                                    corun(nt, b, wgs, rounds, sleep_units, lds, side)
                            torch.cuda.synchronize()
                            t = e0.elapsed_time(e1)
                            cyc, rep = counters()
                            if best is None or t < best[0]:
                                best = (t, cyc, rep)
                        t, cyc, rep = best
                        line += "\n        %s, %-14s: %.3f ms = %.3f us/step (x%.2f), %.0f cycles/step (%.2f GHz), %.2f re-polls" % (
                            which, order, t, t * 1e3 / Tp, t / base[which], cyc, cyc / (t * 1e3 / Tp) / 1e3, rep)
                print(line, flush=True)
                try:
                    ops.check_persistent_kernels()
                except Exception as e:  # noqa: BLE001
                    print("        !! persistent kernel time-out:", e)


if __name__ == "__main__":
    main()
