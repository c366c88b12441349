#!/usr/bin/env python
"""Second co-residency probe (round 3).  The first one (tools/probe_coresidency.py, profiles/r03a_coresidency.txt) showed that a
BPTT sweep is slowed by a co-runner's global -> LDS DMA / global loads (+0.8 / +1.3 us per time step at saturation) and by
saturating 32x32x16 MFMAs (+2.6), hardly by 16x16x32 MFMAs, LDS reads or VALU work.  This one asks HOW the memory interference
depends on the co-runner's queue depth (loads in flight per wave), on where its data comes from (L2 vs MALL/HBM) and on its
rate, and how the matrix-pipe interference depends on the duty cycle -- the design inputs of a co-resident GEMM that the sweep
does not feel.

    gpurun -- 'python tools/probe_coresidency2.py > gpurun_out/coresidency2.txt'
"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepspeech.pytorch_amd import ops  # noqa: E402

SRC = r"""
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) void* glb_ptr_t;

// global -> LDS DMA with at most DEPTH 1-KiB pieces in flight per wave, `gap` s_sleep units (64 cycles each) between pieces
template <int DEPTH>
__global__ void __launch_bounds__(256, 1) k_dma(const uint4* __restrict__ src, long n_chunks, int iters, int gap, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long pos = (((long)blockIdx.x * 4 + wave) * 64 * 8) % n_chunks;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      long c = pos + (long)j * 64 + lane;
      if (c >= n_chunks) c -= n_chunks;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + c), (lds_ptr_t)(lds + wave * 8192 + j * 1024), 16, 0, 0);
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      if (gap == 1) __builtin_amdgcn_s_sleep(1);
      if (gap == 2) __builtin_amdgcn_s_sleep(2);
      if (gap == 4) __builtin_amdgcn_s_sleep(4);
      if (gap == 8) __builtin_amdgcn_s_sleep(8);
    }
    pos += 64 * 8 * 4 * (long)gridDim.x;
    if (pos >= n_chunks) pos %= n_chunks;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = lds[0];
}
// 32x32x16 MFMAs, 4 per burst, then `gap` sleep units
__global__ void __launch_bounds__(256, 1) k_mfma32(int iters, int gap, float* sink) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    if (gap == 1) __builtin_amdgcn_s_sleep(1);
    if (gap == 2) __builtin_amdgcn_s_sleep(2);
    if (gap == 4) __builtin_amdgcn_s_sleep(4);
    if (gap == 8) __builtin_amdgcn_s_sleep(8);
  }
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
// 16x16x32 MFMAs, 16 independent accumulators (8 per burst = the same flops as 4 x 32x32x16)
__global__ void __launch_bounds__(256, 1) k_mfma16(int iters, int gap, float* sink) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    if (gap == 1) __builtin_amdgcn_s_sleep(1);
    if (gap == 2) __builtin_amdgcn_s_sleep(2);
    if (gap == 4) __builtin_amdgcn_s_sleep(4);
    if (gap == 8) __builtin_amdgcn_s_sleep(8);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][i & 3];
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int run_dma(int depth, const void* src, long n_chunks, int iters, int gap, void* sink, int grid, void* st) {
  hipStream_t s = (hipStream_t)st;
  if (depth == 1) hipLaunchKernelGGL(k_dma<1>, dim3(grid), dim3(256), 0, s, (const uint4*)src, n_chunks, iters, gap, (uint32_t*)sink);
  else if (depth == 2) hipLaunchKernelGGL(k_dma<2>, dim3(grid), dim3(256), 0, s, (const uint4*)src, n_chunks, iters, gap, (uint32_t*)sink);
  else if (depth == 4) hipLaunchKernelGGL(k_dma<4>, dim3(grid), dim3(256), 0, s, (const uint4*)src, n_chunks, iters, gap, (uint32_t*)sink);
  else hipLaunchKernelGGL(k_dma<8>, dim3(grid), dim3(256), 0, s, (const uint4*)src, n_chunks, iters, gap, (uint32_t*)sink);
  return (int)hipGetLastError();
}
extern "C" int run_mfma32(int iters, int gap, void* sink, int grid, void* st) {
  hipLaunchKernelGGL(k_mfma32, dim3(grid), dim3(256), 0, (hipStream_t)st, iters, gap, (float*)sink);
  return (int)hipGetLastError();
}
extern "C" int run_mfma16(int iters, int gap, void* sink, int grid, void* st) {
  hipLaunchKernelGGL(k_mfma16, dim3(grid), dim3(256), 0, (hipStream_t)st, iters, gap, (float*)sink);
  return (int)hipGetLastError();
}
"""


def build():
    d = tempfile.mkdtemp()
    src, lib = os.path.join(d, "co2.hip"), os.path.join(d, "libco2.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", lib])
    return ctypes.CDLL(lib)


def main():
    L = build()
    dev = "cuda"
    kind, D, N, H, Tp = "gru", 2, 32, 1024, 751
    G = ops.GATES[kind]
    torch.manual_seed(0)
    GI = torch.randn(Tp * N, D * G * H, device=dev).to(torch.bfloat16)
    Whh = ((torch.rand(D, G * H, H, device=dev) * 2 - 1) / H ** 0.5).to(torch.bfloat16)
    WhhT = Whh.transpose(1, 2).contiguous()
    bhh = torch.zeros(D, G * H, device=dev)
    lens = torch.from_numpy(np.sort(np.random.RandomState(0).randint(600, Tp + 1, N))[::-1].copy().astype(np.int32)).to(dev)
    lens[0] = Tp
    dout = torch.randn(Tp, N, H, device=dev).to(torch.bfloat16)
    hext, Sv, hn, cn = ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    sink = torch.zeros(256 * 256, dtype=torch.float32, device=dev)

    def P(t):
        return ctypes.c_void_p(t.data_ptr())

    def S(st):
        return ctypes.c_void_p(st.cuda_stream)

    def timed_pair(run, which):
        """min over 3 of (sweep ms, interferer ms) with the interferer launched first on the side stream."""
        best = None
        for _ in range(4):
            torch.cuda.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if run is not None:
                g0.record(side)
                run(side)
                g1.record(side)
            e0.record()
            if which == "bwd":
                ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp)
            else:
                ops.rnn_fwd(kind, GI, Whh, bhh, lens, D, N, H, Tp)
            e1.record()
            torch.cuda.synchronize()
            r = (e0.elapsed_time(e1), g0.elapsed_time(g1) if run is not None else 0.0)
            best = r if best is None or r[0] < best[0] else best
        return best

    def alone_ms(run):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record(side)
            run(side)
            g1.record(side)
            torch.cuda.synchronize()
            ts.append(g0.elapsed_time(g1))
        return min(ts)

    base = {w: timed_pair(None, w)[0] for w in ("bwd", "fwd")}
    print("sweeps alone: BPTT {:.3f} ms = {:.2f} us per time step, forward {:.3f} ms = {:.2f} us per time step".format(
        base["bwd"], base["bwd"] * 1e3 / Tp, base["fwd"], base["fwd"] * 1e3 / Tp))

    def report(label, mk, unit_per_iter, unit):
        # calibrate the iteration count so that the interferer alone lasts ~2.5 ms (longer than either sweep)
        it = 2000
        a = alone_ms(lambda st: mk(it, st))
        it = max(50, int(it * 2.5 / max(a, 1e-3)))
        a = alone_ms(lambda st: mk(it, st))
        rate_alone = unit_per_iter * it / (a * 1e-3)
        for w in ("bwd", "fwd"):
            t, tg = timed_pair(lambda st: mk(it, st), w)
            print("{:<62s} {}: {:.2f} us per time step (+{:.2f}); co-runner {:.1f} {} alone, {:.1f} beside".format(
                label, w, t * 1e3 / Tp, (t - base[w]) * 1e3 / Tp, rate_alone, unit, unit_per_iter * it / (tg * 1e-3)))

    nbig = big.numel() // 16
    for src_name, nchunks in (("L2-resident 2 MB", (2 << 20) // 16), ("64 MB (MALL/HBM)", nbig)):
        for depth in (1, 2, 4, 8):
            for gap in ((0,) if depth > 1 else (0, 2, 8)):
                report("dma {} depth {} gap {}".format(src_name, depth, gap),
                       lambda it, st, depth=depth, gap=gap, nchunks=nchunks: L.run_dma(depth, P(big), ctypes.c_long(nchunks), it, gap, P(sink), 256, S(st)),
                       32768 / 1e9, "GB/s per CU")
    fl32 = 4 * 4 * 2 * 32 * 32 * 16 / 1e12     # TFLOP per workgroup-iteration
    for gap in (0, 1, 2, 4, 8):
        report("mfma 32x32x16 x4 per burst, gap {}".format(gap), lambda it, st, gap=gap: L.run_mfma32(it, gap, P(sink), 256, S(st)), fl32 * 256, "TFLOP/s")
    for gap in (0, 1, 2, 4, 8):
        report("mfma 16x16x32 x16 per burst, gap {}".format(gap), lambda it, st, gap=gap: L.run_mfma16(it, gap, P(sink), 256, S(st)), fl32 * 256, "TFLOP/s")
    ops.check_persistent_kernels()


if __name__ == "__main__":
    main()
