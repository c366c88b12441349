#!/usr/bin/env python
"""Where does the co-residency tax of the BPTT sweep come from?  (a sweep runs at 2.45 us per time
step alone and 3.5 under the weight-gradient GEMMs.)  Times one BPTT sweep at the cfg3 shape alone and next to three synthetic
kernels that each load ONE resource of every CU -- the global -> LDS DMA path (what a GEMM's tile staging uses), the matrix pipe,
the LDS read path -- sized to run as long as the sweep, low enough in registers (<= 64) and LDS (32 KB) to share the CUs.

    gpurun -- 'python tools/probe_coresidency.py > gpurun_out/coresidency.txt'

Written at the end of round 2 without GPU time left to run it: the helper kernels are compiled here (hipcc, gfx950) and were
only compile-checked."""
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
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) void* glb_ptr_t;

// (a) global -> LDS DMA traffic only: every wave streams 16-byte chunks of a (L2-resident or HBM) buffer into LDS
extern "C" __global__ void __launch_bounds__(256, 1) k_dma(const uint4* __restrict__ src, long n_chunks, int iters, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long pos = ((long)blockIdx.x * 4 + wave) * 64 * 8;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long c = (pos + (long)j * 64 + lane) % n_chunks;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + c), (lds_ptr_t)(lds + wave * 8192 + j * 1024), 16, 0, 0);
    }
    pos += 64 * 8 * 4 * (long)gridDim.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = lds[0];
}
// (b) matrix pipe only
extern "C" __global__ void __launch_bounds__(256, 1) k_mfma(int iters, float* sink) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
// (c) LDS reads only
extern "C" __global__ void __launch_bounds__(256, 1) k_lds(int iters, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) uint4 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_uint4(i, i, i, i);
  __syncthreads();
  uint32_t s = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint4 v = lds[(threadIdx.x + j * 256 + it) & 2047];
      s += v.x ^ v.w;
    }
  }
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = s;
}
// (d) global loads into registers (L2-resident source, through the CU's texture-address / L1 path), 8 x 16 B in flight per lane
extern "C" __global__ void __launch_bounds__(256, 1) k_gload(const uint4* __restrict__ src, long n_chunks, int iters, uint32_t* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long pos = ((long)blockIdx.x * 4 + wave) * 64 * 8;
  uint32_t s = 0;
  for (int it = 0; it < iters; ++it) {
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(pos + (long)j * 64 + lane) % n_chunks];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j].x ^ v[j].w;
    pos += 64 * 8 * 4 * (long)gridDim.x;
  }
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = s;
}
// (e) VALU only: four independent fma chains per lane
extern "C" __global__ void __launch_bounds__(256, 1) k_valu(int iters, float* sink) {
  float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) { a = fmaf(a, 1.0001f, 0.5f); b = fmaf(b, 0.9999f, 0.25f); c = fmaf(c, 1.0002f, 0.125f); d = fmaf(d, 0.9998f, 0.0625f); }
  }
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
// (f) matrix pipe with the short instruction (16x16x32: 16 cycles of pipe per instruction instead of 32)
typedef __attribute__((ext_vector_type(4))) float f32x4;
extern "C" __global__ void __launch_bounds__(256, 1) k_mfma16(int iters, float* sink) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][1] + acc[6][2] + acc[7][3];
}
// (g) matrix pipe at ~1/3 duty: 4 x 32x32x16 (128 cycles of pipe), then ~256 cycles asleep
extern "C" __global__ void __launch_bounds__(256, 1) k_mfma_duty(int iters, float* sink) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    __builtin_amdgcn_s_sleep(4);
  }
  if (sink) sink[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
extern "C" int run_gload(const void* src, long n_chunks, int iters, void* sink, int grid, void* st) {
  hipLaunchKernelGGL(k_gload, dim3(grid), dim3(256), 0, (hipStream_t)st, (const uint4*)src, n_chunks, iters, (uint32_t*)sink);
  return (int)hipGetLastError();
}
extern "C" int run_valu(int iters, void* sink, int grid, void* st) {
  hipLaunchKernelGGL(k_valu, dim3(grid), dim3(256), 0, (hipStream_t)st, iters, (float*)sink);
  return (int)hipGetLastError();
}
extern "C" int run_mfma16(int iters, void* sink, int grid, void* st) {
  hipLaunchKernelGGL(k_mfma16, dim3(grid), dim3(256), 0, (hipStream_t)st, iters, (float*)sink);
  return (int)hipGetLastError();
}
extern "C" int run_mfma_duty(int iters, void* sink, int grid, void* st) {
  hipLaunchKernelGGL(k_mfma_duty, dim3(grid), dim3(256), 0, (hipStream_t)st, iters, (float*)sink);
  return (int)hipGetLastError();
}
extern "C" int run_dma(const void* src, long n_chunks, int iters, void* sink, int grid, void* st) {
  hipLaunchKernelGGL(k_dma, dim3(grid), dim3(256), 0, (hipStream_t)st, (const uint4*)src, n_chunks, iters, (uint32_t*)sink);
  return (int)hipGetLastError();
}
extern "C" int run_mfma(int iters, void* sink, int grid, void* st) {
  hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, (hipStream_t)st, iters, (float*)sink);
  return (int)hipGetLastError();
}
extern "C" int run_lds(int iters, void* sink, int grid, void* st) {
  hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 0, (hipStream_t)st, iters, (uint32_t*)sink);
  return (int)hipGetLastError();
}
"""


def build():
    d = tempfile.mkdtemp()
    src, lib = os.path.join(d, "co.hip"), os.path.join(d, "libco.so")
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
    big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)          # 64 MB source for the DMA kernel (mostly L2 / MALL hits)
    sink = torch.zeros(256 * 256, dtype=torch.float32, device=dev)

    def P(t):
        return ctypes.c_void_p(t.data_ptr())

    def S(st):
        return ctypes.c_void_p(st.cuda_stream)

    kernels = {
        "dma (global -> LDS, 8 x 1 KB per wave and iteration)": lambda it, st: L.run_dma(P(big), ctypes.c_long(big.numel() // 16), it, P(sink), 256, S(st)),
        "mfma (4 x 32x32x16 per wave and iteration)": lambda it, st: L.run_mfma(it, P(sink), 256, S(st)),
        "lds (8 x ds_read_b128 per lane and iteration)": lambda it, st: L.run_lds(it, P(sink), 256, S(st)),
        "gload (8 x 16-B global loads to registers per lane)": lambda it, st: L.run_gload(P(big), ctypes.c_long(big.numel() // 16), it, P(sink), 256, S(st)),
        "valu (64 dependent-chain fmas per lane and iteration)": lambda it, st: L.run_valu(it, P(sink), 256, S(st)),
        "mfma16 (8 x 16x16x32 per wave and iteration)": lambda it, st: L.run_mfma16(it, P(sink), 256, S(st)),
        "mfma at ~1/3 duty (4 x 32x32x16, s_sleep 4)": lambda it, st: L.run_mfma_duty(it, P(sink), 256, S(st)),
    }

    def sweep_ms(before=None):
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            if before is not None:
                before()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts[1:])

    base = sweep_ms()
    print("BPTT sweep alone: {:.3f} ms = {:.2f} us per time step".format(base, base * 1e3 / Tp))
    for name, run in kernels.items():
        it = 1000                                  # calibrate the interferer to ~1.5x the sweep's duration
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        rc = run(it, torch.cuda.current_stream())
        e1.record()
        torch.cuda.synchronize()
        alone = e0.elapsed_time(e1)
        it = max(1, int(it * 1.5 * base / max(alone, 1e-3)))
        t = sweep_ms(lambda: run(it, side))
        print("next to {:<52s} (rc {}, {} iterations, {:.2f} ms alone per 1000): {:.3f} ms = {:.2f} us per time step (+{:.2f})".format(
            name, rc, it, alone, t, t * 1e3 / Tp, (t - base) * 1e3 / Tp))
    # the real thing: the low-register weight-gradient GEMM (dW_ih shape of cfg3: [6144][1024] = dGI^T [6144][K] x Xh^T [1024][K], K = T'N)
    Kp = (Tp * N + 63) // 64 * 64
    At = torch.randn(D * G * H, Kp, device=dev).to(torch.bfloat16)
    Bt = torch.randn(H, Kp, device=dev).to(torch.bfloat16)
    for cores, label in ((True, "co-resident 128x128 low-register GEMM (x4 launches)"), (False, "full-size-tile GEMM (x4 launches; cannot share the CUs)")):
        def run_gemm(cores=cores):
            with torch.cuda.stream(side):
                for _ in range(4):
                    ops.gemm_nt(At, Bt, out_dtype=torch.float32, coresident=cores)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        run_gemm()
        e1.record(side)
        torch.cuda.synchronize()
        alone = e0.elapsed_time(e1)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record(side)
            run_gemm()
            g1.record(side)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.rnn_bwd(kind, dout, WhhT, hext, Sv, lens, D, N, H, Tp)
            e1.record()
            torch.cuda.synchronize()
            ts.append((e0.elapsed_time(e1), g0.elapsed_time(g1)))
        t, tg = min(ts[1:])
        print("next to {:<58s}: sweep {:.3f} ms = {:.2f} us per time step (+{:.2f}); the GEMMs {:.3f} ms alone, {:.3f} ms beside the sweep".format(
            label, t, t * 1e3 / Tp, (t - base) * 1e3 / Tp, alone, tg))
    ops.check_persistent_kernels()


if __name__ == "__main__":
    main()
