#!/usr/bin/env python
"""Round 6 groundwork for the fp32 (1e-3 parity) recurrent sweeps: `v_mfma_f32_4x4x1_16b_f32` with A-block broadcast.

A group of config 2 has 4 clips: a 16x16x4 fp32 MFMA leaves 12 of its 16 rows to padding (the fp32 matrix pipe is the floor of those
sweeps: 156 instructions x 32 cycles per wave and step).  The 4x4x1 form computes SIXTEEN independent 4x4 blocks (K = 1) per
instruction: lane l = (block l / 4, index l % 4); A: row l % 4 of block l / 4, B: column l % 4 of block l / 4, D: 4 registers = the 4
rows of column l % 4 of block l / 4.  With CBSZ = c, ABID = a the A operand of block (g * 2^c + a) is broadcast to the 2^c blocks of
its group g: all 16 blocks (c = 4) then multiply the SAME four rows (= the 4 clips at one k) with 16 different column quads (= 64 hidden
units), i.e. one instruction does 4 clips x 64 units x 1 k with no padding rows -- 4x the useful work per matrix-pipe cycle.

The script (1) checks that model of the instruction (for CBSZ = 0, 2, 4 and every ABID) against numpy on random operands, (2) times
back-to-back issue of the 4x4x1 and the 16x16x4 form on one wave per SIMD.

    gpurun -- 'python tools/probe_mfma4x4.py > gpurun_out/mfma4x4.txt'"""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import torch

CASES = [(0, 0)] + [(2, a) for a in range(4)] + [(4, a) for a in range(16)]
SRC = r'''
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) float f4;
template <int CBSZ, int ABID>
__global__ void k_one(const float* a, const float* b, float* c) {
  const int l = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, CBSZ, ABID, 0);
  for (int i = 0; i < 4; ++i) c[l * 4 + i] = acc[i];
}
#define CASE(C, A) if (cbsz == C && abid == A) hipLaunchKernelGGL((k_one<C, A>), dim3(1), dim3(64), 0, (hipStream_t)st, a, b, c);
extern "C" int run_one(const float* a, const float* b, float* c, int cbsz, int abid, void* st) {
  CASE(0, 0) CASE(2, 0) CASE(2, 1) CASE(2, 2) CASE(2, 3)
  CASE(4, 0) CASE(4, 1) CASE(4, 2) CASE(4, 3) CASE(4, 4) CASE(4, 5) CASE(4, 6) CASE(4, 7)
  CASE(4, 8) CASE(4, 9) CASE(4, 10) CASE(4, 11) CASE(4, 12) CASE(4, 13) CASE(4, 14) CASE(4, 15)
  return (int)hipGetLastError();
}
// issue rate: `iters` rounds of 16 instructions on one wave per SIMD; FORM 0: 4x4x1 with broadcast (ABID cycles), 1: 16x16x4
template <int FORM>
__global__ void __launch_bounds__(256, 1) k_rate(unsigned long long* out, float* sink, int iters) {
  const int l = threadIdx.x;
  float av[4], bv[16];
  for (int i = 0; i < 4; ++i) av[i] = (float)((l + i) & 3);
  for (int i = 0; i < 16; ++i) bv[i] = (float)((l * 3 + i) & 3);
  f4 acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (FORM == 0) {
#define M4(A) acc[(A) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[(A) >> 2], bv[A], acc[(A) & 3], 4, A, 0);
      M4(0) M4(1) M4(2) M4(3) M4(4) M4(5) M4(6) M4(7) M4(8) M4(9) M4(10) M4(11) M4(12) M4(13) M4(14) M4(15)
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[t & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t & 3], bv[t], acc[t & 3], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  sink[blockIdx.x * 256 + l] = s;
  if (l == 0) out[blockIdx.x] = t1 - t0;
}
extern "C" int run_rate(int form, unsigned long long* out, float* sink, int iters, void* st) {
  if (form == 0) hipLaunchKernelGGL((k_rate<0>), dim3(256), dim3(256), 0, (hipStream_t)st, out, sink, iters);
  else hipLaunchKernelGGL((k_rate<1>), dim3(256), dim3(256), 0, (hipStream_t)st, out, sink, iters);
  return (int)hipGetLastError();
}
'''


def model(a, b, cbsz, abid):
    """numpy model: a, b [64]; returns c [64][4]."""
    c = np.zeros((64, 4))
    for l in range(64):
        blk, j = l // 4, l % 4
        src = (blk >> cbsz << cbsz) + abid if cbsz else blk          # the block whose A rows this block multiplies
        for r in range(4):
            c[l, r] = a[src * 4 + r] * b[blk * 4 + j]
    return c


def main():
    d = tempfile.mkdtemp()
    src, lib = os.path.join(d, "p.hip"), os.path.join(d, "libp.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", lib])
    L = ctypes.CDLL(lib)
    L.run_one.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.run_rate.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rs = np.random.RandomState(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ok = True
    for cbsz, abid in CASES:
        a, b = rs.randint(-4, 5, 64).astype(np.float32), rs.randint(-4, 5, 64).astype(np.float32)
        ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        cd = torch.zeros(64, 4, device="cuda")
        assert L.run_one(ad.data_ptr(), bd.data_ptr(), cd.data_ptr(), cbsz, abid, st) == 0
        torch.cuda.synchronize()
        got, want = cd.cpu().numpy(), model(a, b, cbsz, abid)
        good = np.array_equal(got, want)
        ok &= good
        print("cbsz %d abid %2d: %s" % (cbsz, abid, "matches the model" if good else "DIFFERS (first rows got %s want %s)" % (got[:2].tolist(), want[:2].tolist())))
    print("model of v_mfma_f32_4x4x1_16b_f32 with A-block broadcast:", "CONFIRMED" if ok else "WRONG")
    out = torch.zeros(256, dtype=torch.int64, device="cuda")
    sink = torch.zeros(256 * 256, device="cuda")
    iters = 2000
    for form, name, macs in ((0, "4x4x1 x16 blocks, CBSZ 4", 256), (1, "16x16x4", 1024)):
        for _ in range(2):
            assert L.run_rate(form, out.data_ptr(), sink.data_ptr(), iters, st) == 0
            torch.cuda.synchronize()
        cyc = float(out.double().mean().item()) / (iters * 16)
        print("%-28s %.1f cycles per instruction back to back on one wave per SIMD = %.1f MAC per cycle and SIMD (with 4 clips: %.1f useful)" % (
            name, cyc, macs / cyc, (256 if form == 0 else 256) / cyc))


if __name__ == "__main__":
    main()
