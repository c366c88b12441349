#!/usr/bin/env python
"""What does gfx950's `v_smfmac_f32_16x16x64_bf16` compute, and what does it cost?  (Groundwork for the recurrent sweeps: a group of
<= 8 clips fills only half of the 16 rows of a dense 16x16x32 MFMA tile.  With the structured-sparse form the 8 clips can use all 16
rows: row s carries the k = 0, 1 (mod 4) elements of clip s, row s + 8 the k = 2, 3 (mod 4) ones -- each row is 2:4 sparse by
construction, one instruction covers K = 64 instead of 32, and C[s] + C[s + 8] is the full dot product.)

The script (1) discovers, with one-hot operands, which dense-B slot (lane, element) every compressed-A slot (lane, element) is
multiplied with for a given index word, (2) checks the layout model the kernels assume against random operands, (3) times
back-to-back issue of the sparse and the dense instruction on one wave per SIMD.

    gpurun -- 'python tools/probe_smfmac.py > gpurun_out/smfmac.txt'"""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import torch

SRC = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) __bf16 bf16v;
typedef __attribute__((ext_vector_type(4))) float f4;
// one wave: a = 64 lanes x 8 bf16 (compressed sparse A), b = 64 lanes x 16 bf16 (dense B), idx = 64 index words, c = 64 x 4 floats
__global__ void k_one(const uint16_t* a, const uint16_t* b, const int* idx, float* c, int abid) {
  const int l = threadIdx.x;
  bf8 av;
  bf16v bv;
  for (int i = 0; i < 8; ++i) av[i] = __builtin_bit_cast(__bf16, a[l * 8 + i]);
  for (int i = 0; i < 16; ++i) bv[i] = __builtin_bit_cast(__bf16, b[l * 16 + i]);
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (abid == 0) acc = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(av, bv, acc, idx[l], 0, 0);
  else acc = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(av, bv, acc, idx[l], 0, 1);
  for (int i = 0; i < 4; ++i) c[l * 4 + i] = acc[i];
}
extern "C" int run_one(const void* a, const void* b, const void* idx, void* c, int abid, void* st) {
  hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, (hipStream_t)st, (const uint16_t*)a, (const uint16_t*)b, (const int*)idx, (float*)c, abid);
  return (int)hipGetLastError();
}
// issue rate: `iters` rounds of 6 independent accumulators, one wave per SIMD (256 threads, 512-register budget not needed here)
template <int SPARSE>
__global__ void __launch_bounds__(256, 1) k_rate(unsigned long long* out, float* sink, int iters) {
  const int l = threadIdx.x;
  bf8 a8;
  bf16v b16;
  for (int i = 0; i < 8; ++i) a8[i] = (__bf16)(float)((l + i) & 3);
  for (int i = 0; i < 16; ++i) b16[i] = (__bf16)(float)((l * 3 + i) & 3);
  f4 acc[6];
  for (int t = 0; t < 6; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
  const int idx = (l & 8) ? 0xEEEE : 0x4444;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      if (SPARSE) acc[t] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a8, b16, acc[t], idx, 0, 0);
      else {
        bf8 b8;
        for (int i = 0; i < 8; ++i) b8[i] = b16[i];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[t], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 6; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((l & 63) == 0) out[blockIdx.x * 4 + (l >> 6)] = t1 - t0;
  if (s == 12345.678f) sink[0] = s;
}
extern "C" int run_rate(int sparse, void* out, void* sink, int iters, int blocks, void* st) {
  if (sparse) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, (hipStream_t)st, (unsigned long long*)out, (float*)sink, iters);
  else hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, (hipStream_t)st, (unsigned long long*)out, (float*)sink, iters);
  return (int)hipGetLastError();
}
'''


def bf16_bits(x):
    """float32 array -> uint16 bf16 bit patterns (values used here are exactly representable)"""
    return (np.asarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def model(a, b, idxw):
    """The layout the kernels assume.  a [64][8] compressed A, b [64][16] dense B, idxw [64] index words -> c [64][4].
    lane = 16*lq + i: A row i, compressed elements 8*lq..8*lq+7 = logical k 16*lq + 4*(e/2) + index(e); B column i as two dense
    16x16x32 fragments: elements 0..7 = k 8*lq + e, elements 8..15 = k 32 + 8*lq + (e - 8);
    index of compressed element e = bits [2e+1:2e] of the lane's word; D row 4*lq + r, column i in c[lane][r]."""
    A = np.zeros((16, 64))
    B = np.zeros((64, 16))
    for lane in range(64):
        i, lq = lane % 16, lane // 16
        for e in range(8):
            pos = (int(idxw[lane]) >> (2 * e)) & 3
            A[i, 16 * lq + 4 * (e // 2) + pos] += a[lane, e]
        for e in range(16):
            B[32 * (e // 8) + 8 * lq + (e % 8), i] = b[lane, e]
    D = A @ B
    c = np.zeros((64, 4))
    for lane in range(64):
        i, lq = lane % 16, lane // 16
        for r in range(4):
            c[lane, r] = D[4 * lq + r, i]
    return c


def main():
    d = tempfile.mkdtemp()
    src, lib = os.path.join(d, "sm.hip"), os.path.join(d, "libsm.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", lib])
    L = ctypes.CDLL(lib)
    dev = "cuda"
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(a, b, idxw, abid=0):
        ta = torch.from_numpy(bf16_bits(a).view(np.int16).reshape(-1).copy()).to(dev)
        tb = torch.from_numpy(bf16_bits(b).view(np.int16).reshape(-1).copy()).to(dev)
        ti = torch.from_numpy(np.asarray(idxw, dtype=np.int64).astype(np.uint32).view(np.int32).copy()).to(dev)
        tc = torch.zeros(256, dtype=torch.float32, device=dev)
        rc = L.run_one(ctypes.c_void_p(ta.data_ptr()), ctypes.c_void_p(tb.data_ptr()), ctypes.c_void_p(ti.data_ptr()), ctypes.c_void_p(tc.data_ptr()), abid, st)
        torch.cuda.synchronize()
        assert rc == 0, rc
        return tc.cpu().numpy().reshape(64, 4).astype(np.float64)

    # (2) first: the model against random small-integer operands and random VALID index words (first index < second index)
    rs = np.random.RandomState(0)
    pairs = [(p, q) for p in range(4) for q in range(4) if p < q]
    worst = 0.0
    for trial in range(20):
        a = rs.randint(-3, 4, (64, 8)).astype(np.float64)
        b = rs.randint(-3, 4, (64, 16)).astype(np.float64)
        idxw = np.zeros(64, dtype=np.int64)
        for lane in range(64):
            w = 0
            for g in range(4):
                p, q = pairs[rs.randint(len(pairs))]
                w |= (p | (q << 2)) << (4 * g)
            idxw[lane] = w | (rs.randint(0, 1 << 16) << 16)        # upper half: garbage that ABID = 0 must ignore
        got, want = run(a, b, idxw), model(a, b, idxw)
        worst = max(worst, float(np.abs(got - want).max()))
    print("layout model vs hardware, 20 random trials (ABID 0): max abs difference %g  -> %s" % (worst, "MODEL HOLDS" if worst == 0 else "MODEL WRONG"))
    # ABID = 1 should take the upper 16 bits
    a = rs.randint(-3, 4, (64, 8)).astype(np.float64)
    b = rs.randint(-3, 4, (64, 16)).astype(np.float64)
    lo = np.full(64, 0x4444, dtype=np.int64)
    hi = np.full(64, 0xEEEE, dtype=np.int64)
    d1 = float(np.abs(run(a, b, lo | (hi << 16), abid=1) - model(a, b, hi)).max())
    print("ABID 1 selects the upper 16 index bits: max abs difference %g" % d1)
    # the construction the sweeps use: 8 clips on 16 rows
    h = rs.randint(-3, 4, (8, 64)).astype(np.float64)
    W = rs.randint(-3, 4, (64, 16)).astype(np.float64)
    a = np.zeros((64, 8))
    b = np.zeros((64, 16))
    idxw = np.zeros(64, dtype=np.int64)
    for lane in range(64):
        i, lq = lane % 16, lane // 16
        s, odd = i % 8, i // 8
        for e in range(8):
            a[lane, e] = h[s, 16 * lq + 4 * (e // 2) + 2 * odd + (e % 2)]
        idxw[lane] = 0xEEEE if odd else 0x4444
        for e in range(16):
            b[lane, e] = W[32 * (e // 8) + 8 * lq + (e % 8), i]
    c = run(a, b, idxw)
    D = np.zeros((16, 16))
    for lane in range(64):
        i, lq = lane % 16, lane // 16
        for r in range(4):
            D[4 * lq + r, i] = c[lane, r]
    err = float(np.abs(D[:8] + D[8:] - h @ W).max())
    print("8 clips on 16 rows (rows s | s+8 = k mod 4 in {0,1} | {2,3}): max abs difference from h @ W: %g" % err)

    if True:
        # (1) discovery: which B slot does every A slot meet?  B slot (lane, e) carries the code 1 + 16*(lane/16) + e (< 65: exact in
        # bf16) in column lane%16; A one-hot at (row lanes, element e) -> D[row][col] = the code of the B slot it was multiplied with
        print("-- discovery (A slot -> B slot it multiplies), per index word")
        b = np.zeros((64, 16))
        for lane in range(64):
            for e in range(16):
                b[lane, e] = 1 + 16 * (lane // 16) + e
        for word in (0x4444, 0xEEEE, 0x8888, 0xDDDD, 0x9999):
            print("index word %#06x" % word)
            for lq in range(4):
                row = []
                for e in range(8):
                    a = np.zeros((64, 8))
                    for i in range(16):
                        a[16 * lq + i, e] = 1.0
                    c = run(a, b, np.full(64, word, dtype=np.int64))
                    vals = set(int(v) for v in c.reshape(-1))
                    row.append(sorted(vals - {0}))
                print("   A lanes 16*%d+i, elements 0..7 meet B codes (1 + 16*lq_b + e_b): %s" % (lq, row))

    # (3) issue rate
    out = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    for name, sp in (("v_mfma_f32_16x16x32_bf16 (dense)", 0), ("v_smfmac_f32_16x16x64_bf16 (sparse)", 1)):
        for blocks in (1, 256):
            for rep in range(2):
                iters = 2000
                rc = L.run_rate(sp, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(sink.data_ptr()), iters, blocks, st)
                torch.cuda.synchronize()
            cyc = out[:blocks * 4].cpu().numpy().astype(np.float64) / (iters * 6)
            print("%-40s %3d blocks: %.2f cycles per instruction per wave (min %.2f max %.2f; 6 independent accumulators, one wave per SIMD)" % (
                name, blocks, cyc.mean(), cyc.min(), cyc.max()))


if __name__ == "__main__":
    main()
