#!/usr/bin/env python
"""What does gfx950's `ds_read_b64_tr_b16` return?  (Groundwork for a TN GEMM that would drop the operand transposes of the
weight-gradient GEMMs)  LDS holds u16 element i at index i; every lane passes a byte address and gets
four 16-bit elements back; the script prints, per lane, the element indices it received for a few address patterns.

    gpurun -- 'python tools/probe_ds_read_tr.py > gpurun_out/ds_read_tr.txt'"""
import ctypes
import os
import subprocess
import tempfile

import torch

SRC = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k_tr(uint32_t* out, const uint32_t* addr) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  // LDS byte address = the array's own LDS offset (passing it to the asm also keeps the initialising stores alive) + the pattern
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  const uint32_t a = base + addr[lane];
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[2 * lane] = (uint32_t)v;
  out[2 * lane + 1] = (uint32_t)(v >> 32);
}
extern "C" int run(uint32_t* out, const uint32_t* addr, void* stream) {
  hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, (hipStream_t)stream, out, addr);
  return (int)hipGetLastError();
}
'''


def main():
    d = tempfile.mkdtemp()
    src, lib = os.path.join(d, "tr.hip"), os.path.join(d, "libtr.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", lib])
    L = ctypes.CDLL(lib)
    patterns = {
        "lane*8 (contiguous 4-element groups)": [8 * l for l in range(64)],
        "row = lane%16 (row stride 64 B), chunk = lane/16 (8 B)": [(l % 16) * 64 + (l // 16) * 8 for l in range(64)],
        "row = lane/4 (row stride 64 B), chunk = lane%4 (8 B)": [(l // 4) * 64 + (l % 4) * 8 for l in range(64)],
        "row = lane%16 (row stride 32 B), chunk = lane/16 (8 B)": [(l % 16) * 32 + (l // 16) * 8 for l in range(64)],
        "row = lane%32 (row stride 64 B), chunk = lane/32 (8 B)": [(l % 32) * 64 + (l // 32) * 8 for l in range(64)],
    }
    for name, addrs in patterns.items():
        a = torch.tensor(addrs, dtype=torch.int32, device="cuda")
        out = torch.zeros(128, dtype=torch.int32, device="cuda")
        rc = L.run(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        o = out.cpu().numpy().astype("uint32")
        print("==", name, "rc", rc)
        for l in range(64):
            e = [int(o[2 * l] & 0xffff), int(o[2 * l] >> 16), int(o[2 * l + 1] & 0xffff), int(o[2 * l + 1] >> 16)]
            print("lane %2d addr %5d (elem %4d): got elements %s" % (l, addrs[l], addrs[l] // 2, e))


if __name__ == "__main__":
    main()
