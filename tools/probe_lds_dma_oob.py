#!/usr/bin/env python
"""What does a buffer load STRAIGHT INTO LDS (`buffer_load_dwordx4 ... lds`, gfx950) write for a lane whose offset is out of range?
LDS is pre-filled with 0xAAAA, lanes 0-15 and 40-47 pass an offset beyond the resource.  (Groundwork for the conv2 weight-gradient
kernel's staging: if the answer is "zeros", the tile edges need no code.)

    gpurun -- 'python tools/probe_lds_dma_oob.py'"""
import ctypes
import os
import subprocess
import tempfile

import torch

SRC = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k(uint32_t* out, const uint32_t* src, int nbytes) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[512];
  for (int i = threadIdx.x; i < 512; i += blockDim.x) lds[i] = 0xAAAAAAAAu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  const int lane = threadIdx.x;
  const bool oob = lane < 16 || (lane >= 40 && lane < 48);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, oob ? 0x7ffffff0 : lane * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = lds[i];
}
extern "C" int run(uint32_t* out, const uint32_t* src, int nbytes, void* stream) {
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, (hipStream_t)stream, out, src, nbytes);
  return (int)hipGetLastError();
}
'''


def main():
    d = tempfile.mkdtemp()
    src, lib = os.path.join(d, "p.hip"), os.path.join(d, "libp.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", lib])
    L = ctypes.CDLL(lib)
    s = torch.arange(1, 257, dtype=torch.int32, device="cuda")      # 1 KiB, never zero
    out = torch.zeros(512, dtype=torch.int32, device="cuda")
    rc = L.run(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(s.data_ptr()), 1024, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype("uint32")
    print("rc", rc)
    for lane in range(64):
        print("lane %2d -> LDS dwords %s" % (lane, ["%08x" % v for v in o[4 * lane:4 * lane + 4]]))
    print("beyond the wave's 1 KiB:", ["%08x" % v for v in o[256:260]])


if __name__ == "__main__":
    main()
