#!/usr/bin/env python
"""What does ONE inter-workgroup hand-off cost on this chip, and what is the floor of a 32-way all-to-all step?

The persistent sweeps are paced by their exchange (shortening a workgroup's gate phase without moving its
publish buys nothing), so the next question is the exchange's own floor.  Two micro-benchmarks, no recurrence, no MFMA:

  ping-pong  two workgroups hand a counter back and forth (A stores i, B polls it and stores i back, A polls): cycles per round
             trip = two hops.  Pairs on ONE XCD and on two different XCDs; store forms: plain non-temporal store (what the tuned
             kernels use inside an XCD), agent-scope atomic store (write-through, the cross-XCD form); poll forms: agent-scope
             atomic load, `buffer_load ... sc1` (what the gathers issue), scalar load with glc.
  fan-in     32 workgroups publish one dword each and poll the 32 dwords of their peers, step after step (lane l polls peer l):
             the pure-exchange floor of a group step.  Groups inside one XCD (ids = g mod 8) and spread over all 8.

    gpurun -- 'python tools/probe_exchange.py > gpurun_out/exchange.txt'

First run: profiles/r03k_exchange.txt (the `buffer_load sc1` ping-pong rows of that run are void: the poll had been hoisted out of
its loop -- fixed since, not re-run).  Every spin is bounded; a time-out is reported, not waited out."""
import ctypes
import os
import subprocess
import sys
import tempfile

import torch

SRC = r"""
#include <hip/hip_runtime.h>
#include <stdint.h>
constexpr unsigned SPIN_LIMIT = 1u << 22;
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }   // HW_REG_XCC_ID[3:0]

template <int STORE>
__device__ __forceinline__ void put(uint32_t* p, uint32_t v) {
  if (STORE == 0) __builtin_nontemporal_store(v, p);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int POLL>
__device__ __forceinline__ uint32_t get(uint32_t* p, __amdgpu_buffer_rsrc_t rsrc, int off) {
  if (POLL == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (POLL == 1) {
    asm volatile("" ::: "memory");     // the intrinsic only READS memory: without this the poll is hoisted out of its loop (the
    return __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 16 /* sc1 */);   // first run of this probe "timed out" at once)
  }
  uint32_t r;
  const uint64_t a = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)p >> 32)) << 32) |
                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)p);
  asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(a) : "memory");
  return r;
}

// flags: [0] = A's word, [64] = B's word (separate 256-byte lines); out[0] = cycles, out[1] = time-outs; xcc[0/1] = XCC ids
template <int STORE, int POLL>
__global__ void __launch_bounds__(64, 1) k_pingpong(uint32_t* flags, int a, int b, int iters, unsigned long long* out, uint32_t* xcc) {
  const int me = blockIdx.x == a ? 0 : blockIdx.x == b ? 1 : -1;
  if (me < 0 || threadIdx.x != 0) return;
  xcc[me] = xcc_id();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)flags, 0, 1024, 0x00020000);
  uint32_t* mine = flags + (me ? 64 : 0);
  uint32_t* theirs = flags + (me ? 0 : 64);
  const int their_off = me ? 0 : 256;
  unsigned long long t0 = 0;
  unsigned timeouts = 0;
  for (int i = 1; i <= iters + 8; ++i) {
    if (i == 9) t0 = __builtin_readcyclecounter();           // 8 warm-up round trips
    if (me == 0) put<STORE>(mine, (uint32_t)i);
    unsigned spins = 0;
    while (get<POLL>(theirs, rsrc, their_off) != (uint32_t)i)
      if (++spins > SPIN_LIMIT) { ++timeouts; break; }
    if (me == 1) put<STORE>(mine, (uint32_t)i);
    if (timeouts) break;
  }
  if (me == 0) {
    out[0] = __builtin_readcyclecounter() - t0;
    out[1] = timeouts;
  }
}

// 32 members: workgroup ids first, first + stride, ...; slots[parity][32] dwords (one 128-byte line per parity); lane l polls peer l
template <int STORE>
__global__ void __launch_bounds__(64, 1) k_fanin(uint32_t* slots, int first, int stride, int iters, unsigned long long* out, uint32_t* xcc) {
  const int rel = blockIdx.x - first;
  if (rel < 0 || rel % stride != 0 || rel / stride >= 32) return;
  const int m = rel / stride, lane = threadIdx.x;
  if (lane == 0) xcc[m] = xcc_id();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)slots, 0, 2 * 32 * 4, 0x00020000);
  unsigned long long t0 = 0;
  unsigned timeouts = 0;
  for (int i = 1; i <= iters + 8; ++i) {
    if (i == 9) t0 = __builtin_readcyclecounter();
    const int par = i & 1;
    if (lane == 0) put<STORE>(slots + par * 32 + m, (uint32_t)i);
    unsigned spins = 0;
    for (;;) {
      uint32_t v = (uint32_t)i;
      asm volatile("" ::: "memory");
      if (lane < 32) v = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (par * 32 + lane) * 4, 0, 16 /* sc1 */);
      if (!__any(v != (uint32_t)i)) break;
      if (++spins > SPIN_LIMIT) { ++timeouts; break; }
    }
    if (timeouts) break;
  }
  if (m == 0 && lane == 0) {
    out[0] = __builtin_readcyclecounter() - t0;
    out[1] = timeouts;
  }
}

#define PP(S, P)                                                                                                              \
  extern "C" int run_pingpong_##S##_##P(void* flags, int a, int b, int iters, void* out, void* xcc, void* st) {               \
    hipLaunchKernelGGL((k_pingpong<S, P>), dim3(256), dim3(64), 0, (hipStream_t)st, (uint32_t*)flags, a, b, iters,            \
                       (unsigned long long*)out, (uint32_t*)xcc);                                                             \
    return (int)hipGetLastError();                                                                                            \
  }
PP(0, 0) PP(0, 1) PP(1, 0) PP(1, 1) PP(1, 2)
#define FI(S)                                                                                                                 \
  extern "C" int run_fanin_##S(void* slots, int first, int stride, int iters, void* out, void* xcc, void* st) {               \
    hipLaunchKernelGGL((k_fanin<S>), dim3(256), dim3(64), 0, (hipStream_t)st, (uint32_t*)slots, first, stride, iters,         \
                       (unsigned long long*)out, (uint32_t*)xcc);                                                             \
    return (int)hipGetLastError();                                                                                            \
  }
FI(0) FI(1)
"""


def build(keep=None):
    d = keep or tempfile.mkdtemp()
    src, lib = os.path.join(d, "exchange.hip"), os.path.join(d, "libexchange.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", lib])
    return lib


def main():
    if "--compile-only" in sys.argv:
        print("compiled:", build())
        return
    L = ctypes.CDLL(build())
    dev = "cuda"
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    iters = 20000
    names = {(0, 0): "nt store / atomic load", (0, 1): "nt store / buffer_load sc1", (1, 0): "atomic store / atomic load",
             (1, 1): "atomic store / buffer_load sc1", (1, 2): "atomic store / s_load glc"}
    print("ping-pong, %d round trips (one round trip = two hops)" % iters)
    for a, b, where in ((0, 8, "same XCD (ids 0, 8)"), (0, 1, "two XCDs (ids 0, 1)"), (0, 4, "two XCDs (ids 0, 4)")):
        for (s_, p_), what in names.items():
            if s_ == 0 and a % 8 != b % 8:
                continue                      # a plain store is only a valid publish inside one XCD's L2
            flags = torch.zeros(256, dtype=torch.int32, device=dev)
            out = torch.zeros(2, dtype=torch.int64, device=dev)
            xcc = torch.full((2,), -1, dtype=torch.int32, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = getattr(L, "run_pingpong_%d_%d" % (s_, p_))(P(flags), a, b, iters, P(out), P(xcc), st)
            e1.record()
            torch.cuda.synchronize()
            cyc, to = [int(v) for v in out.cpu()]
            print("  %-22s %-32s rc %d  XCC %s  %7.0f cycles per round trip  (launch %.2f ms, %.0f ns per round trip incl. launch)%s" % (
                where, what, rc, xcc.cpu().tolist(), cyc / iters, e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e6 / (iters + 8),
                "  TIME-OUT" if to else ""))
    print("fan-in of 32 workgroups, %d steps" % iters)
    for first, stride, where in ((0, 8, "one XCD (ids 0, 8, ..., 248)"), (0, 1, "eight XCDs (ids 0..31)")):
        for s_ in (0, 1):
            if s_ == 0 and stride != 8:
                continue
            slots = torch.zeros(64, dtype=torch.int32, device=dev)
            out = torch.zeros(2, dtype=torch.int64, device=dev)
            xcc = torch.full((32,), -1, dtype=torch.int32, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = getattr(L, "run_fanin_%d" % s_)(P(slots), first, stride, iters, P(out), P(xcc), st)
            e1.record()
            torch.cuda.synchronize()
            cyc, to = [int(v) for v in out.cpu()]
            print("  %-30s %-14s rc %d  XCCs %s  %7.0f cycles per step  (%.0f ns per step incl. launch)%s" % (
                where, "nt store" if s_ == 0 else "atomic store", rc, sorted(set(xcc.cpu().tolist())), cyc / iters,
                e0.elapsed_time(e1) * 1e6 / (iters + 8), "  TIME-OUT" if to else ""))


if __name__ == "__main__":
    main()
