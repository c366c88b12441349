"""Fault-injection helper of the GPU tests: a SQUATTER kernel that holds whole compute units (all of a CU's LDS, so that no
persistent-sweep workgroup fits beside it) for a bounded time, with its pre-condition made OBSERVABLE:

* every squatter workgroup bumps a counter in host-mapped memory when it has started, so the test can wait until the CUs are
  really taken before it launches the sweep (and can tell "the squatters never ran" from "the sweep did not notice");
* the release flag lives in host-mapped memory too: the host releases the squatters with a plain store, no stream involved.

Two ways to run it:

* `Squatter.in_process(stream)` -- a kernel on a side stream of THIS process.  HIP multiplexes streams onto a handful of hardware
  queues (GPU_MAX_HW_QUEUES, 4 by default); two streams on one hardware queue run their kernels one after the other, so a squatter
  on a stream that shares the queue of the sweep's stream never overlaps the sweep (round 5's red gate).
  `independent_streams()` returns streams that PROVABLY run concurrently with the current one and each other.
* `Squatter.second_process()` -- the same kernel from a child process (its own HSA queues): the production scenario the sweep's
  error message names ("a second process on this GPU").
"""
import ctypes
import os
import subprocess
import tempfile
import time

import torch

SRC = r"""
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <poll.h>
// holds a CU's whole LDS until `ticks` of the 100 MHz wall clock have passed or *release != 0; flags[0] = release (host writes),
// flags[1] = number of squatters that have started (device increments, host reads)
__global__ void __launch_bounds__(64) k_squat(unsigned long long ticks, int* flags, int* sink) {
  extern __shared__ unsigned char lds[];
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flags + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks && __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) __builtin_amdgcn_s_sleep(32);
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = lds[1];
}
static int launch(int blocks, unsigned long long ticks, int* flags, int* sink, hipStream_t st) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)k_squat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(k_squat, dim3(blocks), dim3(64), 160 * 1024, st, ticks, flags, sink);
  return (int)hipGetLastError();
}
extern "C" int run_squat(int blocks, unsigned long long ticks, int* flags, int* sink, void* st) {
  return launch(blocks, ticks, flags, sink, (hipStream_t)st);
}
// child-process form:  squat <blocks> <max_ms>   prints "READY <n>" when <blocks> squatters run (or "PARTIAL <n>" after 5 s),
// leaves when a line arrives on stdin, stdin closes, or max_ms have passed; prints "DONE <n>"
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 128;
  const long max_ms = argc > 2 ? atol(argv[2]) : 3000;
  int* flags = nullptr;
  if (hipHostMalloc((void**)&flags, 64, hipHostMallocMapped) != hipSuccess) { printf("ERROR hipHostMalloc\n"); return 2; }
  memset(flags, 0, 64);
  hipStream_t st;
  if (hipStreamCreate(&st) != hipSuccess) { printf("ERROR hipStreamCreate\n"); return 2; }
  if (launch(blocks, (unsigned long long)max_ms * 100000ull, flags, nullptr, st) != 0) { printf("ERROR launch\n"); return 2; }
  volatile int* vf = flags;
  for (int i = 0; i < 5000 && vf[1] < blocks; ++i) usleep(1000);
  printf("%s %d\n", vf[1] >= blocks ? "READY" : "PARTIAL", vf[1]);
  fflush(stdout);
  struct pollfd pfd = {0, POLLIN, 0};
  (void)poll(&pfd, 1, (int)max_ms);
  vf[0] = 1;
  (void)hipStreamSynchronize(st);
  printf("DONE %d\n", vf[1]);
  fflush(stdout);
  return 0;
}
"""

_BUILD = {}


def _build():
    if not _BUILD:
        d = tempfile.mkdtemp(prefix="ds2squat")
        src, lib, exe = os.path.join(d, "squat.hip"), os.path.join(d, "libsquat.so"), os.path.join(d, "squat")
        open(src, "w").write(SRC)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", lib])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", src, "-o", exe])
        L = ctypes.CDLL(lib)
        L.run_squat.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _BUILD.update(lib=L, exe=exe)
    return _BUILD


def runs_concurrently(stream, other=None):
    """True iff a kernel on `stream` and a kernel on `other` (default: the current stream) overlap in time: a one-workgroup squatter
    on `stream` is still running when a fill on `other` has completed."""
    L = _build()["lib"]
    other = other or torch.cuda.current_stream()
    flags = torch.zeros(16, dtype=torch.int32).pin_memory()
    x = torch.zeros(64, device="cuda")
    torch.cuda.synchronize()
    assert L.run_squat(1, ctypes.c_ulonglong(20_000_000), flags.data_ptr(), None, ctypes.c_void_p(stream.cuda_stream)) == 0   # <= 0.2 s
    t0 = time.perf_counter()
    while int(flags[1]) < 1 and time.perf_counter() - t0 < 0.15:
        time.sleep(0.0005)
    started = int(flags[1]) >= 1
    with torch.cuda.stream(other):
        x.fill_(1.0)
        ev = torch.cuda.Event()
        ev.record()
    t0 = time.perf_counter()
    while not ev.query() and time.perf_counter() - t0 < 0.1:
        time.sleep(0.0005)
    overlapped = started and ev.query()
    flags[0] = 1                                      # release
    torch.cuda.synchronize()
    return overlapped


def independent_streams(n=2, tries=40):
    """n torch side streams whose kernels provably overlap kernels of the current stream AND of each other (i.e. n + 1 different
    hardware queues).  Returns (streams, streams_tried)."""
    found = []
    for k in range(tries):
        s = torch.cuda.Stream()
        if runs_concurrently(s) and all(runs_concurrently(s, o) for o in found):
            found.append(s)
            if len(found) == n:
                return found, k + 1
    raise RuntimeError("only %d of %d side streams out of %d run concurrently with the current stream and each other "
                       "(GPU_MAX_HW_QUEUES too small?)" % (len(found), n, tries))


class Squatter:
    """Holds `blocks` CUs.  wait_started() -> how many squatter workgroups run; release() lets them go (idempotent)."""

    def __init__(self):
        self.proc = None
        self.flags = None
        self.blocks = 0
        self.started_n = 0

    @classmethod
    def in_process(cls, stream, blocks=128, max_s=3.0):
        self = cls()
        L = _build()["lib"]
        self.blocks = blocks
        self.flags = torch.zeros(16, dtype=torch.int32).pin_memory()
        self.sink = torch.zeros(max(blocks, 1), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        rc = L.run_squat(blocks, ctypes.c_ulonglong(int(max_s * 1e8)), self.flags.data_ptr(), self.sink.data_ptr(), ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0, rc
        return self

    @classmethod
    def second_process(cls, blocks=128, max_s=5.0):
        self = cls()
        self.blocks = blocks
        self.proc = subprocess.Popen([_build()["exe"], str(blocks), str(int(max_s * 1000))], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                     text=True)
        return self

    def wait_started(self, timeout=8.0):
        if self.proc is not None:
            line = self.proc.stdout.readline().split()           # READY n | PARTIAL n | ERROR ...
            self.started_n = int(line[1]) if len(line) == 2 and line[0] in ("READY", "PARTIAL") else 0
            return self.started_n
        t0 = time.perf_counter()
        while int(self.flags[1]) < self.blocks and time.perf_counter() - t0 < timeout:
            time.sleep(0.001)
        self.started_n = int(self.flags[1])
        return self.started_n

    def running(self):
        """Still holding its CUs (as far as the host can tell without a synchronisation)?"""
        if self.proc is not None:
            return self.proc.poll() is None
        return int(self.flags[0]) == 0

    def release(self):
        if self.proc is not None:
            if self.proc.poll() is None:
                try:
                    self.proc.stdin.write("\n")
                    self.proc.stdin.flush()
                except (BrokenPipeError, OSError):
                    pass
                try:
                    self.proc.wait(timeout=15)
                except subprocess.TimeoutExpired:
                    self.proc.kill()
            return
        if self.flags is not None:
            self.flags[0] = 1


def sweep_handshake_slots(ws, copy_stream, groups=8):
    """How many of a tuned sweep's start-up handshake slots (scratch bytes [1024, 1024 + groups * 32 * 8)) carry the signature, i.e.
    how many of its workgroups have become resident so far.  Copies on `copy_stream` (one of independent_streams(): the sweep's own
    stream is busy and the squatters' is held)."""
    host = torch.empty(groups * 32 * 2, dtype=torch.int32).pin_memory()
    s = copy_stream
    with torch.cuda.stream(s):
        host.copy_(ws[1024:1024 + groups * 32 * 8].view(torch.int32), non_blocking=True)
    t0 = time.perf_counter()
    ev = torch.cuda.Event()
    ev.record(s)
    while not ev.query():
        if time.perf_counter() - t0 > 1.0:
            return -1                                   # the copy itself is stuck behind something: treat as "cannot tell"
        time.sleep(0.001)
    hi = host.view(-1, 2)[:, 1]
    return int((hi == 0x5ca1ab1e).sum())
