"""Safety of the persistent recurrent sweeps next to other work on the chip (what a data-parallel run does to them: RCCL kernels
on their own stream while the sweeps own the CUs).

* a >= 64 MB all-reduce on a 1-rank RCCL group, enqueued on a side stream, runs concurrently with a full cfg3 BPTT sweep: the sweep
  finishes without a time-out, bit-identical to the sweep alone; the slow-down is printed;
* FAULT INJECTION: a squatter kernel that holds half of the chip's CUs (all of their LDS) while a sweep starts, with the spin
  budget lowered through the C ABI's test hook (ds2_rnn_persist_set_spin_limit): the resident half of the sweep gives up, the
  launch ENDS (bounded spins), the device error word is raised, the outputs are NaN-poisoned, and
  ops.poll_persistent_error -- what DeepSpeech.training_step calls once per step -- raises at the latest one call later;
  afterwards a clean sweep works again."""
import ctypes
import os
import socket
import subprocess
import tempfile
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

SQUAT_SRC = r"""
#include <hip/hip_runtime.h>
#include <stdint.h>
// occupies a CU's whole LDS (so that no persistent-sweep workgroup fits beside it) until `until` (wall-clock ticks, 100 MHz) or
// until *release becomes non-zero
__global__ void __launch_bounds__(64) k_squat(unsigned long long ticks, int* release, int* sink) {
  extern __shared__ unsigned char lds[];
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks && __hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(32);
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = lds[1];
}
extern "C" int run_squat(int blocks, unsigned long long ticks, int* release, int* sink, void* st) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)k_squat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(k_squat, dim3(blocks), dim3(64), 160 * 1024, (hipStream_t)st, ticks, release, sink);
  return (int)hipGetLastError();
}
"""


def _compile(src, name):
    d = tempfile.mkdtemp()
    f, lib = os.path.join(d, name + ".hip"), os.path.join(d, "lib" + name + ".so")
    open(f, "w").write(src)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", f, "-o", lib])
    return ctypes.CDLL(lib)


def _problem(Tp=751, N=32, H=1024, D=2, kind="gru"):
    from deepspeech.pytorch_amd import ops
    G = ops.GATES[kind]
    torch.manual_seed(0)
    GI = torch.randn(Tp * N, D * G * H, device=DEV).to(torch.bfloat16)
    Whh = ((torch.rand(D, G * H, H, device=DEV) * 2 - 1) / H ** 0.5).to(torch.bfloat16)
    WhhT = Whh.transpose(1, 2).contiguous()
    bhh = torch.zeros(D, G * H, device=DEV)
    lens = torch.from_numpy(np.sort(np.random.RandomState(0).randint(min(600, Tp // 2), Tp + 1, N))[::-1].copy().astype(np.int32)).to(DEV)
    lens[0] = Tp
    dout = torch.randn(Tp, N, H, device=DEV).to(torch.bfloat16)
    return dict(kind=kind, D=D, N=N, H=H, Tp=Tp, GI=GI, Whh=Whh, WhhT=WhhT, bhh=bhh, lens=lens, dout=dout)


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    if dist.is_initialized():
        dist.destroy_process_group()


def test_allreduce_on_a_side_stream_beside_a_full_bptt_sweep(one_rank_group):
    import torch.distributed as dist
    from deepspeech.pytorch_amd import ops
    if not ops.use_persistent("gru", torch.bfloat16, 2, 32, 1024):
        pytest.skip("persistent sweeps need all 256 CUs")
    p = _problem()
    hext, Sv, hn, cn = ops.rnn_fwd(p["kind"], p["GI"], p["Whh"], p["bhh"], p["lens"], p["D"], p["N"], p["H"], p["Tp"])
    side = torch.cuda.Stream()
    payload = torch.randn(16 << 20, device=DEV)               # 64 MB fp32: one recurrent layer's gradients

    def sweep():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rg = ops.rnn_bwd(p["kind"], p["dout"], p["WhhT"], hext, Sv, p["lens"], p["D"], p["N"], p["H"], p["Tp"])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1), rg

    sweep()
    alone, ref = min((sweep() for _ in range(3)), key=lambda r: r[0])
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(8):                                 # several collectives in flight across the whole sweep
                dist.all_reduce(payload, op=dist.ReduceOp.SUM)
        t, rg = sweep()
        times.append(t)
        assert torch.equal(rg.dGI, ref.dGI) and torch.equal(rg.dQ, ref.dQ)
    ops.check_persistent_kernels()                             # no workgroup gave up
    print("BPTT sweep alone %.3f ms, beside 8 x 64 MB all-reduce on a 1-rank RCCL group %.3f ms (x%.2f)" % (alone, min(times), min(times) / alone))
    assert min(times) < 3.0 * alone


def test_sweep_that_cannot_get_its_cus_times_out_loudly_and_recovers():
    from deepspeech.pytorch_amd import _lib, ops
    if not ops.use_persistent("gru", torch.bfloat16, 2, 32, 1024):
        pytest.skip("persistent sweeps need all 256 CUs")
    L = _compile(SQUAT_SRC, "squat")
    L.run_squat.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    p = _problem(Tp=64)
    ref = ops.rnn_fwd(p["kind"], p["GI"], p["Whh"], p["bhh"], p["lens"], p["D"], p["N"], p["H"], p["Tp"])[0].clone()
    ops.check_persistent_kernels()
    side = torch.cuda.Stream()
    release = torch.zeros(1, dtype=torch.int32, device=DEV)
    sink = torch.zeros(256, dtype=torch.int32, device=DEV)
    lib = _lib.load()
    old = lib.ds2_rnn_persist_set_spin_limit(20000)            # ~20 k polls instead of seconds
    try:
        torch.cuda.synchronize()
        # 128 squatters, one per CU they land on (160 KB of LDS each), for at most 3 s of wall clock (100 MHz ticks)
        rc = L.run_squat(128, ctypes.c_ulonglong(300_000_000), ctypes.c_void_p(release.data_ptr()), ctypes.c_void_p(sink.data_ptr()),
                         ctypes.c_void_p(side.cuda_stream))
        assert rc == 0
        hext = ops.rnn_fwd(p["kind"], p["GI"], p["Whh"], p["bhh"], p["lens"], p["D"], p["N"], p["H"], p["Tp"])[0]
        ops.poll_persistent_error(torch.device(DEV, 0))        # enqueues the asynchronous look at the error word
        time.sleep(1.0)                                        # the resident half of the sweep has spent its start-up budget by now (1e6
                                                               # polls = ~0.3 s, measured by the next test: 0.3 s of sleep was a coin toss;
                                                               # the squatters stay for up to 3 s)
        with torch.cuda.stream(torch.cuda.Stream()):
            release.fill_(1)                                   # the squatters leave; the rest of the sweep's workgroups start, see the
        torch.cuda.synchronize()                               # launch's error word and end at once
        with pytest.raises(_lib.Ds2HipError, match="co-resident"):   # the message names the cause: the start-up handshake never completed
            ops.poll_persistent_error(torch.device(DEV, 0))    # the copy has landed: raises now, without a host synchronisation
            ops.poll_persistent_error(torch.device(DEV, 0))
        assert not torch.isfinite(hext.float()).all()          # poisoned, never silently wrong
        with pytest.raises(_lib.Ds2HipError, match="another kernel holds compute units"):
            ops.check_persistent_kernels()
    finally:
        lib.ds2_rnn_persist_set_spin_limit(old)
        release.fill_(1)
        torch.cuda.synchronize()
    # the error word is sticky by design; a fresh word + a clean sweep: same result as before
    ops._PERSIST_ERR.clear()
    ops._ERR_MIRROR.clear()
    again = ops.rnn_fwd(p["kind"], p["GI"], p["Whh"], p["bhh"], p["lens"], p["D"], p["N"], p["H"], p["Tp"])[0]
    ops.check_persistent_kernels()
    assert torch.equal(again, ref)


def test_sweep_without_its_cus_gives_up_within_a_fraction_of_a_second_by_itself():
    """Round 5: the start-up handshake has its own budget (~0.1 s instead of the seconds a mid-sweep wait may take) -- with NO test hook
    lowering the spin limit, a sweep whose workgroups cannot all become resident ends by itself while the squatters are still there,
    and the error names the cause."""
    from deepspeech.pytorch_amd import _lib, ops
    if not ops.use_persistent("gru", torch.bfloat16, 2, 32, 1024):
        pytest.skip("persistent sweeps need all 256 CUs")
    L = _compile(SQUAT_SRC, "squat")
    L.run_squat.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    p = _problem(Tp=64)
    ops.check_persistent_kernels()
    ops._PERSIST_ERR.clear()
    ops._ERR_MIRROR.clear()
    side = torch.cuda.Stream()
    release = torch.zeros(1, dtype=torch.int32, device=DEV)
    sink = torch.zeros(256, dtype=torch.int32, device=DEV)
    try:
        torch.cuda.synchronize()
        rc = L.run_squat(128, ctypes.c_ulonglong(300_000_000), ctypes.c_void_p(release.data_ptr()), ctypes.c_void_p(sink.data_ptr()),
                         ctypes.c_void_p(side.cuda_stream))                     # squatters stay for up to 3 s unless released
        assert rc == 0
        t0 = time.perf_counter()
        hext = ops.rnn_fwd(p["kind"], p["GI"], p["Whh"], p["bhh"], p["lens"], p["D"], p["N"], p["H"], p["Tp"])[0]
        err = ops._persist_err(torch.device(DEV, 0))
        done = torch.cuda.Event()
        done.record()                                                           # behind the sweep on the caller's stream
        gave_up_after = None
        while time.perf_counter() - t0 < 2.0:
            time.sleep(0.02)
            # the resident workgroups have raised the word although the launch cannot end before the squatters leave
            with torch.cuda.stream(side):
                pass
            probe = torch.empty(1, dtype=torch.int32).pin_memory()
            with torch.cuda.stream(torch.cuda.Stream()):
                probe.copy_(err[:1], non_blocking=True)
                torch.cuda.current_stream().synchronize()
            if int(probe[0]) != 0:
                gave_up_after = time.perf_counter() - t0
                break
        assert gave_up_after is not None and gave_up_after < 1.0, gave_up_after
        assert int(probe[0]) == 2
    finally:
        with torch.cuda.stream(torch.cuda.Stream()):
            release.fill_(1)
        torch.cuda.synchronize()
    with pytest.raises(_lib.Ds2HipError, match="co-resident"):
        ops.check_persistent_kernels()
    assert not torch.isfinite(hext.float()).all()
    print("a sweep with half of the CUs taken gave up by itself after %.2f s" % gave_up_after)
    ops._PERSIST_ERR.clear()
    ops._ERR_MIRROR.clear()
