"""Safety of the persistent recurrent sweeps next to other work on the chip (what a data-parallel run does to them: RCCL kernels
on their own stream while the sweeps own the CUs).

* a >= 64 MB all-reduce on a 1-rank RCCL group, enqueued on a side stream, runs concurrently with a full cfg3 BPTT sweep: the sweep
  finishes without a time-out, bit-identical to the sweep alone; the slow-down is printed;
* FAULT INJECTION (tests/squat.py): squatter workgroups hold half of the chip's CUs (all of their LDS) while a sweep starts.
  The pre-condition is OBSERVED, not assumed: the squatters count themselves in, the sweep is launched only when all of them run,
  and the test reads how many of the sweep's workgroups have signed the start-up handshake WHILE the squatters are still there --
  0 means the squatters' stream shares a hardware queue with the sweep's (the two were serialised: round 5's red gate), all of them
  means the squatters did not hold their CUs; either fails with that message instead of a misleading one.
    - single-process budget: the resident half of the sweep gives up after its start-up budget by itself, the launch ENDS, the
      device error word says 2 ("never co-resident"), the outputs are NaN-poisoned, ops.poll_persistent_error -- what
      DeepSpeech.training_step calls once per step -- raises at the latest one call later; afterwards a clean sweep works again;
    - the same with the squatters in a SECOND PROCESS (the scenario the error message names);
    - DATA-PARALLEL budget: the squatters hold the CUs for a second -- an RCCL collective waiting for a late peer rank -- and the
      sweep WAITS, then completes bit-identical with no error."""
import os
import socket
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


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


def _sweep(p):
    from deepspeech.pytorch_amd import ops
    return ops.rnn_fwd(p["kind"], p["GI"], p["Whh"], p["bhh"], p["lens"], p["D"], p["N"], p["H"], p["Tp"])[0]


def _fresh_error_word():
    from deepspeech.pytorch_amd import ops
    ops._PERSIST_ERR.clear()
    ops._ERR_MIRROR.clear()


def _need_full_chip():
    from deepspeech.pytorch_amd import ops
    if not ops.use_persistent("gru", torch.bfloat16, 2, 32, 1024):
        pytest.skip("persistent sweeps need all 256 CUs")


def _launch_beside_squatters(sq, p, copy_stream, **opts):
    """Launches a sweep once all squatters run; returns (outputs, scratch, handshake slots signed while the squatters were there)."""
    from deepspeech.pytorch_amd import ops
    import squat
    n = sq.wait_started()
    assert n == sq.blocks, "only %d of %d squatter workgroups started: the fault was not injected" % (n, sq.blocks)
    with ops.persist_options(**opts):
        hext = _sweep(p)
    ws = ops.LAST_PERSIST_WS
    time.sleep(0.05)
    slots = squat.sweep_handshake_slots(ws, copy_stream)
    assert sq.running(), "the squatters left before the sweep was observed"
    assert slots != 0, ("none of the sweep's workgroups became resident while the squatters ran: the two were SERIALISED (the squatters' "
                        "stream shares a hardware queue with the sweep's) -- the test did not create the condition it is about")
    assert 0 < slots < 256, "%d of 256 sweep workgroups resident beside %d squatters: the squatters do not hold their CUs" % (slots, n)
    return hext, ws, slots


@pytest.mark.parametrize("where", ["side_stream", "second_process"])
def test_sweep_that_cannot_get_its_cus_times_out_loudly_and_recovers(where):
    from deepspeech.pytorch_amd import _lib, ops
    import squat
    _need_full_chip()
    p = _problem(Tp=64)
    ref = _sweep(p).clone()
    ops.check_persistent_kernels()
    _fresh_error_word()
    (side, copy_s), tried = squat.independent_streams(2)
    sq = squat.Squatter.in_process(side) if where == "side_stream" else squat.Squatter.second_process()
    try:
        t0 = time.perf_counter()
        hext, ws, slots = _launch_beside_squatters(sq, p, copy_s)           # the default single-process budget: 0.3 s
        ops.poll_persistent_error(torch.device(DEV, 0))                    # enqueues the asynchronous look at the error word
        # the resident workgroups raise the word by themselves although the launch cannot end before the squatters leave
        err = ops._persist_err(torch.device(DEV, 0))
        probe = torch.zeros(1, dtype=torch.int32).pin_memory()
        gave_up_after = None
        while time.perf_counter() - t0 < 3.0:
            with torch.cuda.stream(copy_s):
                probe.copy_(err[:1], non_blocking=True)
            copy_s.synchronize()
            if int(probe[0]) != 0:
                gave_up_after = time.perf_counter() - t0
                break
            time.sleep(0.02)
        assert sq.running()
        assert gave_up_after is not None and gave_up_after < 1.5, gave_up_after
        assert int(probe[0]) == 2
    finally:
        sq.release()                                   # the squatters leave; the rest of the sweep's workgroups start, see the
        torch.cuda.synchronize()                       # launch's error word and end at once
    with pytest.raises(_lib.Ds2HipError, match="co-resident"):   # the message names the cause: the start-up wait never completed
        ops.poll_persistent_error(torch.device(DEV, 0))    # the copy has landed: raises now, without a host synchronisation
        ops.poll_persistent_error(torch.device(DEV, 0))
    assert not torch.isfinite(hext.float()).all()          # poisoned, never silently wrong
    with pytest.raises(_lib.Ds2HipError, match="another kernel holds compute units"):
        ops.check_persistent_kernels()
    print("%s: %d of 256 sweep workgroups resident beside 128 squatters (side stream found after %d draws); gave up by itself after %.2f s"
          % (where, slots, tried, gave_up_after))
    # the error word is sticky by design; a fresh word + a clean sweep: same result as before
    _fresh_error_word()
    again = _sweep(p)
    ops.check_persistent_kernels()
    assert torch.equal(again, ref)


@pytest.mark.parametrize("shape", [("gru", 2, 32, 1024), ("lstm", 2, 64, 1280), ("gru", 2, 8, 800)])
def test_every_kernel_family_gives_up_at_start_up_with_code_2(shape):
    """The tuned kernels and the XCD-local general kernels find out in their XCC-id handshake; the general kernels whose groups span
    XCDs (H = 1280) and the round-2 general kernels (fp32 / H = 800 ...) have no handshake: one arrival word per launch.  A short
    per-launch budget (startup_ms = 30) keeps this quick; the squatters are released by the test."""
    from deepspeech.pytorch_amd import _lib, ops
    import squat
    kind, D, N, H = shape
    dtype = torch.float32 if H == 800 else torch.bfloat16
    if not ops.use_persistent(kind, dtype, D, N, H):
        pytest.skip("persistent sweeps need all 256 CUs")
    p = _problem(Tp=16, N=N, H=H, D=D, kind=kind)
    if dtype == torch.float32:
        for k in ("GI", "Whh", "WhhT", "dout"):
            p[k] = p[k].float()
    ref = _sweep(p).clone()
    ops.check_persistent_kernels()
    _fresh_error_word()
    (side, copy_s), _ = squat.independent_streams(2)
    sq = squat.Squatter.in_process(side, blocks=160)
    try:
        n = sq.wait_started()
        assert n == 160, n
        with ops.persist_options(startup_ms=30):
            hext = _sweep(p)
        time.sleep(0.3)
        assert sq.running()
    finally:
        sq.release()
        torch.cuda.synchronize()
    code = int(ops._persist_err(torch.device(DEV, 0))[0].item())
    assert code == 2, "family %d: error word %d" % (ops.persist_kind(dtype, kind, D, N, H), code)
    assert not torch.isfinite(hext.float()).all()
    _fresh_error_word()
    assert torch.equal(_sweep(p), ref)
    ops.check_persistent_kernels()


@pytest.mark.parametrize("shape", [("gru", 2, 32, 1024), ("lstm", 2, 64, 1280)])
def test_data_parallel_budget_sweep_waits_for_its_cus_and_completes(shape):
    """Under data parallelism (ops.persist_startup_ms: the process group's time-out) a sweep that finds half of the chip taken --
    an RCCL collective waiting for a late peer rank (loader/data_loader.py:320-360 hands ranks unequal batches) -- WAITS for its
    CUs as a stock kernel would queue, then runs: bit-identical, no error."""
    from deepspeech.pytorch_amd import ops
    import squat
    kind, D, N, H = shape
    if not ops.use_persistent(kind, torch.bfloat16, D, N, H):
        pytest.skip("persistent sweeps need all 256 CUs")
    p = _problem(Tp=64, N=N, H=H, D=D, kind=kind)
    ref = _sweep(p).clone()
    ops.check_persistent_kernels()
    _fresh_error_word()
    (side, copy_s), _ = squat.independent_streams(2)
    sq = squat.Squatter.in_process(side, max_s=10.0)
    try:
        n = sq.wait_started()
        assert n == sq.blocks, n
        t0 = time.perf_counter()
        ops.FORCE_DATA_PARALLEL_BUDGET[0] = True           # what dist.wrap_data_parallel does
        try:
            assert ops.persist_startup_ms() >= 30_000
            hext = _sweep(p)
        finally:
            ops.FORCE_DATA_PARALLEL_BUDGET[0] = False
        done = torch.cuda.Event()
        done.record()
        time.sleep(1.0)                                     # the "late peer": three start-up budgets of a single-process run
        assert sq.running() and not done.query(), "the sweep did not have to wait: the squatters were not in its way"
        sq.release()
        torch.cuda.synchronize()
        waited = time.perf_counter() - t0
    finally:
        sq.release()
        torch.cuda.synchronize()
    ops.check_persistent_kernels()                          # no error word
    assert torch.equal(hext, ref)
    print("%s: the sweep waited %.2f s for its CUs under the data-parallel budget and completed bit-identical" % (shape, waited))
