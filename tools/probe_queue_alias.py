"""Round 6: root cause of the round-5 red gate (tests/test_gpu_persist_safety.py failing only in full-suite order).

Hypothesis: the squatter's side stream and the sweep's stream shared a HARDWARE queue, so the two never overlapped.
1. map which of torch's pool streams run concurrently with the null stream (tests/squat.runs_concurrently);
2. a sweep beside an in-process squatter on (a) an independent stream, (b) an aliased stream: handshake slots seen while the
   squatters hold their CUs, error word afterwards;
3. the same with the squatter in a SECOND PROCESS.
Prints a table; run through gpurun, output kept under profiles/."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import squat  # noqa: E402
from test_gpu_persist_safety import _problem  # noqa: E402
from deepspeech.pytorch_amd import _lib, ops  # noqa: E402


def sweep_beside(sq_factory, copy_stream, label):
    p = _problem(Tp=64)
    ops._PERSIST_ERR.clear()
    ops._ERR_MIRROR.clear()
    sq = sq_factory()
    n = sq.wait_started()
    t0 = time.perf_counter()
    with ops.persist_options(spin_limit=20000):
        hext = ops.rnn_fwd(p["kind"], p["GI"], p["Whh"], p["bhh"], p["lens"], p["D"], p["N"], p["H"], p["Tp"])[0]
    ws = ops.LAST_PERSIST_WS
    time.sleep(0.05)
    slots = squat.sweep_handshake_slots(ws, copy_stream)
    still = sq.running()
    sq.release()
    torch.cuda.synchronize()
    code = int(ops._persist_err(torch.device("cuda", 0))[0].item())
    print("%-44s squatters started %3d, sweep workgroups resident beside them %4d (squatters still there: %s), error word %d, "
          "outputs finite %s, %.2f s" % (label, n, slots, still, code, bool(torch.isfinite(hext.float()).all()), time.perf_counter() - t0))
    ops._PERSIST_ERR.clear()
    ops._ERR_MIRROR.clear()


def main():
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(unset: 4)"))
    x = torch.zeros(1, device="cuda")
    pool = [torch.cuda.Stream() for _ in range(36)]
    conc = [squat.runs_concurrently(s) for s in pool]
    print("pool stream k runs concurrently with the null stream:")
    print("  " + "".join("Y" if c else "-" for c in conc), " (%d of %d aliased with the null stream's hardware queue)" % (conc.count(False), len(conc)))
    handles = {}
    for k, s in enumerate(pool):
        handles.setdefault(s.cuda_stream, []).append(k)
    print("  distinct hipStream_t handles in 36 draws: %d" % len(handles))
    ind = [s for s, c in zip(pool, conc) if c]
    ali = [s for s, c in zip(pool, conc) if not c]
    (side, copy_s), tried = squat.independent_streams(2)
    print("independent_streams(2): found after %d draws" % tried)
    sweep_beside(lambda: squat.Squatter.in_process(side), copy_s, "in-process squatter, independent stream")
    if ali:
        sweep_beside(lambda: squat.Squatter.in_process(ali[0], max_s=1.0), copy_s, "in-process squatter, ALIASED stream (1 s)")
    sweep_beside(lambda: squat.Squatter.second_process(), copy_s, "squatter in a second process")
    sweep_beside(lambda: squat.Squatter.second_process(), copy_s, "squatter in a second process (again)")
    # clean sweep afterwards
    p = _problem(Tp=64)
    ops.rnn_fwd(p["kind"], p["GI"], p["Whh"], p["bhh"], p["lens"], p["D"], p["N"], p["H"], p["Tp"])
    ops.check_persistent_kernels()
    print("clean sweep afterwards: ok")


if __name__ == "__main__":
    main()
