"""Model check of the tag-free four-slot exchange of the persistent recurrent kernels (csrc/ds2_rnn_persist_impl.h,
gather_mma_tf; csrc/ds2_rnn_persist2_impl.h, gather_mma2f) -- host-only, no GPU.

Protocol of one workgroup at step s: (1) gather: poll every peer's dword of slot (s-1) & 3 until none is the sentinel;
(2) publish the own dword of slot s & 3; (3) re-arm the own dword of slot (s+2) & 3 with the sentinel.  The hardware assumption
the kernels rely on, and the only ordering this model grants: a workgroup's stores of step s are all visible before any of its
stores of step s+1 is issued (its gather of step s+1 waits on vmcnt, which retires in order) -- WITHIN a step the two stores may
become visible in either order and arbitrarily late.  Checked: a gather never returns anything but the value the peer published
for exactly that step (no stale data of step s-4 in the re-used slot, no lost step), under random schedules; and the control
experiment: with three slots the same schedules DO go wrong (a re-arm overtakes data a slow peer still needs: the peer reads the
sentinel forever, or the next-but-two step's data), i.e. the model can tell the difference."""
import random

import pytest

SENT = None


def run(n_wg, n_steps, n_slots, seed, rearm_ahead=2):
    rng = random.Random(seed)
    mem = [[SENT] * n_wg for _ in range(n_slots)]          # mem[slot][producer]
    pending = [[] for _ in range(n_wg)]                     # issued, not yet visible stores of the producer's CURRENT step
    step = [0] * n_wg                                       # step each workgroup is in
    phase = [0] * n_wg                                      # 0 = gathering, 1 = stores issued (waiting to advance)
    got = [[False] * n_wg for _ in range(n_wg)]             # per consumer: which peers' dwords of this step it has seen
    stale = 0
    for _ in range(400 * n_wg * n_steps):                   # a correct run needs a few tens of scheduler picks per step
        if min(step) >= n_steps:
            return stale
        w = rng.randrange(n_wg)
        # a random pending store of ANY workgroup may become visible now (out of order within its step)
        if rng.random() < 0.5:
            cands = [p for p in range(n_wg) if pending[p]]
            if cands:
                p = rng.choice(cands)
                slot, val = pending[p].pop(rng.randrange(len(pending[p])))
                mem[slot][p] = val
        s = step[w]
        if s >= n_steps:
            continue
        if phase[w] == 0:
            if s == 0:
                done = True
            else:
                # poll a random subset of the peers' dwords of slot (s-1): what arrived counts, the rest is re-polled later
                for p in range(n_wg):
                    if not got[w][p] and rng.random() < 0.7:
                        v = mem[(s - 1) % n_slots][p]
                        if v is not SENT:
                            if v != (p, s - 1):
                                stale += 1
                            got[w][p] = True
                done = all(got[w])
            if done:
                # the gather has returned: every earlier store of this workgroup is visible by now (in-order vmcnt)
                for slot, val in pending[w]:
                    mem[slot][w] = val
                pending[w] = [(s % n_slots, (w, s)), ((s + rearm_ahead) % n_slots, SENT)]
                got[w] = [False] * n_wg
                phase[w] = 1
        else:
            step[w] = s + 1
            phase[w] = 0
    return stale + 1000                                     # stuck: some gather waits for data that was overwritten


@pytest.mark.parametrize("seed", range(20))
def test_four_slots_never_return_stale_data(seed):
    assert run(n_wg=6, n_steps=40, n_slots=4, seed=seed) == 0


def test_three_slots_go_wrong_under_the_same_model():
    # the re-armed slot (s+2) % 3 == (s-1) % 3 is the one the peers are still reading: a sentinel or the next data can overtake
    assert sum(run(n_wg=6, n_steps=40, n_slots=3, seed=seed) for seed in range(20)) > 0
