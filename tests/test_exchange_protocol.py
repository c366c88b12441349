"""Model check of the tag-free four-slot exchange of the persistent recurrent kernels (csrc/ds2_rnn_persist_impl.h,
gather_mma_tf; csrc/ds2_rnn_persist2_impl.h, gather_mma2f) -- host-only, no GPU.

Protocol of one workgroup at step s: (1) gather: poll every peer's dword of slot (s-1) & 3 until none is the sentinel;
(2) publish the own dword of slot s & 3; (3) re-arm the own dword of slot (s+2) & 3 with the sentinel.  The hardware assumption
the kernels rely on, and the only ordering this model grants: a workgroup's stores of step s are all visible before any of its
stores of step s+1 is issued (its gather of step s+1 waits on vmcnt, which retires in order) -- WITHIN a step the two stores may
become visible in either order and arbitrarily late.  Checked: a gather never returns anything but the value the peer published
for exactly that step (no stale data of step s-4 in the re-used slot, no lost step), under random schedules; and the control
experiment: with three slots the same schedules DO go wrong (a re-arm overtakes data a slow peer still needs: the peer reads the
sentinel forever, or the next-but-two step's data), i.e. the model can tell the difference.

Round 4 adds two things to the model (csrc/ds2_rnn_persist3_impl.h).  SET SCHEDULES: a sample set executes only the steps
[lo, hi) in which one of its clips is inside its sequence -- it joins the protocol at step lo (all four slots armed by the launch's
reset; with an initial state every workgroup first publishes "step lo - 1" into slot (lo + 3) & 3) and simply stops after hi - 1:
checked for every residue of lo mod 4.  And the control experiment that was a real bug for an afternoon: workgroups that gather
only while one of their clips is inside its sequence stop WAITING once all clips have ended -- the wait is what keeps a fast
workgroup from re-arming a slot a slow peer has not read yet; the model shows the peer reading the sentinel forever (the
kernel's time-out) or another step's data."""
import random

import pytest

SENT = None


def run(n_wg, n_steps, n_slots, seed, rearm_ahead=2, first_step=0, initial_state=False, gather_until=None):
    """Steps first_step .. first_step + n_steps - 1 of one sample set.  initial_state: every workgroup publishes "step first_step - 1"
    before its first step and the first step gathers it.  gather_until = E: the (wrong) variant in which a workgroup gathers -- and
    therefore waits for its peers -- only at steps <= E (all clips of the group end at E) and free-runs afterwards."""
    rng = random.Random(seed)
    mem = [[SENT] * n_wg for _ in range(n_slots)]          # mem[slot][producer]
    pending = [[] for _ in range(n_wg)]                     # issued, not yet visible stores of the producer's CURRENT step
    step = [first_step] * n_wg                              # step each workgroup is in
    n_steps = first_step + n_steps
    if initial_state:
        for p in range(n_wg):
            mem[(first_step - 1) % n_slots][p] = (p, first_step - 1)
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
            if (s == first_step and not initial_state) or (gather_until is not None and s > gather_until):
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


@pytest.mark.parametrize("lo", [1, 2, 3, 4, 7])
@pytest.mark.parametrize("initial_state", [False, True])
def test_a_set_may_join_the_protocol_at_any_step(lo, initial_state):
    """Set schedules: the set's first executed step is lo, whatever its residue mod 4; every slot is armed when it joins."""
    for seed in range(8):
        assert run(n_wg=6, n_steps=30, n_slots=4, seed=seed, first_step=lo, initial_state=initial_state) == 0


def test_gathering_only_while_a_clip_is_inside_its_sequence_breaks_the_lock_step():
    """All clips of the group end at step 20 of 40: from step 21 on nobody waits for anybody.  A workgroup that is still at step 20
    needs its peers' step-19 words; a peer that stopped waiting is at step 22 already and has re-armed that very slot: stuck
    (1000 = the kernels' time-out) or stale data in some schedules -- while the same schedules are clean when every workgroup
    gathers at every executed step (first test of this file)."""
    assert sum(run(n_wg=6, n_steps=40, n_slots=4, seed=seed, gather_until=20) for seed in range(20)) > 0
