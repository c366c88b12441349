"""Executable statement of the set schedule of the round-4 general sweeps (csrc/ds2_rnn_persist3_impl.h: sched3, the sample -> (set,
row) map of k_rnn_persist3_fwd / _bwd, plan3h in ds2_rnn_persist.hip) -- host-only, no GPU.  The kernels are tested against the
oracle on the device (tests/test_gpu_kernels.py::test_rnn_persist3_set_schedules); this file pins the RULE they implement, so that
a change of the rule is a deliberate one:

  * group g of a direction owns the clips n = slice + gpd * i, i < Ns;
  * one set: rows i; two sets: set 0 = the first min(16, Ns) of them (a full MFMA m-tile), set 1 the rest;
  * a set executes step s of a sweep iff s in [lo, hi): lo = 0, hi = max length for a sweep that visits t = s; lo = T' - max length,
    hi = T' for one that visits t = T' - 1 - s (max over the set's clips, lengths clipped to T').

Properties checked over random batches: every clip sits in exactly one (set, row) with row < 16; outside [lo, hi) NO clip of the set
is inside its sequence (so leaving the half-step out changes nothing: reference semantics of pack_padded_sequence, model.py:96); the
range is tight (its first / last step has an active clip); the union of the sets' ranges is contiguous and contains every active
(clip, step); with lengths sorted descending -- the loader's order, data_loader.py:249 -- set 1 never outlasts set 0."""
import numpy as np
import pytest


def plan(N, D, H):
    """plan3h: (clips per group slot gpd, groups NG, sets) or None when the shape is not covered (cf. ds2_rnn_persist.hip)."""
    P = H // 32
    slots = 8 * (32 // P) if P <= 32 else 256 // P
    gpd = min(slots // D, N)
    if gpd < 1:
        return None
    ns = -(-N // gpd)
    nset = 1 if ns <= 16 else 2 if ns <= 32 else 0
    return (gpd, gpd * D, nset) if nset else None


def sched(lens, slice_, gpd, N, Tp, nset, ascending):
    Ns = -(-(N - slice_) // gpd)
    RPS = Ns if nset == 1 else min(16, Ns)
    sets = []
    for q in range(nset):
        rows = max(0, min(RPS, Ns - q * RPS))
        clips = [slice_ + gpd * (q * RPS + r) for r in range(rows)]
        mx = max([min(int(lens[n]), Tp) for n in clips], default=0)
        lo, hi = (0, mx) if ascending else (Tp - mx, Tp)
        sets.append((clips, lo, hi))
    return sets


@pytest.mark.parametrize("seed", range(30))
def test_schedule_rule(seed):
    rs = np.random.RandomState(seed)
    H = int(rs.choice([512, 768, 800, 1024, 1280, 1536]))
    D = int(rs.choice([1, 2]))
    N = int(rs.randint(1, 129))
    pl = plan(N, D, H)
    if pl is None:
        pytest.skip("more than 32 clips per group: the round-2 kernels take it")
    gpd, NG, nset = pl
    Tp = int(rs.randint(5, 200))
    lens = rs.randint(1, Tp + 1, N)
    if seed % 3:
        lens = np.sort(lens)[::-1]
        lens[0] = Tp
    seen = set()
    for ascending in (True, False):
        for slice_ in range(gpd):
            sets = sched(lens, slice_, gpd, N, Tp, nset, ascending)
            t_of = (lambda s: s) if ascending else (lambda s: Tp - 1 - s)
            for q, (clips, lo, hi) in enumerate(sets):
                assert len(clips) <= 16
                for n in clips:
                    if ascending:
                        assert n not in seen
                        seen.add(n)
                active = [[t_of(s) < lens[n] for n in clips] for s in range(Tp)]
                for s in range(Tp):
                    if not lo <= s < hi:
                        assert not any(active[s]), (q, s, lo, hi)
                if clips:
                    assert any(active[lo]) and any(active[hi - 1])
            los, his = [x[1] for x in sets if x[0]], [x[2] for x in sets if x[0]]
            assert (min(los) == 0 and all(lo == 0 for lo in los)) if ascending else all(hi == Tp for hi in his)   # contiguous union
            if nset == 2 and seed % 3 and sets[1][0]:
                assert sets[1][2] - sets[1][1] <= sets[0][2] - sets[0][1]      # sorted batch: the second set is the shorter one
    assert seen == set(range(N))
