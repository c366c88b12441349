"""Executable statement of the index rules of the round-5 conv2 kernels (csrc/ds2_conv.hip: k_conv2_wgrad_bf16d, k_conv_rtap) --
host-only, no GPU.  The kernels are held to the oracle and, at the old tiling, bit for bit to their predecessors on the device
(tests/test_gpu_kernels.py::test_conv2_fwd_dgrad_wgrad, tools/bench_conv2_wgrad.py --compare); this file pins the RULES they implement:

weight gradient (reference: the autograd of nn.Conv2d(32, 32, (21, 11), stride (2, 1), padding (10, 5)), model.py:161)
  * work map: workgroup L runs unit u = 64 (L % 8) + L / 8 = (position split u / 6, row group u % 6); every (split, row group) exactly
    once on the 512 workgroup slots, the six row groups of a split on ONE XCD (L % 8) unless the split straddles a block of 64;
  * the cursor (sample n, j, tile t) of a split advances by additions and carries: identical to dividing the item number;
    over all splits every live item (input row 2 j - 10 + kf0 inside [0, 81)) is visited exactly once, dead ones never;
  * row groups {0,2,4,6} {8..14} {16,18,20,-} {1..7} {9..15} {17,19,-,-} x taps kt = wave + 4 i cover the 21 x 11 taps exactly once;
  * a transposing read of (k-step ks, tap window i, half h) of wave w covers X rows p + kt + 3 of the staged tile (position t0 - 8
    first): never beyond its 16 (KSN + 1) rows.

forward / data gradient (k_conv_rtap)
  * LDS-DMA pieces: wave w stages the patch rows w, w + 4, ... in three chunks of 16 positions: every (row, chunk) exactly once;
  * swizzle: the 16-byte chunk c of staged position p lives at slot c ^ ((p >> 2) & 3) -- the DMA lane l of a chunk fetches channel
    chunk (l & 3) ^ (l >> 4), the reader of position p = li + kt asks for slot (2 half + lq) ^ ((p >> 2) & 3): the same bytes; and each
    of ds_read_b128's four lane groups touches sixteen DIFFERENT 16-byte slots of the 256-byte bank row (no bank conflict) for every kt."""
import itertools

import pytest

F1, F2, K2F, K2T = 81, 41, 21, 11
R = 4
NJ = F2 + R - 1
UNITS_PER_XCD = 64
NSPLIT = 85


def kf0_of(rg):
    return 8 * rg if rg < 3 else 1 + 8 * (rg - 3)


def test_wgrad_work_map_covers_every_unit_once_and_keeps_a_split_on_one_xcd():
    seen = {}
    for L in range(8 * UNITS_PER_XCD):
        unit = (L & 7) * UNITS_PER_XCD + (L >> 3)
        if unit >= NSPLIT * 6:
            continue
        split, rg = divmod(unit, 6)
        assert (split, rg) not in seen
        seen[(split, rg)] = L & 7
    assert len(seen) == NSPLIT * 6
    straddling = 0
    for split in range(NSPLIT):
        xcds = {seen[(split, rg)] for rg in range(6)}
        assert len(xcds) <= 2
        straddling += len(xcds) == 2
    assert straddling <= 7                      # only the splits that cross a block of 64 units


def test_wgrad_row_groups_and_taps_cover_the_kernel_once():
    hit = {}
    for rg in range(6):
        for r in range(R):
            kf = kf0_of(rg) + 2 * r
            for wave in range(4):
                for i in range(3):
                    kt = wave + 4 * i
                    if kf < K2F and kt < K2T:          # what the kernel writes out
                        assert (kf, kt) not in hit
                        hit[(kf, kt)] = (rg, r, wave, i)
    assert len(hit) == K2F * K2T


@pytest.mark.parametrize("N,Tp,KSN", [(32, 751, 7), (64, 751, 7), (3, 77, 7), (2, 1501, 8), (1, 16, 7), (5, 113, 8)])
def test_wgrad_cursor_equals_division_and_visits_every_live_item_once(N, Tp, KSN):
    TB = 16 * KSN
    ntiles = -(-Tp // TB)
    nwork = N * NJ * ntiles
    d_t, d_q = NSPLIT % ntiles, NSPLIT // ntiles
    d_j, d_n = d_q % NJ, d_q // NJ
    for rg in range(6):
        kf0 = kf0_of(rg)
        visited = set()
        for split in range(NSPLIT):
            q0 = split // ntiles
            t, n, j = split - q0 * ntiles, q0 // NJ, q0 % NJ
            wk = split
            while n < N:
                assert wk < nwork and (n, j, t) == ((wk // ntiles) // NJ, (wk // ntiles) % NJ, wk % ntiles)
                fi = 2 * j - 10 + kf0
                if 0 <= fi < F1:
                    assert (n, j, t) not in visited
                    visited.add((n, j, t))
                # advance (the kernel's lambda)
                t += d_t
                if t >= ntiles:
                    t -= ntiles
                    j += 1
                j += d_j
                if j >= NJ:
                    j -= NJ
                    n += 1
                n += d_n
                wk += NSPLIT
            assert wk >= nwork
        live = {(n, j, t) for n in range(N) for j in range(NJ) for t in range(ntiles) if 0 <= 2 * j - 10 + kf0 < F1}
        assert visited == live


@pytest.mark.parametrize("KSN", [7, 8])
def test_wgrad_tap_windows_stay_inside_the_staged_x_tile(KSN):
    rows = 16 * (KSN + 1)
    worst = 0
    for wave, i, ks, lq, q, h in itertools.product(range(4), range(3), range(KSN), range(2), range(4), range(2)):
        # lane group gq = 2 lq + (channel half), lane i of the group passes position row q = i / 4; the second read adds 4 rows
        row = (wave + 3) + 4 * i + 16 * ks + 8 * lq + q + 4 * h
        worst = max(worst, row)
        # the operand's k-th position p = 16 ks + 8 lq + 4 h + q pairs dY[p] with X[p + kt - 5]; the tile starts at t0 - 8
        assert row == (16 * ks + 8 * lq + 4 * h + q) + (wave + 4 * i) - 5 + 8
    assert worst < rows


@pytest.mark.parametrize("KF", [10, 11])
def test_rtap_dma_pieces_cover_every_patch_chunk_once(KF):
    PR = 3 + KF
    seen = set()
    for wave in range(4):
        for piece in range(12):
            pr, k = wave + 4 * (piece // 3), piece % 3
            if pr < PR:
                assert (pr, k) not in seen
                seen.add((pr, k))
    assert seen == {(pr, k) for pr in range(PR) for k in range(3)}


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def test_rtap_swizzle_reader_finds_the_dma_writers_bytes_and_is_conflict_free():
    # writer: DMA chunk k of a patch row, lane l -> LDS bytes (16 k + l / 4) * 64 + (l % 4) * 16 hold channel chunk (l & 3) ^ (l >> 4)
    where = {}
    for k in range(3):
        for lane in range(64):
            p = 16 * k + (lane >> 2)
            where[(p, (lane & 3) ^ (lane >> 4))] = p * 64 + (lane & 3) * 16
    for kt in range(K2T):
        for half in range(2):
            for lq in range(2):
                addr = {}
                for li in range(32):
                    p = li + kt
                    a = p * 64 + (((half * 2 + lq) ^ ((p >> 2) & 3)) << 4)
                    assert where[(p, half * 2 + lq)] == a            # same bytes
                    addr[li] = a
                for grp in B128_GROUPS:                              # lanes 32-63 (lq = 1) form the same groups shifted by 32
                    slots = {(addr[li] % 256) // 16 for li in grp}
                    assert len(slots) == 16, (kt, half, lq, sorted(slots))


def test_wgrad_decomposition_in_numpy_equals_the_oracle():
    """The kernel's decomposition restated in numpy on a small batch (full 81 -> 41 frequency geometry, short clips): tiles of 16 KSN
    positions staged as they lie in memory with ZEROS outside the clip / for rows that do not exist (what the LDS-DMA delivers), the X
    tile starting at position t0 - 8, work items (n, j, tile) with input row 2 j - 10 + kf0 against the output rows j - r, tap kt of
    wave w and slot i read at row offset p + kt + 3 -- summed over every split -- is the weight gradient of the oracle
    (oracle/ds2_oracle.py:conv2d_bwd = autograd of model.py:161)."""
    import numpy as np
    from oracle import ds2_oracle as O

    rs = np.random.RandomState(5)
    N, Tp, KSN = 2, 37, 7
    TB = 16 * KSN
    a1 = rs.standard_normal((N, 32, F1, Tp))
    dy = rs.standard_normal((N, 32, F2, Tp))
    _, dw_ref, _ = O.conv2d_bwd(a1, np.zeros((32, 32, K2F, K2T)), dy, (2, 1), (10, 5), need_dx=False)      # [co][ci][kf][kt]
    X = a1.transpose(0, 2, 3, 1)          # NFTC [n][f][t][ci]
    DY = dy.transpose(0, 2, 3, 1)         # [n][fo][t][co]
    ntiles = -(-Tp // TB)
    dw = np.zeros((K2F, K2T, 32, 32))     # [kf][kt][co][ci]
    for rg in range(6):
        kf0 = kf0_of(rg)
        for n, j, t in itertools.product(range(N), range(NJ), range(ntiles)):
            fi = 2 * j - 10 + kf0
            if not 0 <= fi < F1:
                continue
            t0 = t * TB
            xt = np.zeros((16 * (KSN + 1), 32))                       # staged X tile: positions t0 - 8 ...
            for q in range(xt.shape[0]):
                pos = t0 - 8 + q
                if 0 <= pos < Tp:
                    xt[q] = X[n, fi, pos]
            for r in range(R):
                fo, kf = j - r, kf0 + 2 * r
                dyt = np.zeros((TB, 32))                              # staged dY^T tile of wave r
                if 0 <= fo < F2 and kf < K2F:
                    m = min(TB, Tp - t0)
                    dyt[:m] = DY[n, fo, t0:t0 + m]
                if kf >= K2F:
                    continue                                          # (the kernel multiplies, and never writes the result out)
                for wave, i in itertools.product(range(4), range(3)):
                    kt = wave + 4 * i
                    if kt >= K2T:
                        continue
                    win = xt[kt + 3:kt + 3 + TB]                      # row p + kt + 3 for p = 0 .. TB - 1
                    dw[kf, kt] += dyt.T @ win
    got = dw.transpose(2, 3, 0, 1)        # [co][ci][kf][kt]
    assert np.abs(got - dw_ref).max() <= 1e-9 * np.abs(dw_ref).max()
