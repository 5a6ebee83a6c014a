"""CPU test of the step lists trunkw_kernel (two fused trunk layers per launch as Winograd F(2,3),
csrc/uva_wino.hip.h) walks, decoded through the host-only hook uva_debug_trunkw_schedule.

The kernel keeps state from step to step -- the producer's ring of six transformed input rows, of which a
step replaces four, and the ring of ten intermediate rows between producer and consumer -- so the test does
not only decode the entries: it WALKS every workgroup's list the way the kernel does (same ring-position
arithmetic), with row identities in place of pixel data, and checks that every stored output row was
computed from the right three intermediate rows, each of those from the right three input rows of the same
strip, that rows outside the plane are masked to zero, that every plane pixel is stored exactly once, that
the look-ahead entries are inert and that the work is spread evenly."""
import ctypes

import numpy as np
import pytest

SW = 30          # csrc/uva_wino.hip.h TW_SW
PAD = 2          # TW_PAD_STEPS
AROWS, BROWS = 6, 10


def schedule(uva, h, w, tile, border, grid=256):
    from upscale_video_amd import _lib
    L = _lib.load()
    need = ctypes.c_size_t()
    stride, nplanes, guard = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
    pinfo = np.zeros(64 * 4, np.int64)
    nsteps = np.zeros(grid, np.int32)
    L.uva_debug_trunkw_schedule(h, w, tile, border, grid, None, 0, need, None, stride, pinfo.ctypes.data, 64, nplanes, guard)
    words = np.zeros(need.value, np.uint32)
    rc = L.uva_debug_trunkw_schedule(h, w, tile, border, grid, words.ctypes.data, words.size, need, nsteps.ctypes.data,
                                     stride, pinfo.ctypes.data, 64, nplanes, guard)
    assert rc == 0, L.uva_last_error()
    planes = pinfo[:4 * nplanes.value].reshape(-1, 4)   # h, w, pitch, act_off
    return words.reshape(grid, stride.value, 8), nsteps, planes, guard.value


def locate(off, planes, guard, pi):
    """byte offset -> (row, column) in plane pi's padded array; rows / columns may lie above or left of it"""
    pix = (off - guard) // 128
    assert (off - guard) % 128 == 0 and off >= 0
    rel = pix - planes[pi, 3]
    pitch = int(planes[pi, 2])
    row = rel // pitch if rel >= 0 else -((-rel + pitch - 1) // pitch)
    col = int(rel - row * pitch)
    if col == pitch - 1:        # x0 = 0: the origin is one column left of the array row = the previous row's last column
        row, col = row + 1, -1
    return int(row), col


@pytest.mark.parametrize("h,w,tile,border", [
    (1080, 1920, 960, 10), (2160, 3840, 960, 10), (1080, 1920, 0, 0), (256, 256, 960, 10),
    (24, 40, 0, 0), (70, 75, 32, 10), (5, 3, 0, 0), (131, 61, 64, 10), (1, 1, 0, 0), (960, 960, 0, 0), (96, 128, 64, 10)])
@pytest.mark.parametrize("six", [0, 1, -1])
def test_walking_the_lists_like_the_kernel(uva, h, w, tile, border, six, monkeypatch):
    # 1: a workgroup's first segment starts with all six input rows; -1 (the default): only where that shortens the longest list
    if six >= 0:
        monkeypatch.setenv("UVA_TW_SIX", str(six))
    else:
        monkeypatch.delenv("UVA_TW_SIX", raising=False)
    steps, nsteps, planes, guard = schedule(uva, h, w, tile, border)
    if six >= 0:
        assert bool(((steps[:, 0, 1] >> 25) & 1).any()) == bool(six)
    grid, stride, _ = steps.shape
    cover = [np.zeros((int(p[0]), int(p[1])), np.int32) for p in planes]
    folded_planes = set()
    for b in range(grid):
        n = int(nsteps[b])
        assert 0 <= n <= stride - PAD
        dec = []
        for g in range(n + PAD):
            a, bb = steps[b, g, :4], steps[b, g, 4:]
            a_off = int(a[0]) | ((int(a[1]) & 0xff) << 32)
            a_act = (int(a[1]) >> 24) & 1
            if g >= n:      # look-ahead padding: a valid address, nothing active
                assert a_act == 0 and (int(bb[1]) >> 24) & 1 == 0 and (int(a[1]) >> 8) & 15 == 0
                if n:
                    assert a_off == (int(steps[b, n - 1, 0]) | ((int(steps[b, n - 1, 1]) & 0xff) << 32))
                continue
            assert a_act == 1
            fold = (int(a[1]) >> 27) & 1
            pi2 = -1
            if fold:
                # a FOLDED step (round 5): the narrow last strips of two planes of one size in one walk; .w = the byte distance of
                # the second plane's pixels - 2048, the first plane's index in the top byte of .z
                pi = int(a[2]) >> 24
                delta = int(np.uint32(a[3])) + 2048
                assert delta % 128 == 0
                match = [j for j in range(len(planes)) if int(planes[j, 3]) == int(planes[pi, 3]) + delta // 128]
                assert len(match) == 1 and match[0] > pi
                pi2 = match[0]
                assert tuple(planes[pi2, :3]) == tuple(planes[pi, :3]), "folded planes have one size"
                assert int(planes[pi, 1]) - (c_ := locate(a_off, planes, guard, pi)[1] + 1) <= 12      # (13, 14: producer pair 7 would read the other plane)
                folded_planes.add((pi, pi2))
            else:
                pi = int(a[3])
            row, col = locate(a_off, planes, guard, pi)
            # first new input row = pixel (yA + 1, x0 - 2) = array position (yA + 2, x0 - 1)
            yA, x0 = row - 2, col + 1
            ph, pw, pitch = int(planes[pi, 0]), int(planes[pi, 1]), int(planes[pi, 2])
            assert x0 % SW == 0 and 0 <= x0 < pw and (int(a[2]) & 0xffffff) == pitch * 128 and (fold or int(a[2]) >> 24 == 0)
            # the raw rows stay inside the 8-row guard around the plane arrays
            assert yA + 2 >= -8
            rmask, c_lo, c_hi = (int(a[1]) >> 8) & 15, (int(a[1]) >> 12) & 63, (int(a[1]) >> 18) & 63
            for r in range(4):
                assert ((rmask >> r) & 1) == (0 <= yA + r < ph)
            assert c_lo == (1 if x0 == 0 else 0) and c_hi == min(32, pw - x0 + 1)
            six = (int(a[1]) >> 25) & 1          # only a workgroup's first step may ask for all six input rows
            assert six == 0 or (g == 0 and not fold)
            dec.append((pi, yA, x0, bb, six, pi2))
        # ---- the kernel's walk: iteration `it`, phase X: A k-loop(it); phase Y: A epilogue(it) -> B-ring, raw rows of
        # step it + 1 -> A-ring, B k-loop(it - 1); B's stores of step it - 1 follow in iteration it + 1
        aring = [None] * AROWS          # (plane, x0, input row)
        bring = [None] * BROWS          # (plane, x0, intermediate row, valid, zero)
        a6 = b10 = 0

        def put_rows(g, pos0):
            pi, yA, x0, _, _, _ = dec[g]
            for wv in range(4):
                aring[(pos0 + wv) % AROWS] = (pi, x0, yA + 1 + wv)
        if n:
            put_rows(0, 2)              # prologue: step 0's new rows; positions 0, 1 hold nothing ...
            if dec[0][4]:               # ... unless the entry asks for the two rows above them as well
                aring[0], aring[1] = (dec[0][0], dec[0][2], dec[0][1] - 1), (dec[0][0], dec[0][2], dec[0][1])
        for it in range(n + 1):
            if it < n:
                pi, yA, x0, _, _, _ = dec[it]
                ph = int(planes[pi, 0])
                rows6 = [aring[(a6 + r) % AROWS] for r in range(6)]
                for nn in range(4):
                    want = [(pi, x0, yA + nn - 1 + d) for d in range(3)]
                    valid = rows6[nn:nn + 3] == want
                    inside = 0 <= yA + nn < ph
                    bring[(b10 + nn) % BROWS] = (pi, x0, yA + nn, valid or not inside, not inside)
                if it + 1 < n:
                    put_rows(it + 1, (a6 + 4 + 2) % AROWS)
            if 1 <= it <= n:
                pi, yA, x0, bb, _, pi2 = dec[it - 1]
                if (int(bb[1]) >> 24) & 1:
                    ph, pw = int(planes[pi, 0]), int(planes[pi, 1])
                    b_off = int(bb[0]) | ((int(bb[1]) & 0xff) << 32)
                    vy, vx, v0 = (int(bb[1]) >> 8) & 7, (int(bb[1]) >> 11) & 63, (int(bb[1]) >> 17) & 7
                    assert ((int(bb[1]) >> 25) & 1) == (1 if pi2 >= 0 else 0)
                    if pi2 >= 0:
                        assert int(np.uint32(bb[3])) + 2048 == (int(planes[pi2, 3]) - int(planes[pi, 3])) * 128 and vx <= 12
                    else:
                        assert int(bb[3]) == pi
                    assert (int(bb[2]) & 0xffffff) == int(planes[pi, 2]) * 128 and int(bb[2]) >> 24 == (pi if pi2 >= 0 else 0)
                    row, col = locate(b_off, planes, guard, pi)
                    yo = row - 1
                    assert col - 1 == x0 and yo == yA - 1 and 0 <= v0 < vy <= 4 and vx == min(SW, pw - x0)
                    assert 0 <= yo + v0 and yo + vy <= ph
                    bp = (b10 - 4 - 2) % BROWS          # the kernel's window: two rows above block it - 1
                    win = [bring[(bp + r) % BROWS] for r in range(6)]
                    for nn in range(v0, vy):
                        for d in range(3):
                            e = win[nn + d]
                            assert e is not None and e[:3] == (pi, x0, yo + nn - 1 + d) and e[3], (b, it, nn, d, e)
                    cover[pi][yo + v0:yo + vy, x0:x0 + vx] += 1
                    if pi2 >= 0:                         # pairs 8..15 of the same step: the same rows and columns of the second plane
                        cover[pi2][yo + v0:yo + vy, x0:x0 + vx] += 1
            a6 = (a6 + 4) % AROWS
            b10 = (b10 + 4) % BROWS
    for c in cover:
        assert c.min() == 1 and c.max() == 1
    if (h, w, tile, border) == (1080, 1920, 960, 10):
        # the reference tiling at 1080p: planes 970 x 970 (x 2) and 130 x 970 (x 2), last strips of 10 columns: both pairs fold
        assert folded_planes == {(0, 1), (2, 3)}
    # balance: the longest list is within a few steps of the mean over the workgroups that have work
    busy = int((nsteps > 0).sum())
    assert int(nsteps.max()) <= -(-int(nsteps.sum()) // busy) + 4


def test_folded_last_strips_save_their_steps(uva, monkeypatch):
    """970 columns are 32 strips and a third of one: with the two planes of a size sharing ONE walk of their last strips the
    reference tiling at 1080p needs 1.5 % fewer steps; UVA_TW_FOLD=0 is round 4's schedule"""
    monkeypatch.delenv("UVA_TW_SIX", raising=False)
    monkeypatch.setenv("UVA_TW_FOLD", "0")
    _, n0, _, _ = schedule(uva, 1080, 1920, 960, 10)
    monkeypatch.delenv("UVA_TW_FOLD")
    steps, n1, _, _ = schedule(uva, 1080, 1920, 960, 10)
    assert int(n0.sum()) == 18597 and int(n0.max()) == 73
    assert int(n0.sum()) - int(n1.sum()) >= 270 and int(n1.max()) <= 72
    assert ((steps[:, :, 1] >> 27) & 1).sum() >= 270


def test_six_row_starts_only_where_they_shorten_the_longest_list(uva, monkeypatch):
    monkeypatch.delenv("UVA_TW_SIX", raising=False)
    monkeypatch.setenv("UVA_TW_FOLD", "0")
    steps, nsteps, planes, guard = schedule(uva, 1080, 1920, 960, 10)        # reference tiling: 73 steps either way
    assert int(nsteps.max()) == 73 and not ((steps[:, 0, 1] >> 25) & 1).any()
    steps, nsteps, planes, guard = schedule(uva, 1080, 1920, 0, 0)           # whole frame: 69 -> 68
    assert int(nsteps.max()) == 68 and ((steps[:, 0, 1] >> 25) & 1).any()


def test_consecutive_ranges_share_an_xcd(uva, monkeypatch):
    """block b runs on XCD b % 8: the k-th contiguous range of the sequence goes to block (k % 32) * 8 + k // 32"""
    monkeypatch.setenv("UVA_TW_FOLD", "0")           # (folded entries carry no plane index: the placement rule is the same)
    steps, nsteps, planes, guard = schedule(uva, 1080, 1920, 960, 10)
    keys = []
    for b in range(0, 256, 8):
        if nsteps[b]:
            a = steps[b, 0, :4]
            row, col = locate(int(a[0]) | ((int(a[1]) & 0xff) << 32), planes, guard, int(a[3]))
            keys.append((int(a[3]), col + 1, row))
    assert keys == sorted(keys)
