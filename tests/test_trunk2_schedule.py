"""CPU test of the step lists trunk2_kernel (two fused trunk layers per launch) walks: decoded through
the host-only hook uva_debug_trunk2_schedule and checked for the properties the kernel relies on --
every plane pixel is produced by exactly one consumer step, the two producer blocks a consumer step
reads are the right rows of the same strip, the producer's zero masks are exactly "outside the plane",
the dummy look-ahead entries are inert, and the work is spread evenly."""
import ctypes

import numpy as np
import pytest

SW = 30   # csrc/uva_kernels.hip.h T2_SW


def schedule(uva, h, w, tile, border, grid=256):
    from upscale_video_amd import _lib
    L = _lib.load()
    need = ctypes.c_size_t()
    stride, nplanes, guard = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
    pinfo = np.zeros(64 * 4, np.int64)
    nsteps = np.zeros(grid, np.int32)
    L.uva_debug_trunk2_schedule(h, w, tile, border, grid, None, 0, need, None, stride, pinfo.ctypes.data, 64, nplanes, guard)
    words = np.zeros(need.value, np.uint32)
    rc = L.uva_debug_trunk2_schedule(h, w, tile, border, grid, words.ctypes.data, words.size, need, nsteps.ctypes.data,
                                     stride, pinfo.ctypes.data, 64, nplanes, guard)
    assert rc == 0, L.uva_last_error()
    planes = pinfo[:4 * nplanes.value].reshape(-1, 4)   # h, w, pitch, act_off
    return words.reshape(grid, stride.value, 8), nsteps, planes, guard.value


def locate(off, planes, guard, pi):
    """byte offset -> (plane, row, column) in plane pi's array; rows / columns may lie above or left of it"""
    pix = (off - guard) // 128
    assert (off - guard) % 128 == 0
    rel = pix - planes[pi, 3]
    pitch = planes[pi, 2]
    row = rel // pitch if rel >= 0 else -((-rel + pitch - 1) // pitch)
    return pi, int(row), int(rel - row * pitch)


@pytest.mark.parametrize("h,w,tile,border", [
    (1080, 1920, 960, 10), (2160, 3840, 960, 10), (1080, 1920, 0, 0), (256, 256, 960, 10),
    (24, 40, 0, 0), (70, 75, 32, 10), (5, 3, 0, 0), (131, 61, 64, 10), (1, 1, 0, 0), (960, 960, 0, 0)])
def test_step_lists_cover_every_pixel_once(uva, h, w, tile, border):
    steps, nsteps, planes, guard = schedule(uva, h, w, tile, border)
    grid, stride, _ = steps.shape
    cover = [np.zeros((int(p[0]), int(p[1])), np.int32) for p in planes]
    total = 0
    costs = []
    for b in range(grid):
        n = int(nsteps[b])
        cost = 0
        assert 0 <= n <= stride - 3
        total += n
        dec = []
        for g in range(n + 3):
            a, bb = steps[b, g, :4], steps[b, g, 4:]
            a_off = int(a[0]) | ((int(a[1]) & 0xff) << 32)
            a_act = (int(a[1]) >> 24) & 1
            if g >= n:      # look-ahead padding: a valid address, nothing active
                assert a_act == 0 and (int(bb[1]) >> 24) & 1 == 0 and (int(a[1]) >> 8) & 15 == 0
                assert a_off == (int(steps[b, n - 1, 0]) | ((int(steps[b, n - 1, 1]) & 0xff) << 32)) if n else True
                continue
            assert a_act == 1
            pi, row, col = locate(a_off, planes, guard, int(a[3]))
            # at x0 = 0 the halo origin is one column left of the array row: array col -1 = previous row, col pitch-1
            pitch = int(planes[pi, 2])
            if col == pitch - 1:
                row, col = row + 1, -1
            yA, x0 = row, col + 1          # halo origin = array (yA, x0 - 1)
            ph, pw = int(planes[pi, 0]), int(planes[pi, 1])
            assert x0 % SW == 0 and 0 <= x0 < pw and int(a[2]) == pitch * 128
            rmask, c_lo, c_hi = (int(a[1]) >> 8) & 15, (int(a[1]) >> 12) & 63, (int(a[1]) >> 18) & 63
            for r in range(4):
                assert ((rmask >> r) & 1) == (0 <= yA + r < ph)
            assert c_lo == (1 if x0 == 0 else 0) and c_hi == min(32, pw - x0 + 1)
            narrow = (int(a[1]) >> 25) & 1
            assert narrow == (1 if pw - x0 <= 14 else 0)
            cost += 8 if narrow else 10
            b_act = (int(bb[1]) >> 24) & 1
            dec.append((pi, yA, x0, b_act, bb))
        for g, (pi, yA, x0, b_act, bb) in enumerate(dec):
            if not b_act:
                continue
            ph, pw = int(planes[pi, 0]), int(planes[pi, 1])
            b_off = int(bb[0]) | ((int(bb[1]) & 0xff) << 32)
            vy, vx = (int(bb[1]) >> 8) & 7, (int(bb[1]) >> 11) & 31
            qi, row, col = locate(b_off, planes, guard, int(bb[3]))
            assert qi == pi and col - 1 == x0 and int(bb[2]) == int(planes[pi, 2]) * 128
            assert (int(bb[1]) >> 25) & 1 == (1 if pw - x0 <= 14 else 0)
            yo = row - 1
            assert yo == yA + 1 and 1 <= vy <= 4 and vx == min(SW, pw - x0) and yo + vy <= ph
            if vy >= 3:     # rows yo+2 .. need intermediate rows of the NEXT block: same strip, 4 rows further down
                assert g + 1 < len(dec) and dec[g + 1][:3] == (pi, yA + 4, x0)
            cover[pi][yo:yo + vy, x0:x0 + vx] += 1
        costs.append(cost)
    for c in cover:
        assert c.min() == 1 and c.max() == 1
    # balance: by COST -- a step of a narrow strip (<= 14 columns, flagged in both halves: one fragment column instead of
    # two) counts 8 tenths -- the heaviest list is within a few steps of the mean over the workgroups that have work
    busy = int((nsteps > 0).sum())
    assert max(costs) <= -(-sum(costs) // busy) + 4 * 10


def test_consecutive_ranges_share_an_xcd(uva):
    """block b runs on XCD b % 8: the k-th contiguous range of the sequence goes to block (k % 32) * 8 + k // 32"""
    steps, nsteps, planes, guard = schedule(uva, 1080, 1920, 960, 10)
    first = {}
    for b in range(256):
        if nsteps[b]:
            a = steps[b, 0, :4]
            first[b] = int(a[0]) | ((int(a[1]) & 0xff) << 32)
    # within XCD 0 (blocks 0, 8, 16, ...) the ranges advance monotonically through plane / strip order
    seq = [locate(first[b], planes, guard, int(steps[b, 0, 3])) for b in range(0, 256, 8) if b in first]
    keys = [(pi, col + 1 if col < 900 else 0, row) for pi, row, col in seq]   # x0 = 0: origin one column left of the row
    assert keys == sorted(keys)
