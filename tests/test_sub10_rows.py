"""Row lists of sub10_kernel (the whole 24-feature 1x net in one launch, csrc/uva_kernels.hip.h) -- built on the
host by build_sub10_rows (csrc/uva_api.hip), fetched through the host-only hook uva_debug_sub10_rows and checked
for the properties the kernel relies on: every output pixel is written exactly once, every written row is
preceded by the 10 warm-up rows and followed by the 9 rows the layers in between still need, rows inside a
segment are consecutive, and the work is balanced."""
import ctypes

import numpy as np
import pytest

from upscale_video_amd import _lib

NL, VALID, WC = 10, 60, 80


def rows_for(h, w, grid=256):
    L = _lib.load()
    need, stride = ctypes.c_size_t(0), ctypes.c_int(0)
    L.uva_debug_sub10_rows(h, w, grid, None, 0, need, None, stride)
    words = np.zeros(need.value, np.uint32)
    nrows = np.zeros(grid, np.int32)
    rc = L.uva_debug_sub10_rows(h, w, grid, words.ctypes.data, words.size, need, nrows.ctypes.data, stride)
    assert rc == 0, L.uva_last_error()
    return words.view(np.int32).reshape(grid, stride.value, 4), nrows, stride.value


@pytest.mark.parametrize("h,w", [(1080, 1920), (480, 640), (7, 5), (1, 1), (61, 121), (2160, 3840), (33, 2000)])
def test_every_pixel_written_once_with_its_halo(h, w):
    rows, nrows, stride = rows_for(h, w)
    cover = np.zeros((h, w), np.int32)
    for b in range(rows.shape[0]):
        r = rows[b, :nrows[b]]
        assert (rows[b, nrows[b]:] == 0).all()
        i = 0
        while i < len(r):
            # a segment: rows y0-10 .. y1+9 of one strip, consecutive
            j = i
            while j + 1 < len(r) and r[j + 1, 0] == r[j, 0] + 1 and r[j + 1, 1] == r[i, 1]:
                j += 1
            seg = r[i:j + 1]
            emit = np.flatnonzero(seg[:, 2])
            assert len(emit) >= 1
            assert emit[0] == NL and len(seg) - 1 - emit[-1] == NL, "warm-up / tail rows"
            assert (np.diff(emit) == 1).all()
            # word 3: how far a row lies outside the rows that are written out (layer s of the net computes a row only where this
            # is <= 9 - s: the kernel's waves skip what nobody reads) -- 10, 9, .., 1 above, 0 inside, 1, .., 10 below
            k = np.arange(len(seg))
            want = np.where(k < emit[0], emit[0] - k, np.where(k > emit[-1], k - emit[-1], 0))
            assert (seg[:, 3] == want).all() and seg[0, 3] == NL and seg[-1, 3] == NL
            x0 = int(seg[0, 1])
            assert (x0 + NL) % VALID == 0
            y0, y1 = int(seg[emit[0], 0]), int(seg[emit[-1], 0]) + 1
            assert 0 <= y0 and y1 <= h
            xa, xb = max(x0 + NL, 0), min(x0 + NL + VALID, w)
            assert xa < xb
            cover[y0:y1, xa:xb] += 1
            i = j + 1
    assert (cover == 1).all()
    # balance: nobody carries more than the mean plus one segment's overhead (and a bit)
    total = nrows.sum()
    assert nrows.max() <= stride
    if h * ((w + VALID - 1) // VALID) >= 4 * rows.shape[0]:
        assert nrows.max() <= total / rows.shape[0] * 1.35 + 2 * NL


def test_xcd_placement_is_contiguous_per_xcd():
    # consecutive ranges of the (strip, row) sequence go to workgroups of the same XCD (block b -> XCD b % 8)
    rows, nrows, _ = rows_for(1080, 1920)
    first = {}
    for b in range(rows.shape[0]):
        if nrows[b]:
            first[b] = (int(rows[b, 0, 1]), int(rows[b, 0, 0]))
    order = sorted(first, key=lambda b: first[b])
    xcds = [b % 8 for b in order]
    changes = sum(1 for a, b in zip(xcds, xcds[1:]) if a != b)
    assert changes <= 7


def test_too_large_frame_is_refused():
    L = _lib.load()
    need, stride = ctypes.c_size_t(0), ctypes.c_int(0)
    assert L.uva_debug_sub10_rows(20000, 20000, 256, None, 0, need, None, stride) != 0
    assert b"too large" in L.uva_last_error()
