"""Row lists of sub5_kernel (the 1x net as two launches of five layers, two pipelines = a PAIR of 54-column strips per workgroup;
csrc/uva_sub5.hip.h) -- built on the host by build_sub5_rows (csrc/uva_api.hip), fetched through the host-only hook
uva_debug_sub5_rows and checked for what both launches rely on: every pixel of the plane lies in the written-out rows and the
valid columns of exactly one (segment, strip); every segment has its 5 warm-up rows above and 5 below; rows inside a segment are
consecutive; the work is balanced; consecutive ranges go to one XCD."""
import ctypes

import numpy as np
import pytest

from upscale_video_amd import _lib

NL, VALID, WC, PAIRW = 5, 54, 64, 108


def rows_for(h, w, grid=256):
    L = _lib.load()
    need, stride = ctypes.c_size_t(0), ctypes.c_int(0)
    L.uva_debug_sub5_rows(h, w, grid, None, 0, need, None, stride)
    words = np.zeros(need.value, np.uint32)
    nrows = np.zeros(grid, np.int32)
    rc = L.uva_debug_sub5_rows(h, w, grid, words.ctypes.data, words.size, need, nrows.ctypes.data, stride)
    assert rc == 0, L.uva_last_error()
    return words.view(np.int32).reshape(grid, stride.value, 4), nrows, stride.value


@pytest.mark.parametrize("h,w", [(1080, 1920), (480, 640), (7, 5), (1, 1), (61, 121), (2160, 3840), (33, 2000), (50, 54), (50, 55), (9, 109)])
def test_every_pixel_in_exactly_one_segment_and_strip(h, w):
    rows, nrows, stride = rows_for(h, w)
    cover = np.zeros((h, w), np.int32)
    for b in range(rows.shape[0]):
        r = rows[b, :nrows[b]]
        assert (rows[b, nrows[b]:] == 0).all()
        i = 0
        while i < len(r):
            j = i
            while j + 1 < len(r) and r[j + 1, 0] == r[j, 0] + 1 and r[j + 1, 1] == r[i, 1]:
                j += 1
            seg = r[i:j + 1]
            emit = np.flatnonzero(seg[:, 2])
            assert len(emit) >= 1
            assert emit[0] == NL and len(seg) - 1 - emit[-1] == NL, "five warm-up rows above, five below"
            assert (np.diff(emit) == 1).all()
            x0 = int(seg[0, 1])
            assert (x0 + NL) % PAIRW == 0
            y0, y1 = int(seg[emit[0], 0]), int(seg[emit[-1], 0]) + 1
            assert 0 <= y0 and y1 <= h
            wrote = False
            for pipe in (0, 1):                      # the workgroup's two pipelines: computed column 0 at x0 and x0 + 54
                xa, xb = max(x0 + pipe * VALID + NL, 0), min(x0 + pipe * VALID + NL + VALID, w)
                if xa < xb:
                    cover[y0:y1, xa:xb] += 1
                    wrote = True
            assert wrote
            i = j + 1
    assert (cover == 1).all()
    assert nrows.max() <= stride <= 1024
    if h * ((w + PAIRW - 1) // PAIRW) >= 4 * rows.shape[0]:
        assert nrows.max() <= nrows.sum() / rows.shape[0] * 1.35 + 2 * NL


def test_1080p_has_fewer_and_shorter_warm_ups_than_the_ten_layer_kernel():
    """what the split is for: per launch 54 of 64 columns and n of n + 10 rows are kept (sub10_kernel: 60 of 80, n of n + 20)"""
    rows, nrows, _ = rows_for(1080, 1920)
    emitted = int(rows[..., 2].sum())
    assert emitted == 18 * 1080                                   # 18 strip pairs cover 1920 columns (36 x 54 = 1944)
    kept_rows = emitted / nrows.sum()
    assert kept_rows > 0.87
    useful = kept_rows * (1920 / (36 * WC))                       # kept rows x kept columns, per launch
    assert useful > 0.72                                          # sub10_kernel: 0.75 x 0.87 = 0.65


def test_xcd_placement_is_contiguous_per_xcd():
    rows, nrows, _ = rows_for(1080, 1920)
    first = {b: (int(rows[b, 0, 1]), int(rows[b, 0, 0])) for b in range(rows.shape[0]) if nrows[b]}
    order = sorted(first, key=lambda b: first[b])
    xcds = [b % 8 for b in order]
    assert sum(1 for a, b in zip(xcds, xcds[1:]) if a != b) <= 7


def test_too_large_frame_is_refused():
    L = _lib.load()
    need, stride = ctypes.c_size_t(0), ctypes.c_int(0)
    assert L.uva_debug_sub5_rows(20000, 20000, 256, None, 0, need, None, stride) != 0
    assert b"too large" in L.uva_last_error()
