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


# ---- several frames of one geometry in one launch (uva_net_process_u8_device_batch, round 6) -------------------------------
def rows_for_batch(h, w, frames, grid=256):
    L = _lib.load()
    need, stride = ctypes.c_size_t(0), ctypes.c_int(0)
    L.uva_debug_sub10_rows_batch(h, w, frames, grid, None, 0, need, None, stride)
    words = np.zeros(need.value, np.uint32)
    nrows = np.zeros(grid, np.int32)
    rc = L.uva_debug_sub10_rows_batch(h, w, frames, grid, words.ctypes.data, words.size, need, nrows.ctypes.data, stride)
    assert rc == 0, L.uva_last_error()
    return words.view(np.int32).reshape(grid, stride.value, 4), nrows, stride.value


@pytest.mark.parametrize("h,w,frames", [(1080, 1920, 2), (1080, 1920, 3), (1080, 1920, 4), (480, 640, 8), (7, 5, 5), (1, 1, 8), (720, 1280, 8)])
def test_batch_rows_cover_every_frame_once(h, w, frames):
    """the sequence runs over (frame, strip, row): every pixel of every frame written exactly once, a segment never
    straddles two frames, warm-up / tail rows and the distance word as for one frame"""
    rows, nrows, stride = rows_for_batch(h, w, frames)
    assert stride <= 640                     # S10_MAX_ROWS: the kernel's LDS copy
    cover = np.zeros((frames, h, w), np.int32)
    for b in range(rows.shape[0]):
        r = rows[b, :nrows[b]]
        i = 0
        while i < len(r):
            j = i
            while j + 1 < len(r) and r[j + 1, 0] == r[j, 0] + 1 and r[j + 1, 1] == r[i, 1] and (r[j + 1, 2] >> 8) == (r[i, 2] >> 8):
                j += 1
            seg = r[i:j + 1]
            f = int(seg[0, 2]) >> 8
            assert 0 <= f < frames and ((seg[:, 2] >> 8) == f).all()
            emit = np.flatnonzero(seg[:, 2] & 1)
            assert len(emit) >= 1 and emit[0] == NL and len(seg) - 1 - emit[-1] == NL and (np.diff(emit) == 1).all()
            k = np.arange(len(seg))
            assert (seg[:, 3] == np.where(k < emit[0], emit[0] - k, np.where(k > emit[-1], k - emit[-1], 0))).all()
            x0 = int(seg[0, 1])
            y0, y1 = int(seg[emit[0], 0]), int(seg[emit[-1], 0]) + 1
            assert (x0 + NL) % VALID == 0 and 0 <= y0 and y1 <= h
            cover[f, y0:y1, max(x0 + NL, 0):min(x0 + NL + VALID, w)] += 1
            i = j + 1
    assert (cover == 1).all()


def test_batch_of_one_is_the_single_frame_list():
    a, na, sa = rows_for(1080, 1920)
    b, nb, sb = rows_for_batch(1080, 1920, 1)
    assert sa == sb and (na == nb).all() and (a == b).all()


def test_batches_pay_the_warm_up_rows_once_per_launch():
    """what the batch is for: a workgroup's steps per frame (its rows + the pipeline's 20 steps of fill and drain) fall with
    the number of frames in the launch -- 175 for one 1080p frame, ~155 at two, ~145 at four"""
    per_frame = {}
    for k in (1, 2, 4):
        _, nrows, _ = rows_for_batch(1080, 1920, k)
        per_frame[k] = (nrows.max() + 20) / k
    assert per_frame[1] > 170 and per_frame[2] < 160 and per_frame[4] < 150, per_frame


def test_a_batch_that_does_not_fit_is_refused():
    L = _lib.load()
    need, stride = ctypes.c_size_t(0), ctypes.c_int(0)
    assert L.uva_debug_sub10_rows_batch(1080, 1920, 8, 256, None, 0, need, None, stride) != 0      # 8 x 135 rows > 640
    assert L.uva_debug_sub10_rows_batch(1080, 1920, 9, 256, None, 0, need, None, stride) != 0      # > S10_MAXB
    assert L.uva_debug_sub10_rows_batch(1080, 1920, 0, 256, None, 0, need, None, stride) != 0
