"""-m gpu: sub5_kernel (UVA_SUB5=1: the 1x HurrDeblur net as two launches of five layers, csrc/uva_sub5.hip.h) against
sub10_kernel (the whole net in one launch: the same arithmetic in the same order, so the SAME BYTES are asked for), against the
fp32 oracle, and on the geometries that exercise its row lists: planes narrower than a strip, an odd number of strips (the second
pipeline of the last pair idle), one pixel, frames whose segments are cut in the middle of a strip pair."""
import numpy as np
import pytest

try:
    import torch  # noqa: F401  (its HIP runtime first, see test_gpu_parity.py)
except Exception:  # noqa: BLE001
    torch = None

from conftest import load_net
from parity_report import check_u8, fp32_bar

pytestmark = pytest.mark.gpu
FP32 = "fp32 oracle"


@pytest.fixture(scope="module")
def both(uva):
    import os
    assert uva.get_gpu_count() > 0
    os.environ["UVA_SUB5"] = "1"          # (the switches are read when a net's device side is built)
    try:
        split = load_net(uva, "1x")
        split.process_u8(np.zeros((8, 8, 3), np.uint8), tile_size=0)
    finally:
        del os.environ["UVA_SUB5"]
    whole = load_net(uva, "1x")
    return split, whole


@pytest.mark.parametrize("h,w,kind", [(50, 33, "random"), (40, 72, "smooth"), (1, 1, "random"), (3, 2, "random"), (37, 54, "random"),
                                      (21, 55, "random"), (64, 109, "random"), (130, 216, "smooth"), (300, 700, "random"), (9, 1000, "random")])
def test_two_launches_of_five_layers_give_the_bytes_of_the_one_launch_kernel(both, oracle_models, oracle, h, w, kind):
    split, whole = both
    img = oracle.synthetic_frame(h, w, kind=kind, seed=31 * h + w)
    a = split.process_u8(img, tile_size=0)
    b = whole.process_u8(img, tile_size=0)
    assert np.array_equal(a, b), (h, w, int(np.abs(a.astype(int) - b.astype(int)).max()), float((a != b).mean()))
    if h * w <= 130 * 216:
        check_u8(f"1x sub5_kernel {w}x{h} {kind}", a, oracle_models["1x"].apply_model(img), vs=FP32, model="1x", route="whole", **fp32_bar("1x", "whole"))


def test_full_size_frame_equals_the_one_launch_kernel_and_the_oracle_in_windows(both, oracle_models, oracle):
    split, whole = both
    h, w = 1080, 1920
    rad = 10
    img = oracle.synthetic_frame(h, w, seed=20260929)
    a = split.process_u8(img, tile_size=0)
    assert np.array_equal(a, split.process_u8(img, tile_size=0))            # determinism (the image between the launches is reused)
    assert np.array_equal(a, whole.process_u8(img, tile_size=0))
    win = 24
    for (y0, x0) in [(0, 0), (0, w - win), (h - win, 0), (h - win, w - win), (h // 2, w // 2), (500, 96), (75, 1000)]:
        cy0, cx0, cy1, cx1 = max(0, y0 - rad), max(0, x0 - rad), min(h, y0 + win + rad), min(w, x0 + win + rad)
        want = oracle_models["1x"].apply_model(np.ascontiguousarray(img[cy0:cy1, cx0:cx1]))[y0 - cy0:y0 - cy0 + win, x0 - cx0:x0 - cx0 + win]
        check_u8(f"1x sub5_kernel 1080p window ({y0},{x0})", np.ascontiguousarray(a[y0:y0 + win, x0:x0 + win]), np.ascontiguousarray(want),
                 vs=FP32, max_lsb=1, min_psnr=55, model="1x", route="whole")


def test_other_frame_sizes_in_turn_reuse_nothing_stale(both, oracle):
    """a net's workspaces are cached per geometry, the image between the launches with them: sizes in turn, twice"""
    split, whole = both
    frames = [oracle.synthetic_frame(h, w, seed=h) for h, w in ((60, 200), (61, 120), (60, 200), (200, 60))]
    for _ in range(2):
        for f in frames:
            assert np.array_equal(split.process_u8(f, tile_size=0), whole.process_u8(f, tile_size=0))


def test_seeded_random_geometries_equal_the_one_launch_kernel(both, oracle):
    """sub10_kernel computes a layer only where a stored pixel needs it (round 5: four-fragment late layers, rows out of every
    layer's reach skipped); sub5_kernel computes every column and row of its strips.  Forty seeded frame sizes -- workgroups with
    one short segment, with several, with segments cut at a strip's end -- must give the same bytes from both."""
    split, whole = both
    rng = np.random.default_rng(20260929)
    sizes = [(int(rng.integers(1, 1300)), int(rng.integers(1, 2000))) for _ in range(36)] + [(1300, 61), (2, 1999), (1199, 60), (257, 121)]
    for k, (h, w) in enumerate(sizes):
        img = oracle.synthetic_frame(h, w, kind="random" if k & 1 else "smooth", seed=1000 + k)
        a = split.process_u8(img, tile_size=0)
        b = whole.process_u8(img, tile_size=0)
        assert np.array_equal(a, b), (h, w, int(np.abs(a.astype(int) - b.astype(int)).max()), float((a != b).mean()))
