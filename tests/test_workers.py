"""CPU tests of the worker-layer mirror (upscale_video_amd/upscale_processing.py): the pieces of
the reference's host logic that need no GPU."""
import logging
import math

import numpy as np
import pytest

from upscale_video_amd import upscale_processing as up
from upscale_video_amd import _imageio


def test_get_frames():
    assert up.get_frames("1,4-6,9") == [1, 4, 5, 6, 9]
    assert up.get_frames("7") == [7]


def test_logging_callback_exits_on_error(caplog):
    with caplog.at_level(logging.DEBUG):
        up.logging_callback([["info", "a"], ["debug", "b"]])
    assert "a" in caplog.text and "b" in caplog.text
    with pytest.raises(SystemExit):
        up.logging_callback([["info", "ok"], ["error", "boom"], ["info", "never logged"]])


def _reference_window(tile_size, y, x, height, width):
    """literal re-enactment of upscale_processing.py:398-427 (signed border offsets)"""
    ofs_y, ofs_x = y * tile_size, x * tile_size
    sy, ey = ofs_y, min(ofs_y + tile_size, height)
    sx, ex = ofs_x, min(ofs_x + tile_size, width)
    bsy = -10 if sy >= 10 else 0
    bey = 10 if ey <= height - 10 else 0
    bsx = -10 if sx >= 10 else 0
    bex = 10 if ex <= width - 10 else 0
    return (sy, ey, sx, ex), (-bsy, bey, -bsx, bex)


@pytest.mark.parametrize("h,w,ts", [(1080, 1920, 960), (2160, 3840, 960), (256, 256, 960), (70, 75, 32),
                                    (965, 1925, 960), (969, 970, 960), (20, 41, 32)])
def test_tile_window_matches_reference_logic(h, w, ts):
    for y in range(math.ceil(h / ts)):
        for x in range(math.ceil(w / ts)):
            assert up.tile_window(ts, y, x, h, w) == _reference_window(ts, y, x, h, w)


def test_init_worker_records_failures_instead_of_exiting(monkeypatch, tmp_path):
    """A Pool respawns a worker that dies in its initializer, forever: a worker that cannot
    be set up stays alive and every task it gets comes back as the reference's error items."""
    for gpus in ([], [-1]):
        up.init_worker(gpus, 0, "models", "x_Compact_Pretrain", 2, "input", "output")
        assert up.net is None and up.init_error
        items = up.upscale_image(str(tmp_path / "1.extract.png"), str(tmp_path / "1.png"), 2, 1, 1, 1)
        assert items[0] == ["error", "Upscale failed"] and items[1][0] == "error"
        items = up.apply_model(str(tmp_path / "1.extract.png"), str(tmp_path / "1.anime.png"), True)
        assert items[0] == ["error", "Model processing failed"]
        with pytest.raises(SystemExit):
            up.logging_callback(items)
    up.init_worker([0], 0, str(tmp_path), "x_missing_model", 2, "input", "output")
    assert up.net is None and "x_missing_model" in up.init_error


@pytest.mark.parametrize("persistent", [True, False])
def test_frame_queue_with_unusable_gpu_list_returns(monkeypatch, tmp_path, persistent):
    """process_model / upscale_frames with -g -1 end with the reference's exit, they do not hang."""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(up, "PERSISTENT_WORKERS", persistent)
    _imageio.imwrite("1.extract.png", np.zeros((8, 8, 3), np.uint8))
    with pytest.raises(SystemExit):
        up.process_model(1, "models", "x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g", 1, "input", "output",
                         "extract", "anime", [-1], 0)
    with pytest.raises(SystemExit):
        up.upscale_frames(1, 1, 1, "extract", 2, [], 0, "models", "x_Compact_Pretrain", "input", "output")
    assert __import__("os").path.exists("1.extract.png")


def test_reference_shaped_pool_reports_a_failed_worker_and_returns(monkeypatch, tmp_path):
    """PERSISTENT_WORKERS = False (fresh spawn Pool + init_worker, reference :565-577): a model that
    cannot be loaded makes every frame an error item; the run ends with SystemExit from the main
    thread after the pool has been joined, and the inputs are still there for the next run."""
    import os
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(up, "PERSISTENT_WORKERS", False)
    for n in (1, 2, 3):
        _imageio.imwrite("%d.extract.png" % n, np.zeros((8, 8, 3), np.uint8))
    with pytest.raises(SystemExit):
        up.upscale_frames(1, 1, 3, "extract", 2, [0, 0], 0, str(tmp_path), "x_no_such_model", "input", "output")
    assert all(os.path.exists("%d.extract.png" % n) for n in (1, 2, 3))


def test_imageio_roundtrip_is_bgr(tmp_path):
    img = np.zeros((5, 7, 3), np.uint8)
    img[..., 0] = 200   # blue in BGR
    img[2, 3] = (1, 2, 3)
    path = str(tmp_path / "x.png")
    assert _imageio.imwrite(path, img)
    back = _imageio.imread(path)
    assert np.array_equal(back, img)
    from PIL import Image
    rgb = np.asarray(Image.open(path).convert("RGB"))
    assert tuple(rgb[2, 3]) == (3, 2, 1) and rgb[0, 0, 2] == 200
    assert _imageio.imread(str(tmp_path / "missing.png")) is None
    # float canvases are converted like cv2.imwrite does: round half to even, saturate
    f = np.array([[[0.5, 1.5, 2.5], [-4.0, 255.5, 300.0]]])
    assert _imageio.to_u8(f).tolist() == [[[0, 2, 2], [0, 255, 255]]]
