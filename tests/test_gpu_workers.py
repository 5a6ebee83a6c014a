"""GPU tests of the worker-layer mirror: the reference's file-to-file functions driven the way
process_file drives them (spawn pool, one worker per -g entry, filename tasks)."""
import os

import numpy as np
import pytest

from conftest import ROOT, psnr_u8

pytestmark = pytest.mark.gpu
MODELS = os.path.join(ROOT, "models")


def test_config1_png_to_png_256(tmp_path, oracle, oracle_models, uva, monkeypatch):
    """BASELINE config 1: single 256x256 PNG, 2x Compact, through the PNG -> PNG plumbing."""
    from upscale_video_amd import upscale_processing as up
    from upscale_video_amd import _imageio
    assert uva.get_gpu_count() > 0
    monkeypatch.chdir(tmp_path)
    img = oracle.synthetic_frame(256, 256)
    _imageio.imwrite("1.extract.png", img)
    up.init_worker([0], 0, MODELS, "x_Compact_Pretrain", 2, "input", "output")
    items = up.upscale_image("1.extract.png", "1.png", 2, 1, 1, 1, remove=True)
    assert not any(level == "error" for level, _ in items), items
    assert items[-1] == ["info", "Upscaling Batch: 1 : Upscaled 1/1"]
    assert not os.path.exists("1.extract.png")
    out = _imageio.imread("1.png")
    want = oracle_models["2x"].upscale_image(img)      # 256 < 960: one tile, no border
    assert out.shape == (512, 512, 3)
    assert np.abs(out.astype(int) - want.astype(int)).max() <= 2 and psnr_u8(out, want) >= 50
    # float route (reference-shaped) agrees with the fused one
    _imageio.imwrite("2.extract.png", img)
    monkeypatch.setattr(up, "FUSED_DEVICE_PATH", False)
    up.upscale_image("2.extract.png", "2.png", 2, None, 2, 2, remove=False)
    out2 = _imageio.imread("2.png")
    assert np.abs(out2.astype(int) - out.astype(int)).max() <= 1
    # missing input -> error items, like a failed imread in the reference's try block
    items = up.apply_model("nope.png", "x.png", False)
    assert items[0][0] == "error"


@pytest.mark.parametrize("persistent", [True, False])
def test_frame_queue_two_workers_one_gpu(tmp_path, oracle, oracle_models, monkeypatch, persistent):
    """process_model + upscale_frames: '-g 0,0' = two spawned workers sharing GPU 0
    (README.md:45-61), tasks only for existing inputs, inputs deleted after outputs exist -- on the
    persistent FramePool workers and on the reference-shaped route (a fresh spawn Pool per call, pool
    identities offset by `workers_used`)."""
    from upscale_video_amd import upscale_processing as up
    from upscale_video_amd import _imageio
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(up, "PERSISTENT_WORKERS", persistent)
    frames = {}
    for n in (1, 2, 3, 5):
        frames[n] = oracle.synthetic_frame(24, 40, seed=n)
        _imageio.imwrite(f"{n}.extract.png", frames[n])
    gpus = [0, 0]
    used = 0 if persistent else up.workers_spawned_so_far()     # a fresh process would pass 0 (reference :880)
    up.process_model(5, MODELS, "x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g", 1, "input", "output",
                     "extract", "anime", gpus, used, remove=True)
    workers_used = used + len(gpus) if persistent else up.workers_spawned_so_far()
    for n in frames:
        assert os.path.exists(f"{n}.anime.png") and not os.path.exists(f"{n}.extract.png")
    assert not os.path.exists("4.anime.png")
    up.upscale_frames(1, 1, 5, "anime", 2, gpus, workers_used, MODELS, "x_Compact_Pretrain", "input", "output")
    for n, img in frames.items():
        mid = oracle_models["1x"].apply_model(img)
        want = oracle_models["2x"].upscale_image(mid)
        out = _imageio.imread(f"{n}.png")
        assert out is not None and out.shape == (48, 80, 3)
        assert np.abs(out.astype(int) - want.astype(int)).max() <= 3 and psnr_u8(out, want) >= 48
        assert not os.path.exists(f"{n}.anime.png")
    up.shutdown_workers()
