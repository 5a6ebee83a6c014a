"""CPU tests of the persistent frame workers (upscale_video_amd/frame_pool.py) behind process_model /
upscale_frames, with a stand-in net factory (tests/fake_net.py): the reference's queue semantics
(tasks only for inputs that exist, input deleted after the output exists, error item => exit) plus what
the reference does not have -- workers and nets that survive from batch to batch, and a rerun after a
crashed worker that redoes only the missing frames."""
import os

import numpy as np
import pytest

from upscale_video_amd import _imageio
from upscale_video_amd import frame_pool
from upscale_video_amd import upscale_processing as up


def _frame(n, h=6, w=10):
    img = np.random.default_rng(n).integers(0, 200, (h, w, 3), dtype=np.uint8)
    img[0, 0, 0] = n
    return img


@pytest.fixture
def fake(monkeypatch, tmp_path):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(up, "PERSISTENT_WORKERS", True)
    monkeypatch.setattr(up, "NET_FACTORY", "fake_net:make")
    yield str(tmp_path)
    up.shutdown_workers()


def _journal(path):
    p = os.path.join(path, "journal.txt")
    return [tuple(int(v) for v in line.split()) for line in open(p)] if os.path.exists(p) else []


def test_two_batches_one_set_of_workers(fake):
    frames = {n: _frame(n) for n in (1, 2, 3, 5, 6, 7, 8, 11)}
    for n, img in frames.items():
        _imageio.imwrite("%d.extract.png" % n, img)
    gpus = [0, 0, 1]
    up.upscale_frames(1, 1, 6, "extract", 2, gpus, 0, fake, "x_fake", "input", "output")
    for n in (1, 2, 3, 5, 6):
        assert not os.path.exists("%d.extract.png" % n)
        want = np.repeat(np.repeat(frames[n], 2, 0), 2, 1) + 1
        assert np.array_equal(_imageio.imread("%d.png" % n), want)
    assert not os.path.exists("4.png") and os.path.exists("7.extract.png")
    pids1 = {p.pid for p in up.get_frame_pool(gpus).procs}
    up.upscale_frames(2, 7, 12, "extract", 2, gpus, len(gpus), fake, "x_fake", "input", "output")   # the next batch
    assert {p.pid for p in up.get_frame_pool(gpus).procs} == pids1
    for n in (7, 8, 11):
        assert np.array_equal(_imageio.imread("%d.png" % n), np.repeat(np.repeat(frames[n], 2, 0), 2, 1) + 1)
    j = _journal(fake)
    assert sorted(f for _, f, _, _ in j) == sorted(frames)           # every frame exactly once
    assert {p for p, _, _, _ in j} <= pids1
    assert all(t == up.TILE_SIZE and b == up.TILE_BORDER for _, _, t, b in j)   # upscale_image's tiling
    loads = [line.split() for line in open(os.path.join(fake, "loads.txt"))]
    assert len(loads) <= len(gpus)                                    # one net per worker, not one per batch
    assert {int(g) for _, _, _, g in loads} <= {0, 1}


def test_two_batches_from_two_directories_on_one_pool(fake, tmp_path, monkeypatch):
    """the reference's orchestrator changes into every file's temp dir (upscale/upscale_processing.py:842, :971)
    and hands the workers names relative to it; persistent workers keep the directory they were spawned in.  The second
    file's frames -- same names -- must be read, written and deleted in ITS directory, with a relative model path too."""
    gpus = [0, 1]
    dirs = [tmp_path / "file_a", tmp_path / "file_b"]
    for k, d in enumerate(dirs):
        d.mkdir()
        monkeypatch.chdir(d)
        for n in (1, 2, 3):
            _imageio.imwrite("%d.extract.png" % n, _frame(10 * k + n))
        up.upscale_frames(1, 1, 4, "extract", 2, gpus, 0, os.path.relpath(fake), "x_fake", "input", "output")
    pids = {p.pid for p in up.get_frame_pool(gpus).procs}
    assert len(pids) == 2
    for k, d in enumerate(dirs):
        for n in (1, 2, 3):
            assert not (d / ("%d.extract.png" % n)).exists()
            want = np.repeat(np.repeat(_frame(10 * k + n), 2, 0), 2, 1) + 1
            assert np.array_equal(_imageio.imread(str(d / ("%d.png" % n))), want), (k, n)
    assert not [f for f in os.listdir(fake) if f.endswith(".png")]


def test_persistent_route_reports_the_tile_lines(fake, caplog):
    """upscale_image logs one "Processing Tile: i/n" debug item per reference tile before the frame's progress line
    (upscale/upscale_processing.py:507, :524-540); the persistent route builds them from the size the worker decoded."""
    import logging
    _imageio.imwrite("1.extract.png", _frame(1, h=7, w=23))
    up_tile = up.TILE_SIZE
    try:
        up.TILE_SIZE = 10                 # 3 x 1 tiles
        with caplog.at_level(logging.DEBUG):
            up.upscale_frames(None, 1, 1, "extract", 2, [0], 0, fake, "x_fake", "input", "output")
    finally:
        up.TILE_SIZE = up_tile
    msgs = [r.getMessage() for r in caplog.records]
    assert [m for m in msgs if m.startswith("Processing Tile")] == ["Processing Tile: %d/3" % i for i in (1, 2, 3)]
    assert msgs.index("Processing Tile: 3/3") < msgs.index("Upscaled 1/1")


def test_process_model_then_upscale_chain(fake):
    for n in (1, 2, 3):
        _imageio.imwrite("%d.extract.png" % n, _frame(n))
    up.process_model(3, fake, "x_fake", 1, "input", "output", "extract", "anime", [0, 0], 0, remove=True)
    assert all(os.path.exists("%d.anime.png" % n) and not os.path.exists("%d.extract.png" % n) for n in (1, 2, 3))
    assert all(t == 0 for _, _, t, _ in _journal(fake))              # apply_model: whole frame
    up.upscale_frames(1, 1, 3, "anime", 2, [0, 0], 2, fake, "x_fake", "input", "output")
    for n in (1, 2, 3):
        mid = _frame(n) + 1
        assert np.array_equal(_imageio.imread("%d.png" % n), np.repeat(np.repeat(mid, 2, 0), 2, 1) + 1)
        assert not os.path.exists("%d.anime.png" % n)


def test_crashed_worker_then_rerun_redoes_only_missing_frames(fake):
    n_frames = 12
    for n in range(1, n_frames + 1):
        _imageio.imwrite("%d.extract.png" % n, _frame(n))
    with pytest.raises(SystemExit):
        up.upscale_frames(1, 1, n_frames, "extract", 2, [0, 0], 0, fake, "x_fake_die7", "input", "output")
    done_first = {n for n in range(1, n_frames + 1) if not os.path.exists("%d.extract.png" % n)}
    assert 7 not in done_first and os.path.exists("7.extract.png")
    for n in done_first:                                             # delete-after-write: removed inputs have outputs
        assert _imageio.imread("%d.png" % n) is not None
    first = {f for _, f, _, _ in _journal(fake)}
    os.remove(os.path.join(fake, "journal.txt"))
    up.upscale_frames(1, 1, n_frames, "extract", 2, [0, 0], 0, fake, "x_fake", "input", "output")
    second = [f for _, f, _, _ in _journal(fake)]
    assert sorted(second) == sorted(set(range(1, n_frames + 1)) - done_first)   # exactly the missing ones
    assert done_first <= first
    for n in range(1, n_frames + 1):
        assert np.array_equal(_imageio.imread("%d.png" % n), np.repeat(np.repeat(_frame(n), 2, 0), 2, 1) + 1)
        assert not os.path.exists("%d.extract.png" % n)


def test_error_item_ends_the_run_and_keeps_inputs(fake):
    for n in range(1, 7):
        _imageio.imwrite("%d.extract.png" % n, _frame(n))
    with pytest.raises(SystemExit):
        up.upscale_frames(1, 1, 6, "extract", 2, [0], 0, fake, "x_fake_fail2", "input", "output")
    assert os.path.exists("2.extract.png") and not os.path.exists("2.png")
    with pytest.raises(SystemExit):                                  # a model that cannot be loaded: every frame fails
        up.upscale_frames(1, 1, 6, "extract", 2, [0], 0, fake, "x_bad", "input", "output")
    assert os.path.exists("2.extract.png")


def test_unreadable_input_is_an_error_item(fake):
    open("1.extract.png", "wb").write(b"not a png")
    with pytest.raises(SystemExit):
        up.upscale_frames(1, 1, 1, "extract", 2, [0], 0, fake, "x_fake", "input", "output")
    assert os.path.exists("1.extract.png")


def test_parse_cpulist_and_affinity_lookup_never_raises():
    assert frame_pool.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert frame_pool.parse_cpulist("") == []
    aff = frame_pool.gpu_cpu_affinity(0)           # no GPU here: None; on the box: the GPU's NUMA node CPUs
    assert aff is None or (len(aff) > 0 and set(aff) <= os.sched_getaffinity(0))
