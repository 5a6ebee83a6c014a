"""GPU tests (-m gpu) of the BASELINE.json configurations that the small-frame parity tests do not reach:
config 5's frame (3840x2160, 2x Compact, 12 reference tiles), the device limits of the tile schedule, the
test_gpus.py harness, and an opportunistic comparison with the real ncnn / cv2 when
the box happens to have them (it normally does not: the reference pins neither, README.md:29)."""
import logging
import os

import numpy as np
import pytest

try:
    import torch  # noqa: F401  (its HIP runtime first, see test_gpu_parity.py)
except Exception:  # noqa: BLE001
    torch = None

from conftest import ROOT, load_net, psnr_u8

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net2x(uva):
    assert uva.get_gpu_count() > 0
    return load_net(uva, "2x")


def test_config5_frame_2160p_reference_tiling(net2x, oracle_models, oracle):
    """3840x2160 in, 2x Compact, 960/10 tiling = 12 planes of up to 980x980 (tile grid
    upscale_processing.py:499-516):
      (a) away from the seams the tiled result equals the whole-frame result, which is checked against the
          oracle by locality windows (18-px receptive radius);
      (b) ON the seams the fused route equals the reference-shaped float route (process_tile: cut core +
          border, extract, *255, paste core) run on the two tiles either side of a seam;
      (c) the tiled and the whole-frame results differ only within the receptive radius of a seam."""
    from upscale_video_amd import upscale_processing as up
    net, om = net2x, oracle_models["2x"]
    h, w, s, rad = 2160, 3840, 2, net2x.num_convs
    img = oracle.synthetic_frame(h, w, seed=5)
    tiled = net.process_u8(img, tile_size=960, border=10)
    assert tiled.shape == (h * s, w * s, 3)
    assert np.array_equal(tiled, net.process_u8(img, tile_size=960, border=10))
    whole = net.process_u8(img, tile_size=0)
    win = 24
    for (y0, x0) in [(0, 0), (h - win, w - win), (400, 1400), (1500, 2500), (2000, 3700), (1000, 100)]:
        cy0, cx0 = max(0, y0 - rad), max(0, x0 - rad)
        cy1, cx1 = min(h, y0 + win + rad), min(w, x0 + win + rad)
        want = om.apply_model(np.ascontiguousarray(img[cy0:cy1, cx0:cx1]))[
            (y0 - cy0) * s:(y0 - cy0 + win) * s, (x0 - cx0) * s:(x0 - cx0 + win) * s]
        for name, res in (("whole", whole), ("tiled", tiled)):
            got = res[y0 * s:(y0 + win) * s, x0 * s:(x0 + win) * s]
            d = np.abs(got.astype(int) - want.astype(int))
            assert d.max() <= 2 and psnr_u8(got, want) >= 50, (name, (y0, x0), d.max(), psnr_u8(got, want))
    # (c)
    d = np.abs(tiled.astype(np.int16) - whole.astype(np.int16))
    assert d.max() <= 2
    ys, xs = np.nonzero(d.max(axis=2))
    near = np.zeros(len(ys), bool)
    for k in (1, 2, 3):
        near |= np.abs(xs - 960 * k * s) <= (rad + 2) * s
    for k in (1, 2):
        near |= np.abs(ys - 960 * k * s) <= (rad + 2) * s
    assert near.all()
    # (b) tiles (1, 1) and (1, 2): interior tiles with a border on every side, sharing the x = 1920 seam
    up.net = net
    for (ty, tx) in [(1, 1), (1, 2)]:
        canvas = np.zeros((h * s, w * s, 3))
        items = []
        assert up.process_tile(img, 960, s, ty, tx, h, w, canvas, items) == 0, items
        (y0, y1, x0, x1), _ = up.tile_window(960, ty, tx, h, w)
        floaty = np.clip(np.rint(canvas[y0 * s:y1 * s, x0 * s:x1 * s]), 0, 255).astype(np.uint8)
        got = tiled[y0 * s:y1 * s, x0 * s:x1 * s]
        dd = np.abs(got.astype(np.int16) - floaty.astype(np.int16))
        # the float route rounds x/255 to fp16 at the head, the u8 route feeds exact integers
        assert dd.max() <= 1 and (dd > 0).mean() <= 5e-2, ((ty, tx), int(dd.max()), float((dd > 0).mean()))


def test_schedule_limits_fail_with_messages(net2x, uva, oracle):
    """The encodings of the tile schedules have limits (uva_api.hip: 64 planes per frame, 255 tile columns
    and 4095 4-row tile rows per plane): beyond them the call fails with a message, it does not compute."""
    from upscale_video_amd import _lib
    img = oracle.synthetic_frame(90, 90)
    with pytest.raises(_lib.UvaError, match="more than 64 tiles"):
        net2x.process_u8(img, tile_size=10, border=2)                 # 9 x 9 = 81 planes
    ok = net2x.process_u8(img, tile_size=12, border=2)                # 8 x 8 = 64 planes: the most there can be
    assert ok.shape == (180, 180, 3)
    wide = np.zeros((2, 8200, 3), np.uint8)                           # 257 tile columns in one plane
    with pytest.raises(_lib.UvaError, match="too large for the tile schedule"):
        net2x.process_u8(wide, tile_size=0)
    assert net2x.process_u8(np.zeros((2, 8100, 3), np.uint8), tile_size=0).shape == (4, 16200, 3)
    tall = np.zeros((16400, 2, 3), np.uint8)                          # 4100 4-row tile rows in one plane
    with pytest.raises(_lib.UvaError, match="too large for the tile schedule"):
        net2x.process_u8(tall, tile_size=0)
    # the net is still usable afterwards
    assert np.array_equal(net2x.process_u8(img, tile_size=12, border=2), ok)


def test_harness_test_gpus_two_workers(caplog, tmp_path, monkeypatch):
    """test_gpus.py (counterpart of the reference's only benchmark harness, test_gpus.py:15-127) run the
    way BASELINE config 5 names it, here with `-g 0,0 -s 2 -r 4` on one GPU: the device listing, one
    "Testing GPU" and one "seconds to upscale" line per run, the total, and frames/s.  The transcript is
    kept under gpurun_out/ when that directory exists (copied to profiles/ by hand)."""
    import test_gpus
    monkeypatch.chdir(tmp_path)
    with caplog.at_level(logging.INFO):
        test_gpus.run_tests("0,0", 2, 4)
    text = "\n".join(r.getMessage() for r in caplog.records)
    assert "GPU count: " in text and "Default GPU: 0" in text and "gfx950" in text
    assert text.count("Testing GPU: 0") == 4
    assert text.count("seconds to upscale sample.png") == 4
    assert text.count("Upscaled 1/1") == 4
    assert "seconds total to run tests." in text and "frames/s" in text
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "test_gpus_harness.txt"), "w") as f:
            f.write(text + "\n")


def test_harness_config5_sample_size(caplog, tmp_path, monkeypatch):
    """BASELINE config 5 names a 3840x2160 input on the worker list 0..7: `test_gpus.py --size 3840x2160` writes that
    sample; here two workers share the one GPU (`-g 0,0`), two runs, and the output frame has config 5's size"""
    import test_gpus
    monkeypatch.chdir(tmp_path)
    with caplog.at_level(logging.INFO):
        test_gpus.run_tests("0,0", 2, 2, size="3840x2160")
    text = "\n".join(r.getMessage() for r in caplog.records)
    assert text.count("Testing GPU: 0") == 2 and text.count("seconds to upscale sample.png") == 2
    assert "sample.png: a synthetic 3840x2160 frame" in text
    assert "frames/s" in text


def test_opportunistic_pin_against_real_ncnn_and_cv2(uva, oracle, oracle_models):
    """The oracle is pinned only by an independent restatement: the reference's
    arithmetic lives in the un-pinned ncnn_vulkan / opencv-python wheels, which this image does not have.
    If a box ever has them, compare: ncnn on the CPU (use_vulkan_compute = False) against the oracle's f32
    output on a golden input, and cv2.imwrite's float -> u8 conversion against the oracle's."""
    found = {}
    for mod in ("ncnn", "ncnn_vulkan", "cv2"):
        try:
            found[mod] = __import__(mod)
        except Exception as e:  # noqa: BLE001
            found[mod] = None
            print("import %s: %s: %s" % (mod, type(e).__name__, e))
    real = found["ncnn"] or (getattr(found["ncnn_vulkan"], "ncnn", None) if found["ncnn_vulkan"] else None)
    if real is None and found["cv2"] is None:
        pytest.skip("neither ncnn / ncnn_vulkan nor cv2 is importable on this box: the oracle stays pinned by "
                    "oracle/independent_check.py only (parity unpinned, DESIGN.md section 2)")
    g = np.load(os.path.join(ROOT, "tests", "golden", "independent_torch.npz"))
    if found["cv2"] is not None:
        cv2 = found["cv2"]
        f = (np.arange(0, 4096, dtype=np.float64).reshape(64, 64, 1) / 8.0 - 20.0).repeat(3, axis=2)   # .5 ties included
        path = "/tmp/uva_cv2_roundtrip.png"
        assert cv2.imwrite(path, f)
        from upscale_video_amd import _imageio
        assert np.array_equal(cv2.imread(path), _imageio.to_u8(f))
    if real is not None:
        from conftest import model_paths
        for key in ("2x", "1x"):
            net = real.Net()
            net.opt.use_vulkan_compute = False
            p, b = model_paths(key)
            net.load_param(p)
            net.load_model(b)
            tag = [k[:-3] for k in g.files if k.startswith(key) and k.endswith("_in")][0]
            img = g[tag + "_in"]
            mat = real.Mat.from_pixels(img, real.Mat.PixelType.PIXEL_BGR, img.shape[1], img.shape[0])
            mat.substract_mean_normalize([], [1 / 255.0] * 3)
            ex = net.create_extractor()
            ex.input("input", mat)
            ret, out = ex.extract("output")
            assert ret == 0
            got = np.array(out)
            want = oracle_models[key].forward(oracle.from_pixels_normalize(img))
            assert np.abs(got - want).max() <= 2e-3, (key, float(np.abs(got - want).max()))


@pytest.mark.gpu
def test_narrow_strip_kloop_equals_the_full_one():
    """Strips of <= 14 columns run trunk2_kernel's k-loop without the second fragment column (UVA_T2_NARROW=0 turns that
    off): every width 1..47, the 970-wide planes of the reference's tiling and a tiled frame give identical bytes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "narrow_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "differing: none" in r.stdout


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_one_rank(tmp_path):
    """the N > 1 code path of bench.py -- RCCL process group, the barrier + max-reduce around both
    timed regions, NUMA pinning, the per-rank host route -- run on the GPU box before the driver's scaling run does:
    `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`, one JSON line, and a frame rate within 10 % of
    the same command without the launcher."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--gpus", "1", "--steps", "60", "--warmup", "10", "--no-cpu-baseline", "--no-parity"]

    def last_json(out):
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out[-2000:]
        return json.loads(lines[0])

    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["steps"] == 60
    assert d["config"]["host_route_fps_pcie_inclusive"] > 0 and "roofline" in d and "parity" not in d
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    d2 = last_json(r2.stdout)
    assert abs(d["value"] - d2["value"]) <= 0.10 * d2["value"], (d["value"], d2["value"])


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` starts its two ranks itself (no torch.distributed.run, no RCCL: the frame
    queue has no collective and its timing fence is a multiprocessing barrier); `--devices 0,0` puts both on the box's one
    GPU, the reference's way of loading a GPU with several workers.  ONE JSON line with n_gpus 2, and -- two ranks sharing
    one GPU -- a whole-job rate in the neighbourhood of one rank's."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    common = ["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-parity"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0"] + common,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and "no RCCL" in d["config"]["launcher"]
    assert len(d["config"]["timed_regions_s"]) == 3 and d["config"]["host_route_fps_pcie_inclusive"] > 0
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=900, env=env)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    d1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][0])
    assert 0.6 * d1["value"] <= d["value"] <= 1.25 * d1["value"], (d["value"], d1["value"])


@pytest.mark.gpu
def test_bench_dynamic_queue_and_per_rank_records_on_the_box():
    """Round 6: `--dynamic` -- the ranks take frame indices off ONE shared counter -- and the per-rank records
    of the one JSON line, with two self-launched ranks on the box's one GPU: every frame done exactly once between them, every
    rank's own device, NUMA node, K / E rates, PCIe rates and first-frame time in the line, the summary beside them."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("UVA_BENCH_JOB_DIR", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0", "--dynamic", "--steps", "24",
                        "--warmup", "4", "--no-cpu-baseline", "--no-parity"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and "dynamic" in d["config"]["frame_queue"]
    pr = d["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and sum(x["frames_K"] for x in pr) == 2 * 24
    for x in pr:
        assert x["device"] == 0 and x["fps_K"] > 0 and x["fps_E"] > 0 and x["h2d_GBps"] > 1 and x["d2h_GBps"] > 1 and x["first_frame_ms"] > 0
    assert {"fps_K", "fps_E", "h2d_GBps", "d2h_GBps", "first_frame_ms", "frames_K"} <= set(d["per_rank_summary"])


@pytest.mark.gpu
def test_bench_batches_the_1x_net():
    """`--workload 1x_hurrdeblur_1080p` hands four frames per call to uva_net_process_u8_device_batch by default: one sub10_kernel
    launch per four frames in the kernel statistics, the roofline's FLOPs and the replayed traffic per LAUNCH, the one-frame-per-call
    rate of the same run beside it -- and the batch is the faster of the two"""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "1x_hurrdeblur_1080p", "--steps", "80", "--warmup", "8",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=dict(os.environ))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    rf = d["roofline"]
    assert d["config"]["frames_per_call"] == 4 and rf["frames_per_launch"] == pytest.approx(4.0) and rf["launches"] == 20
    assert rf["flops_per_launch"] == pytest.approx(4 * 85536 * 1080 * 1920)
    assert d["value"] > d["config"]["one_frame_per_call_fps"] > 0.5 * d["value"]
    assert d["parity"]["max_abs_lsb"] <= 1


@pytest.mark.gpu
def test_config5_line_eight_ranks_on_the_one_gpu():
    """BASELINE config 5's bench line -- `python bench.py --gpus 8 --workload 2x_compact_2160p` -- with all
    eight ranks on the box's one GPU (`--devices 0,0,0,0,0,0,0,0`): eight spawned processes, eight sets of page-locked rings
    (3 x (24.9 + 99.5) MB each), the multiprocessing fence around both timed regions, NUMA pinning eight times over, ONE JSON
    line with n_gpus 8.  What an 8-GPU node adds to this is seven more devices, not another code path."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--devices", "0,0,0,0,0,0,0,0",
                        "--workload", "2x_compact_2160p", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-parity"],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 4 and d["scaling"] == "weak"
    assert "3840x2160" in d["config"]["workload"] and d["config"]["frames_per_rank"] == 4
    assert d["config"]["pinned_host_bytes_all_ranks"] == 8 * 3 * (2160 * 3840 * 3 + 4320 * 7680 * 3)
    assert d["value"] > 20 and d["config"]["host_route_fps_pcie_inclusive"] > 10         # one GPU's worth, shared by eight ranks
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "bench_config5_eight_ranks_one_gpu.json"), "w") as f:
            f.write(lines[0] + "\n")


@pytest.mark.gpu
def test_a_rank_on_a_missing_device_ends_the_job_fast():
    """the failure the first real 8-GPU run is most likely to meet: a rank whose device does not exist (here device 63 of a
    one-GPU box).  The job must end with a non-zero code and a message within seconds of that rank dying -- not sit in the
    fence until the driver's timeout."""
    import subprocess
    import sys
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,63", "--steps", "10", "--warmup", "2",
                        "--no-cpu-baseline", "--no-parity"], capture_output=True, text=True, timeout=600, env=env)
    dt = time.monotonic() - t0
    assert r.returncode != 0, r.stdout[-2000:]
    assert "rank 1" in r.stderr and "fence was aborted" in r.stderr, r.stderr[-3000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert dt < 240, dt                        # two interpreter start-ups with torch + the healthy rank's set-up, no fence timeout (900 s)


def test_harness_config5_eight_workers(caplog, tmp_path, monkeypatch):
    """BASELINE config 5 as named -- `test_gpus.py -g 0,1,..,7` on a 3840x2160 sample -- with the eight workers on the one GPU:
    eight pool processes, eight nets, one "Testing GPU" / "seconds to upscale" pair per run"""
    import test_gpus
    monkeypatch.chdir(tmp_path)
    with caplog.at_level(logging.INFO):
        test_gpus.run_tests("0,0,0,0,0,0,0,0", 2, 8, size="3840x2160")
    text = "\n".join(r.getMessage() for r in caplog.records)
    assert text.count("Testing GPU: 0") == 8 and text.count("seconds to upscale sample.png") == 8
    assert "sample.png: a synthetic 3840x2160 frame" in text and "frames/s" in text
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "test_gpus_config5_eight_workers.txt"), "w") as f:
            f.write(text + "\n")


@pytest.mark.gpu
def test_bench_valar_workload(tmp_path):
    """BASELINE config 4 as named under bench.py's contract: `--workload 4x_valar_1080p` (random-init weights: the .bin is
    a missing blob upstream) prints one JSON line with the whole-graph roofline object and both routes."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "4x_valar_1080p", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"] == "frames/sec 4x_valar_1080p" and d["n_gpus"] == 1 and d["steps"] == 3
    assert 5.0 < d["value"] < 30.0 and d["config"]["frame_tflop"] == pytest.approx(74.93, abs=0.01)
    assert d["roofline"]["bound"] == "mfma" and 0.1 < d["roofline"]["frac"] < 1.0 and d["roofline"]["launches"] == 3 * 69 and d["roofline"]["launches_per_frame"] == 69
    k = d["config"]["kernel_ms_per_frame"]
    assert 0 < sum(k.values()) < d["ms_per_step"]
    assert d["config"]["host_route_fps_pcie_inclusive"] > 0 and "random-init" in d["data"]
    # a complete line: the CPU restatement timed on a crop beside it, and the kernel's HBM bytes per launch from the PMC summary
    assert d["cpu_baseline"]["kind"] == "port" and 0 < d["cpu_baseline"]["value"] < 0.05
    assert d["roofline"]["traffic"] is None or (5e8 < d["roofline"]["traffic"] < 3e9 and d["roofline"]["traffic_source"].startswith("profiles/"))


def _window_check(got, img, om_apply, rad, s, wins, win=24, max_lsb=2, min_psnr=50):
    """locality: an output window equals the oracle run on the window plus `rad` pixels of context"""
    h, w = img.shape[:2]
    for (y0, x0) in wins:
        cy0, cx0, cy1, cx1 = max(0, y0 - rad), max(0, x0 - rad), min(h, y0 + win + rad), min(w, x0 + win + rad)
        want = om_apply(np.ascontiguousarray(img[cy0:cy1, cx0:cx1]))[(y0 - cy0) * s:(y0 - cy0 + win) * s, (x0 - cx0) * s:(x0 - cx0 + win) * s]
        g = got[y0 * s:(y0 + win) * s, x0 * s:(x0 + win) * s]
        d = np.abs(g.astype(int) - want.astype(int))
        assert d.max() <= max_lsb and psnr_u8(g, want) >= min_psnr, ((y0, x0), int(d.max()), psnr_u8(g, want))


def test_config3_chain_at_1080p(uva, net2x, oracle_models, oracle):
    """BASELINE config 3 at its own size: 1920x1080 through the 1x HurrDeblur pass (whole frame, apply_model), the u8
    hop, and the 2x net (whole frame here, so that a window's context is bounded) -- on the device without the hop
    leaving HBM, equal to the host-route chain; windows at the corners, the centre and two edges against the oracle's
    chain on the window plus the two nets' combined receptive radius; the reference-tiled result differs from it only
    near the tile seams."""
    pre = load_net(uva, "1x")
    h, w = 1080, 1920
    img = oracle.synthetic_frame(h, w, seed=33)
    mid = pre.process_u8(img, tile_size=0)
    whole = net2x.process_u8(mid, tile_size=0)
    assert whole.shape == (2 * h, 2 * w, 3)
    o1, o2 = oracle_models["1x"], oracle_models["2x"]
    rad = pre.num_convs + net2x.num_convs
    wins = [(0, 0), (0, w - 24), (h - 24, 0), (h - 24, w - 24), (h // 2, w // 2), (500, 0), (h - 24, 950)]
    _window_check(whole, img, lambda c: o2.apply_model(o1.apply_model(c)), rad, 2, wins, max_lsb=3, min_psnr=48)
    tiled = net2x.process_u8(mid, tile_size=960, border=10)
    d = np.abs(tiled.astype(int) - whole.astype(int))
    ys, xs = np.nonzero(d.max(axis=2))
    assert d.max() <= 2 and ((np.abs(ys - 1920) <= 40) | (np.abs(xs - 1920) <= 40)).all()
    if torch is not None and torch.cuda.is_available():
        d_in = torch.from_numpy(img).cuda()
        d_mid = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
        d_out = torch.empty((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        pre.wait_for(net2x)
        pre.process_u8_device(d_in.data_ptr(), h, w, d_mid.data_ptr(), tile_size=0)
        net2x.wait_for(pre)
        net2x.process_u8_device(d_mid.data_ptr(), h, w, d_out.data_ptr(), tile_size=960, border=10)
        pre.synchronize()
        net2x.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), tiled)


def test_config3_every_sample_at_1080p(uva, net2x, oracle_models, oracle):
    """BASELINE config 3 as the reference runs it -- apply_model (1x, whole frame) -> u8 -> upscale_image (2x, 960/10 tiling) --
    EVERY output sample of the 3840x2160 result against the fp32 oracle's chain on the host cores (~15 s), with the
    structured-error question (no row or column of |diff| stands out: a seam, a strip fold, a plane edge would)."""
    if oracle.cpu_quota() < 12:
        pytest.skip("the fp32 oracle needs a many-core host for a whole 1080p frame")
    from parity_report import check_u8, fp32_bar
    pre = load_net(uva, "1x")
    img = oracle.synthetic_frame(1080, 1920, seed=34)
    mid = pre.process_u8(img, tile_size=0)
    out = net2x.process_u8(mid, tile_size=960, border=10)
    omid = oracle_models["1x"].apply_model(img)
    check_u8("config 3: 1x stage, WHOLE 1080p frame, every sample", mid, omid, vs="fp32 oracle", model="1x", route="whole", **fp32_bar("1x", "whole"))
    want = oracle_models["2x"].upscale_image(omid, tile_size=960, border=10)
    check_u8("config 3: 1x -> u8 -> 2x (960/10), WHOLE 1080p frame, every sample", out, want, vs="fp32 oracle chain", model="chain", route="tiled",
             **fp32_bar("chain", "tiled"))


def test_config5_every_sample_of_the_2160p_frame(net2x, oracle_models, oracle):
    """BASELINE config 5's frame, 3840x2160 -> 7680x4320 through the 960/10 tiling (12 planes, 5 seams, > 70 000 trunk steps):
    EVERY output sample against the fp32 oracle's upscale_image (~45 s of the host's cores), structured-error question included."""
    if oracle.cpu_quota() < 12:
        pytest.skip("the fp32 oracle needs a many-core host for a whole 2160p frame")
    from parity_report import check_u8, fp32_bar
    img = oracle.synthetic_frame(2160, 3840, seed=55)
    got = net2x.process_u8(img, tile_size=960, border=10)
    want = oracle_models["2x"].upscale_image(img, tile_size=960, border=10)
    check_u8("config 5: 2x (960/10), WHOLE 2160p frame, every sample", got, want, vs="fp32 oracle", model="2x", route="tiled", **fp32_bar("2x", "tiled"))


def test_config4_valar_at_1080p(uva, tmp_path):
    """BASELINE config 4 as named, once at full size under pytest (throughput: tools/valar_bench.py): 4x_Valar_v1,
    synthetic weights (the real ones are a missing blob upstream), 1920x1080 -> 7680x4320 with the reference tiling.
    Shape, finite and non-constant, and -- the tiling being a loop over independent planes -- the top-right tile's
    part of the frame equals that tile run as a frame of its own (upscale_image's crop and paste, :464-477)."""
    from oracle import generic_oracle as go
    from upscale_video_amd import upscale_processing as up
    from upscale_video_amd.synth import synthetic_frame
    param = os.path.join(ROOT, "models", "4x_Valar_v1.param")
    b = str(tmp_path / "4x_Valar_v1.bin")
    go.write_synthetic_bin(param, b, seed=1, gain=0.5)
    net = uva.Net()
    net.set_vulkan_device(0)
    assert net.load_param(param) == 0 and net.load_model(b) == 0, getattr(net, "last_error", "")
    h, w = 1080, 1920
    img = synthetic_frame(h, w, seed=4)
    out = net.process_u8(img, tile_size=960, border=10)
    assert out.shape == (4 * h, 4 * w, 3) and out.std() > 1
    assert np.array_equal(out, net.process_u8(img, tile_size=960, border=10)), "the same frame twice: different bytes"
    (y0, y1, x0, x1), (top, bottom, left, right) = up.tile_window(960, 1, 1, h, w)          # the 130 x 970 plane
    tile = net.process_u8(np.ascontiguousarray(img[y0 - top:y1 + bottom, x0 - left:x1 + right]), tile_size=0)
    assert np.array_equal(out[4 * y0:4 * y1, 4 * x0:4 * x1], tile[4 * top:4 * (top + y1 - y0), 4 * left:4 * (left + x1 - x0)])


def test_frame_pool_four_workers_at_1080p(tmp_path, oracle, oracle_models, monkeypatch):
    """The PNG route as config 5's harness line names it on one GPU, `-g 0,0,0,0`: eight 1920x1080 frames through
    upscale_frames on the persistent workers (GPU-deflated PNGs), every result decoded again and checked against the
    oracle on windows (corners, a tile seam, the centre); inputs gone, outputs complete."""
    from upscale_video_amd import upscale_processing as up
    from upscale_video_amd import _imageio
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(up, "PERSISTENT_WORKERS", True)
    h, w = 1080, 1920
    frames = {n: oracle.synthetic_frame(h, w, seed=700 + n) for n in range(1, 9)}
    for n, f in frames.items():
        _imageio.imwrite("%d.extract.png" % n, f)
    try:
        up.upscale_frames(1, 1, 8, "extract", 2, [0, 0, 0, 0], 0, os.path.join(ROOT, "models"), "x_Compact_Pretrain", "input", "output")
    finally:
        up.shutdown_workers()
    om = oracle_models["2x"]
    for n, f in frames.items():
        assert not os.path.exists("%d.extract.png" % n)
        out = _imageio.imread("%d.png" % n)
        assert out is not None and out.shape == (2 * h, 2 * w, 3)
        wins = [(0, 0), (h - 24, w - 24), (h // 2, w // 2 + 100)] if n > 1 else [(0, 0), (0, w - 24), (h - 24, 0), (h - 24, w - 24), (500, 1200), (100, 30)]
        _window_check(out, f, om.apply_model, 18, 2, wins)
