"""Parity against the REFERENCE'S OWN HOST CODE, executed.

tests/golden/ref_host.npz|json were written by oracle/ref_host_fixtures.py in the build container: it imports
/root/reference/upscale/upscale_processing.py unchanged under stand-in cv2 / ncnn_vulkan / wakepy modules and runs the
reference's get_frames, logging_callback, init_worker, apply_model, process_tile and upscale_image.  What the fixture
pins is therefore the reference's host logic (the 960 / 10 window rule and paste, the float64 canvas, the worker-slot
arithmetic, the order and arguments of the ncnn calls, the log items, input removal); the layer arithmetic behind
Extractor.extract in that run is the torch evaluation of the .param/.bin (ncnn's published semantics), not ncnn itself.

CPU tests (here): the fixture against the host mirror (upscale_video_amd/upscale_processing.py) and against oracle.c.
GPU tests (-m gpu): the HIP path through the C ABI against the fixture.
One more CPU test runs only where /root/reference exists (the build container): the reference's module with
`ncnn_vulkan.ncnn = upscale_video_amd.ncnn` -- INTEGRATION.md section 1's one-line swap, executed -- over a stand-in
for libuva.so that evaluates with the oracle (tests/standin_libuva.py; the GPU box has no /root/reference).
"""
import hashlib
import json
import logging
import math
import multiprocessing
import os
import sys
import types

import numpy as np
import pytest

from conftest import ROOT, load_net
from parity_report import check_u8, record

GOLD = os.path.join(ROOT, "tests", "golden")
REFERENCE_FILE = "/root/reference/upscale/upscale_processing.py"
VS = "reference host code, executed (ref_host.npz)"


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(GOLD, "ref_host.json")))


@pytest.fixture(scope="module")
def arrs():
    return np.load(os.path.join(GOLD, "ref_host.npz"))


def _sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# ------------------------------------------------------------------------------------------------------------------
# host mirror against the reference's recorded behaviour (CPU)
# ------------------------------------------------------------------------------------------------------------------
def test_fixture_is_current(meta):
    """where the reference is present: the fixture was generated from the file that lies there now"""
    if not os.path.exists(REFERENCE_FILE):
        pytest.skip("no /root/reference on this box")
    assert hashlib.sha256(open(REFERENCE_FILE, "rb").read()).hexdigest()[:16] == meta["reference_sha16"]


def test_get_frames_as_the_reference(meta):
    from upscale_video_amd import upscale_processing as up
    for arg, want in meta["get_frames"].items():
        assert up.get_frames(arg) == want, arg


def test_logging_callback_as_the_reference(meta):
    """levels, the exit after the FIRST error item (items behind it are never logged, reference :40-51)"""
    from upscale_video_amd import upscale_processing as up
    for row in meta["logging_callback"]:
        seen = []

        class H(logging.Handler):
            def emit(self, r):
                seen.append([r.levelname, r.getMessage()])
        hd, lg = H(), logging.getLogger()
        old = lg.level
        lg.addHandler(hd)
        lg.setLevel(logging.DEBUG)
        try:
            up.logging_callback(row["items"])
            ex = None
        except SystemExit as e:
            ex = str(e)
        finally:
            lg.removeHandler(hd)
            lg.setLevel(old)
        assert seen == row["logged"] and ex == row["exit"], row


class _RecNcnn(types.ModuleType):
    """a recording ncnn surface for the mirror: nearest-neighbour net, the call log in the fixture's format"""

    def __init__(self, scale=2):
        super().__init__("rec_ncnn")
        calls = self.calls = []
        mod = self

        class Mat:
            class PixelType:
                PIXEL_RGB, PIXEL_BGR = 1, 2

            def __init__(self, a):
                self.a = a

            @staticmethod
            def from_pixels(arr, pt, w, h):
                calls.append(["Mat.from_pixels", int(pt), int(w), int(h)])
                mod.tiles.append(arr.copy())
                return Mat(np.ascontiguousarray(arr.transpose(2, 0, 1)).astype(np.float32))

            def substract_mean_normalize(self, mean_vals, norm_vals):
                calls.append(["Mat.substract_mean_normalize", list(mean_vals), [float(v) for v in norm_vals]])
                self.norm = [np.float32(v) for v in norm_vals]
                for c in range(3):
                    self.a[c] *= self.norm[c]

            def __array__(self, dtype=None, copy=None):
                return self.a

        class Extractor:
            def __init__(self, net):
                self.net = net

            def input(self, name, mat):
                calls.append(["Extractor.input", name])
                self.x = mat
                return 0

            def extract(self, name):
                calls.append(["Extractor.extract", name])
                s = mod.scale
                return 0, Mat(np.repeat(np.repeat(self.x.a, s, 1), s, 2))

        class Net:
            def __init__(self):
                calls.append(["Net"])
                self.opt = types.SimpleNamespace(use_vulkan_compute=False)

            def set_vulkan_device(self, i):
                calls.append(["Net.set_vulkan_device", int(i), bool(self.opt.use_vulkan_compute)])

            def load_param(self, p):
                calls.append(["Net.load_param", os.path.basename(p)])
                return 0

            def load_model(self, p):
                calls.append(["Net.load_model", os.path.basename(p)])
                return 0

            def create_extractor(self):
                calls.append(["Net.create_extractor"])
                return Extractor(self)

        self.Mat, self.Extractor, self.Net, self.scale, self.tiles = Mat, Extractor, Net, scale, []
        self.destroy_gpu_instance = lambda: calls.append(["destroy_gpu_instance"])


def test_worker_slots_and_net_construction_as_the_reference(meta, monkeypatch):
    """init_worker (:54-73): Pool identity - 1 - workers_used picks the -g entry; Net(), use_vulkan_compute = True,
    set_vulkan_device, load_param, load_model in that order with those file names; blob names stored"""
    from upscale_video_amd import upscale_processing as up
    proc = multiprocessing.current_process()
    for row in meta["init_worker"]:
        rec = _RecNcnn()
        monkeypatch.setattr(up, "ncnn", rec)
        monkeypatch.setattr(proc, "_identity", (row["identity"],), raising=False)
        up.init_worker(row["gpus"], row["workers_used"], os.path.join(ROOT, "models"), "x_Compact_Pretrain", 2, "input", "output")
        if row["device"] is None:          # the reference exits the worker; the mirror keeps it alive with an error (see its docstring)
            assert up.net is None and up.init_error == "Unable to assign GPU to new worker.", row
            assert rec.calls == []
        else:
            assert up.net is not None and up.init_error is None, row
            assert rec.calls == row["calls"], (rec.calls, row["calls"])
            assert [up.model_input_name, up.model_output_name] == row["names"]
    up.net = None


def _coords_frame(h, w):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x & 255, y & 255, (x >> 8) | ((y >> 8) << 4)], -1).astype(np.uint8)


def _decode(px):
    b, g, r = (int(px[k]) for k in range(3))
    return g + ((r >> 4) << 8), b + ((r & 15) << 8)


def test_tile_windows_and_paste_as_the_reference(meta, monkeypatch, tmp_path):
    """upscale_image / process_tile (:395-542) on frames that spell their coordinates, through a nearest-neighbour net:
    the mirror cuts the windows the reference cut, in the reference's order, makes the same ncnn calls per tile, pastes
    every output pixel where the reference put it, logs the same items and removes its input"""
    from upscale_video_amd import upscale_processing as up, _imageio
    monkeypatch.setattr(up, "FUSED_DEVICE_PATH", False)
    for case in meta["windows"]:
        h, w, s = case["h"], case["w"], case["scale"]
        # tile_window alone, every tile of the grid
        got = []
        for ty in range(math.ceil(h / 960)):
            for tx in range(math.ceil(w / 960)):
                (y0, y1, x0, x1), (t, b, l, r) = up.tile_window(960, ty, tx, h, w)
                got.append([y0 - t, y1 + b, x0 - l, x1 + r])
        assert got == case["windows"], (h, w)
        if h * w > 2200 * 2200:
            continue                                   # the paste of the largest frames: tile_window above is the rule
        rec = _RecNcnn(s)
        monkeypatch.setattr(up, "ncnn", rec)
        up.net, up.init_error = rec.Net(), None
        up.model_input_name, up.model_output_name = "input", "output"
        img = _coords_frame(h, w)
        src, dst = str(tmp_path / "1.extract.png"), str(tmp_path / "1.png")
        _imageio.imwrite(src, img)
        items = up.upscale_image(src, dst, s, 0, 1, 1, True)
        assert [[lv, str(m)] for lv, m in items] == case["log"]
        assert not os.path.exists(src)
        wins = [[*(_decode(t[0, 0])), t.shape[0], t.shape[1]] for t in rec.tiles]
        assert [[y, y + hh, x, x + ww] for y, x, hh, ww in wins] == case["windows"]
        out = _imageio.imread(dst)
        assert np.array_equal(out, np.repeat(np.repeat(img, s, 0), s, 1)) == case["paste_is_exact_nearest"]
    up.net = None


def test_ncnn_call_sequence_as_the_reference(meta, monkeypatch, tmp_path):
    """per tile: from_pixels(PIXEL_BGR, w, h), substract_mean_normalize([], [1/255] * 3), create_extractor, input("input"),
    extract("output") -- and apply_model's single pass (:265-281) ends in imwrite of a float32 [h][w][3] array"""
    from upscale_video_amd import upscale_processing as up, _imageio
    monkeypatch.setattr(up, "FUSED_DEVICE_PATH", False)
    case = meta["up2_965x970"]
    rec = _RecNcnn(2)
    monkeypatch.setattr(up, "ncnn", rec)
    up.net, up.init_error = rec.Net(), None
    rec.calls.clear()
    src = str(tmp_path / "7.extract.png")
    _imageio.imwrite(src, np.zeros((case["h"], case["w"], 3), np.uint8))
    items = up.upscale_image(src, str(tmp_path / "7.png"), 2, 3, 7, 9, True)
    assert [[lv, str(m)] for lv, m in items] == case["log"]
    want = [c for c in case["calls"] if c[0] != "cv2.imwrite"]
    assert rec.calls == want
    assert [list(t.shape) for t in rec.tiles] == case["tiles"]
    # apply_model
    case = meta["am1_360x480"]
    rec = _RecNcnn(1)
    monkeypatch.setattr(up, "ncnn", rec)
    up.net = rec.Net()
    rec.calls.clear()
    _imageio.imwrite(src, np.zeros((case["h"], case["w"], 3), np.uint8))
    written = []
    monkeypatch.setattr(up, "imwrite", lambda p, a: written.append((str(a.dtype), list(a.shape))))
    items = up.apply_model(src, str(tmp_path / "7.anime.png"), True)
    assert rec.calls == [c for c in case["calls"] if c[0] != "cv2.imwrite"]
    assert written == [(c[2], c[3]) for c in case["calls"] if c[0] == "cv2.imwrite"]
    assert items[-1][0] == "info" and items[-1][1].startswith("Processed Model: ") and items[-1][1].endswith("7.anime.png")
    assert case["log"][-1][1].endswith("7.anime.png") and case["input_removed"] and not os.path.exists(src)
    up.net = None


def test_error_path_as_the_reference(meta, monkeypatch, tmp_path):
    """an extract that raises (:289-293, :454-459): two error items, destroy_gpu_instance, the input stays, nothing written"""
    from upscale_video_amd import upscale_processing as up, _imageio
    monkeypatch.setattr(up, "FUSED_DEVICE_PATH", False)
    rec = _RecNcnn(2)

    class Boom(Exception):
        pass

    def boom(self, name):
        raise Boom("device lost")
    rec.Extractor.extract = boom
    monkeypatch.setattr(up, "ncnn", rec)
    up.net, up.init_error = rec.Net(), None
    src = str(tmp_path / "9.extract.png")
    _imageio.imwrite(src, np.zeros((32, 40, 3), np.uint8))
    for fn, key, dst in ((lambda: up.upscale_image(src, str(tmp_path / "9.png"), 2, 0, 9, 9, True), "error_upscale_image", "9.png"),
                         (lambda: up.apply_model(src, str(tmp_path / "9.anime.png"), True), "error_apply_model", "9.anime.png")):
        rec.calls.clear()
        items = fn()
        want = meta[key]
        assert [[lv, type(m).__name__ if isinstance(m, Exception) else str(m)] for lv, m in items] == want["log"]
        assert (["destroy_gpu_instance"] in rec.calls) == want["destroyed"]
        assert os.path.exists(src) == want["input_kept"] and os.path.exists(str(tmp_path / dst)) == want["output_written"]
    up.net = None


# ------------------------------------------------------------------------------------------------------------------
# arithmetic: oracle.c (CPU) and the HIP path (GPU) against the frames the reference's functions produced
# ------------------------------------------------------------------------------------------------------------------
CASES = ["up2_965x970", "up2_980x1000", "up2_1000x1940", "up4_970x962", "am1_360x480", "am1_1080x1920", "chain_970x990"]


def _pieces(arrs, key):
    pre = key + "/"
    return {k[len(pre):]: arrs[k] for k in arrs.files if k.startswith(pre)}


def _input(meta, key):
    from upscale_video_amd.synth import synthetic_frame
    c = meta[key]
    img = synthetic_frame(c["h"], c["w"], seed=c["seed"])
    assert _sha16(img) == c["input_sha16"], "synthetic_frame changed: regenerate the fixture"
    return img


def _compare(name, out, pieces, s, band, corner, max_lsb, max_share, sum_tol, model, route, vs=VS):
    """`out` (whole u8 result) against the recorded seam bands, corners, lattice and row / column sums"""
    sh, sw, _ = out.shape
    h, w = sh // s, sw // s
    for k, want in pieces.items():
        if k.startswith("rows_"):
            y = int(k[5:])
            got = out[max(0, y - band) * s:min(h, y + band) * s]
        elif k.startswith("cols_"):
            x = int(k[5:])
            got = out[:, max(0, x - band) * s:min(w, x + band) * s]
        elif k.startswith("corner_") and k.endswith("_u8"):
            c = min(corner, h, w) * s
            ys = slice(0, c) if k[7] == "t" else slice(sh - c, sh)
            xs = slice(0, c) if k[8] == "l" else slice(sw - c, sw)
            got = out[ys, xs]
        elif k == "lattice":
            got = out[5::16, 7::16]
        elif k == "window_u8":
            got = out[100:164, 200:264]
        else:
            continue
        check_u8(f"{name} {k}", np.ascontiguousarray(got), want, vs=vs, max_lsb=max_lsb, max_share=max_share, model=model, route=route,
                 structure=k.startswith(("rows_", "cols_")))
    # the whole frame, with its spatial structure: per-row and per-column sums of all samples
    for k, axis in (("rowsum", 1), ("colsum", 0)):
        d = np.abs(out.sum(axis=axis, dtype=np.int64) - pieces[k])
        n = out.shape[axis]
        record(f"{name} {k}", kind="f32", vs=vs, model=model, route=route, what=k, max_abs_err=float(d.max()), rel_to_range=float(d.max()) / n,
               bar_max_abs=float(sum_tol(n)))
        assert d.max() <= sum_tol(n), (name, k, int(d.max()), "at", int(d.argmax()), "bar", sum_tol(n))


def _oracle_result(oracle_models, key, case, img):
    if case["fn"] == "upscale_image":
        return oracle_models[case["model"]].upscale_image(img, 960, 10)
    if case["fn"] == "apply_model":
        return oracle_models["1x"].apply_model(img)
    return oracle_models["2x"].upscale_image(oracle_models["1x"].apply_model(img), 960, 10)


@pytest.mark.parametrize("key", CASES)
def test_oracle_against_the_reference_run(meta, arrs, oracle_models, key):
    """oracle.c (fp32) gives what the reference's upscale_image / apply_model gave over the torch evaluation: equal except
    where the two fp32 summation orders land either side of a .5 tie (<= 1 level on <= 0.01 % of the samples; the chain 0.1 %)"""
    case, img = meta[key], _input(meta, key)
    out = _oracle_result(oracle_models, key, case, img)
    # (the chain: a .5 tie that falls the other way in the 1x stage's u8 frame moves the 2x net's input by one level)
    _compare(f"oracle.c {key}", out, _pieces(arrs, key), case["scale"], meta["band"], meta["corner"], max_lsb=1,
             max_share=1e-3 if "chain" in key else 1e-4, sum_tol=lambda n: 8 if "chain" in key else 4, model=case["model"].split(",")[-1], route="tiled" if "up" in case["fn"] else "whole")


def test_oracle_float_output_against_the_reference_canvas(meta, arrs, oracle, oracle_models):
    """before quantisation: oracle.c's forward * 255 against the float array the reference handed to cv2.imwrite (apply_model)"""
    for key in ("am1_360x480",):
        img = _input(meta, key)
        f = oracle_models["1x"].forward(oracle.from_pixels_normalize(img)).transpose(1, 2, 0) * np.float32(255)
        p = _pieces(arrs, key)
        assert np.abs(f[100:164, 200:264] - p["window_f32"]).max() <= 2e-3
        c = meta["corner"]
        assert np.abs(f[:c, :c] - p["corner_tl_f32"]).max() <= 2e-3 and np.abs(f[-c:, -c:] - p["corner_br_f32"]).max() <= 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("key", CASES)
def test_hip_against_the_reference_run(meta, arrs, uva, key):
    """the HIP path through the C ABI (uva_net_process_u8, 960 / 10) against the reference's own upscale_image / apply_model
    output: <= 1 level (the chain: 2), the share of differing samples inside the measured fp32 bars, and the per-row /
    per-column sums of the whole frame within 6 sigma of what that share makes of them -- a wrong row or column shows there"""
    from parity_report import fp32_bar
    case, img = meta[key], _input(meta, key)
    if case["fn"] == "upscale_image":
        out = load_net(uva, case["model"]).process_u8(img, tile_size=960, border=10)
        bar = fp32_bar(case["model"], "tiled")
    elif case["fn"] == "apply_model":
        out = load_net(uva, "1x").process_u8(img, tile_size=0)
        bar = fp32_bar("1x", "whole")
    else:
        mid = load_net(uva, "1x").process_u8(img, tile_size=0)
        out = load_net(uva, "2x").process_u8(mid, tile_size=960, border=10)
        bar = fp32_bar("chain", "tiled")
    share = bar.get("max_share", 0.1)
    _compare(f"HIP {key}", out, _pieces(arrs, key), case["scale"], meta["band"], meta["corner"], max_lsb=min(bar["max_lsb"], 2 if "chain" in key else 1),
             max_share=share, sum_tol=lambda n: 3 * n * share / 2 + 6 * math.sqrt(3 * n * share), model=case["model"].split(",")[-1],
             route="tiled" if "up" in case["fn"] else "whole")


# ------------------------------------------------------------------------------------------------------------------
# INTEGRATION.md section 1, executed: the reference's module on upscale_video_amd.ncnn
# ------------------------------------------------------------------------------------------------------------------
def test_reference_module_runs_on_our_ncnn(meta, arrs, monkeypatch, tmp_path, oracle_models):
    """`from ncnn_vulkan import ncnn` -> upscale_video_amd.ncnn, nothing else changed: the reference's own init_worker,
    apply_model and upscale_image run on the product's Net / Mat / Extractor.  No GPU here, so libuva.so is replaced by
    tests/standin_libuva.py (the oracle behind the same C entry points); what is exercised is the Python surface the
    swap relies on: names, argument forms, return conventions, np.array(mat), error paths."""
    if not os.path.exists(REFERENCE_FILE):
        pytest.skip("no /root/reference on this box")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_host_fixtures as rf
    import standin_libuva
    from upscale_video_amd import _lib, ncnn as ours
    monkeypatch.setattr(_lib, "_lib", standin_libuva.Lib())
    keep = {k: sys.modules.get(k) for k in ("cv2", "ncnn_vulkan", "ncnn_vulkan.ncnn", "wakepy")}
    try:
        ref = rf.import_reference(ours)
        proc = multiprocessing.current_process()
        monkeypatch.setattr(proc, "_identity", (1,), raising=False)
        from PIL import Image

        def put(img, name):
            Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(str(tmp_path / name))
            return str(tmp_path / name)
        # 2x, tiled: a 965x970 frame has all four border branches
        key = "up2_965x970"
        img = _input(meta, key)
        ref.init_worker([0], 0, os.path.join(ROOT, "models"), "x_Compact_Pretrain", 2, "input", "output")
        assert isinstance(ref.net, ours.Net) and ref.net.scale == 2
        src = put(img, "7.extract.png")
        items = ref.upscale_image(src, str(tmp_path / "7.png"), 2, 3, 7, 9, True)
        assert [[lv, str(m)] for lv, m in items] == meta[key]["log"] and not os.path.exists(src)
        out = np.ascontiguousarray(np.asarray(Image.open(str(tmp_path / "7.png")).convert("RGB"))[:, :, ::-1])
        _compare("reference module on upscale_video_amd.ncnn " + key, out, _pieces(arrs, key), 2, meta["band"], meta["corner"], max_lsb=1,
                 max_share=1e-4, sum_tol=lambda n: 4, model="2x", route="tiled", vs="reference module over the stand-in torch net")
        # 1x, whole frame
        key = "am1_360x480"
        img = _input(meta, key)
        ref.init_worker([0], 0, os.path.join(ROOT, "models"), "x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g", 1, "input", "output")
        src = put(img, "7.extract.png")
        items = ref.apply_model(src, str(tmp_path / "7.anime.png"), True)
        assert items[-1][0] == "info" and not os.path.exists(src)
        out = np.ascontiguousarray(np.asarray(Image.open(str(tmp_path / "7.anime.png")).convert("RGB"))[:, :, ::-1])
        _compare("reference module on upscale_video_amd.ncnn " + key, out, _pieces(arrs, key), 1, meta["band"], meta["corner"], max_lsb=1,
                 max_share=1e-4, sum_tol=lambda n: 4, model="1x", route="whole", vs="reference module over the stand-in torch net")
        # a failing extract: the product's error becomes the reference's error items and destroy_gpu_instance is callable
        standin_libuva.FAIL_EXTRACT = True
        src = put(img, "9.extract.png")
        items = ref.apply_model(src, str(tmp_path / "9.anime.png"), True)
        assert items[0] == ["error", "Model processing failed"] and isinstance(items[1][1], Exception) and os.path.exists(src)
    finally:
        standin_libuva.FAIL_EXTRACT = False
        for k, v in keep.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
