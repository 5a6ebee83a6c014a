"""Runs the REFERENCE'S OWN host code and records what it did -- TEST INFRASTRUCTURE ONLY, build container only.

/root/reference/upscale/upscale_processing.py is imported UNCHANGED from where it lies (never copied: importlib from
/root/reference, which does not exist on the GPU box) with three stand-in modules in sys.modules, because the real
ones are not installed here and cannot be (no network):

    cv2           imread / imwrite over PIL.  imwrite applies cv::Mat::convertTo(CV_8U): saturate_cast<uchar>(cvRound(v)),
                  cvRound = round-half-to-even (documented OpenCV semantics; the one arithmetic step of the path that
                  cv2 owns).  It also hands the float canvas it was given to a recorder.
    ncnn_vulkan   .ncnn = Net / Mat / Extractor / destroy_gpu_instance over oracle/independent_check.py's torch-CPU
                  evaluator of the .param/.bin (ncnn's published layer semantics, fp32), or -- BACKEND = "coords" -- a
                  net that is the nearest-neighbour upscale, which turns the reference's tiling into a map "which input
                  pixel does each output pixel come from, through which tile".
    wakepy        keep (never entered on this path).

What is then the reference's own code, executed: get_frames (:27-37), logging_callback (:40-51), init_worker's slot
arithmetic (:54-73), apply_model (:258-299), process_tile (:395-477), upscale_image (:480-542) -- their window rule,
the float64 canvas, the order of the ncnn calls, the log items, the removal of the input.  What is NOT the reference's:
the layer arithmetic behind Extractor.extract (the stand-in, pinned only to ncnn's published semantics) and PNG I/O.

    python oracle/ref_host_fixtures.py            # writes tests/golden/ref_host.npz + ref_host.json
    python oracle/ref_host_fixtures.py --swap     # the INTEGRATION.md section 1 claim: the same module with
                                                  # ncnn_vulkan.ncnn = upscale_video_amd.ncnn (needs an MI355X)

Fixtures are data: inputs are upscale_video_amd.synth.synthetic_frame(h, w, seed) (sha256 recorded), outputs are seam
bands, corners, windows, row/column sums and call logs.  No reference text is stored.
"""
import argparse
import hashlib
import importlib.util
import json
import logging
import multiprocessing
import os
import sys
import tempfile
import types

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

REFERENCE_FILE = "/root/reference/upscale/upscale_processing.py"
GOLDEN = os.path.join(_ROOT, "tests", "golden")
BAND = 12        # input pixels either side of a tile seam kept as a band (the border is 10)
CORNER = 16      # input pixels of each frame corner kept, float canvas values included


# --------------------------------------------------------------------------------------------------------------
# stand-in modules
# --------------------------------------------------------------------------------------------------------------
class Recorder:
    def __init__(self):
        self.reset()

    def reset(self):
        self.calls = []          # (name, args...) in the order the reference made them
        self.tiles = []          # the u8 arrays handed to Mat.from_pixels
        self.canvas = None       # the float array handed to cv2.imwrite
        self.written = None


REC = Recorder()
BACKEND = "torch"                # or "coords"


def saturate_u8(a):
    """cv::Mat::convertTo(CV_8U) of a float array: cvRound (half to even) then clamp"""
    return np.clip(np.rint(np.asarray(a, np.float64)), 0, 255).astype(np.uint8)


def make_cv2():
    from PIL import Image
    m = types.ModuleType("cv2")

    def imread(path, flags=None):
        if not os.path.exists(path):
            return None
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])

    def imwrite(path, img):
        a = np.asarray(img)
        REC.calls.append(("cv2.imwrite", os.path.basename(path), str(a.dtype), tuple(a.shape)))
        REC.canvas = a
        u8 = a if a.dtype == np.uint8 else saturate_u8(a)
        REC.written = u8
        Image.fromarray(np.ascontiguousarray(u8[:, :, ::-1])).save(path, compress_level=1)
        return True

    def _absent(*a, **k):
        raise NotImplementedError("not on the pinned path")

    m.imread, m.imwrite, m.UMat, m.fastNlMeansDenoisingColored = imread, imwrite, _absent, _absent
    return m


def make_ncnn():
    from oracle import independent_check as ic
    pkg = types.ModuleType("ncnn_vulkan")
    m = types.ModuleType("ncnn_vulkan.ncnn")

    class Mat:
        class PixelType:
            PIXEL_RGB = 1
            PIXEL_BGR = 2

        def __init__(self, a):
            self.a = a

        @staticmethod
        def from_pixels(arr, pixel_type, w, h):
            assert pixel_type == Mat.PixelType.PIXEL_BGR and arr.dtype == np.uint8 and arr.shape == (h, w, 3)
            REC.calls.append(("Mat.from_pixels", int(pixel_type), int(w), int(h)))
            REC.tiles.append(arr.copy())
            return Mat(np.ascontiguousarray(arr.transpose(2, 0, 1)).astype(np.float32))   # BGR kept: byte k -> plane k

        def substract_mean_normalize(self, mean_vals, norm_vals):
            REC.calls.append(("Mat.substract_mean_normalize", list(mean_vals), [float(v) for v in norm_vals]))
            for c in range(self.a.shape[0]):
                if mean_vals:
                    self.a[c] -= np.float32(mean_vals[c])
                if norm_vals:
                    self.a[c] *= np.float32(norm_vals[c])      # ncnn takes float*: the double 1/255.0 narrows to fp32

        def __array__(self, dtype=None, copy=None):
            return self.a if dtype is None else self.a.astype(dtype)

    class Extractor:
        def __init__(self, net):
            self.net, self.x = net, None

        def input(self, name, mat):
            REC.calls.append(("Extractor.input", name))
            self.x = mat
            return 0

        def extract(self, name):
            REC.calls.append(("Extractor.extract", name))
            x = self.x.a
            if BACKEND == "coords":
                s = self.net.scale_hint
                return 0, Mat(np.repeat(np.repeat(x, s, 1), s, 2))
            import torch
            with torch.no_grad():
                return 0, Mat(ic.forward(self.net.layers, self.net.params, x))

    class Net:
        def __init__(self):
            REC.calls.append(("Net",))
            self.opt = types.SimpleNamespace(use_vulkan_compute=False)
            self.layers = self.params = None
            self.scale_hint = 1

        def set_vulkan_device(self, i):
            REC.calls.append(("Net.set_vulkan_device", int(i), bool(self.opt.use_vulkan_compute)))

        def load_param(self, path):
            REC.calls.append(("Net.load_param", os.path.basename(path)))
            if os.path.exists(path):
                self.layers = ic.parse_param(path)
                self.scale_hint = int(os.path.basename(path)[0])
            return 0 if self.layers else -1

        def load_model(self, path):
            REC.calls.append(("Net.load_model", os.path.basename(path)))
            if self.layers and os.path.exists(path):
                self.params, used, size = ic.load_bin(self.layers, path)
                assert used == size
            return 0 if self.params else -1

        def create_extractor(self):
            REC.calls.append(("Net.create_extractor",))
            return Extractor(self)

    def destroy_gpu_instance():
        REC.calls.append(("destroy_gpu_instance",))

    m.Net, m.Mat, m.Extractor, m.destroy_gpu_instance = Net, Mat, Extractor, destroy_gpu_instance
    pkg.ncnn = m
    return pkg, m


def import_reference(ncnn_module=None):
    """-> the reference's module object, executed from its own file"""
    if not os.path.exists(REFERENCE_FILE):
        raise SystemExit("no /root/reference here: this script runs in the build container only")
    sys.modules["cv2"] = make_cv2()
    if ncnn_module is None:
        pkg, m = make_ncnn()
    else:
        pkg = types.ModuleType("ncnn_vulkan")
        pkg.ncnn = m = ncnn_module
    sys.modules["ncnn_vulkan"], sys.modules["ncnn_vulkan.ncnn"] = pkg, m
    wk = types.ModuleType("wakepy")
    wk.keep = types.SimpleNamespace(running=lambda *a, **k: None, presenting=lambda *a, **k: None)
    sys.modules["wakepy"] = wk
    spec = importlib.util.spec_from_file_location("reference_upscale_processing", REFERENCE_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# --------------------------------------------------------------------------------------------------------------
# the cases
# --------------------------------------------------------------------------------------------------------------
def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def coords_frame(h, w):
    """u8 BGR frame whose pixel (y, x) spells its own position: B = x & 255, G = y & 255, R = (x >> 8) | (y >> 8) << 4"""
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x & 255, y & 255, (x >> 8) | ((y >> 8) << 4)], -1).astype(np.uint8)


def decode_coords(px):
    b, g, r = (px[..., k].astype(np.int64) for k in range(3))
    return g + ((r >> 4) << 8), b + ((r & 15) << 8)         # y, x


def slot_table(ref):
    """init_worker (:54-73): which -g entry a spawned worker binds, from its Pool identity and workers_used"""
    rows = []
    proc = multiprocessing.current_process()
    saved = proc._identity
    models = os.path.join(_ROOT, "models")
    try:
        for gpus in ([0], [0, 1], [0, 0, 1], [3, 5, 7, 1]):
            for workers_used in (0, 3):
                for ident in range(workers_used + 1, workers_used + len(gpus) + 2):
                    proc._identity = (ident,)
                    REC.reset()
                    try:
                        ref.init_worker(gpus, workers_used, models, "x_Compact_Pretrain", 2, "input", "output")
                        dev = [c for c in REC.calls if c[0] == "Net.set_vulkan_device"][0]
                        rows.append({"gpus": gpus, "workers_used": workers_used, "identity": ident, "device": dev[1],
                                     "use_vulkan_compute": dev[2], "calls": [list(c) for c in REC.calls],
                                     "names": [ref.model_input_name, ref.model_output_name]})
                    except SystemExit as e:
                        rows.append({"gpus": gpus, "workers_used": workers_used, "identity": ident, "device": None, "exit": str(e)})
    finally:
        proc._identity = saved
    return rows


def window_cases(ref, tmp):
    """upscale_image (:480-542) on a frame that spells coordinates, through the nearest-neighbour net: the tiles the
    reference cut (window per tile, in call order) and proof that its paste puts every output pixel where it belongs."""
    global BACKEND
    BACKEND = "coords"
    out = []
    from PIL import Image
    for h, w, s in [(1080, 1920, 2), (2160, 3840, 2), (256, 256, 2), (965, 970, 2), (980, 1000, 2), (1000, 1940, 2),
                    (969, 1925, 4), (961, 971, 4), (10, 10, 2), (9, 2000, 2), (1930, 975, 1), (1920, 1921, 2), (2890, 9, 4)]:
        img = coords_frame(h, w)
        src = os.path.join(tmp, "1.extract.png")
        Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(src, compress_level=1)
        ref.net = sys.modules["ncnn_vulkan"].ncnn.Net()
        ref.net.scale_hint = s
        REC.reset()
        items = ref.upscale_image(src, os.path.join(tmp, "1.png"), s, 0, 1, 1, True)
        assert not os.path.exists(src)                                  # remove=True (:521-522)
        wins = []
        for t in REC.tiles:
            y0, x0 = decode_coords(t[0, 0])
            wins.append([int(y0), int(y0) + t.shape[0], int(x0), int(x0) + t.shape[1]])
        canvas = REC.canvas
        assert canvas.dtype == np.float64 and canvas.shape == (h * s, w * s, 3)      # np.zeros(output_shape) (:497)
        want = np.repeat(np.repeat(img, s, 0), s, 1)
        exact = bool((REC.written == want).all())
        out.append({"h": h, "w": w, "scale": s, "windows": wins, "paste_is_exact_nearest": exact,
                    "log": [[lv, str(msg)] for lv, msg in items]})
        assert exact, (h, w, s)
    BACKEND = "torch"
    return out


def seams(n, tile=960):
    return [k for k in range(tile, n, tile)]


def record_frame(arrs, key, canvas, u8, s):
    """bands around every seam (u8, full length), corners (u8 + float canvas), row / column sums of the whole result"""
    sh, sw, _ = u8.shape
    h, w = sh // s, sw // s
    for y in seams(h):
        a, b = max(0, y - BAND) * s, min(h, y + BAND) * s
        arrs["%s/rows_%d" % (key, y)] = u8[a:b].copy()
    for x in seams(w):
        a, b = max(0, x - BAND) * s, min(w, x + BAND) * s
        arrs["%s/cols_%d" % (key, x)] = u8[:, a:b].copy()
    c = min(CORNER, h, w) * s
    for name, sl in (("tl", (slice(0, c), slice(0, c))), ("tr", (slice(0, c), slice(sw - c, sw))),
                     ("bl", (slice(sh - c, sh), slice(0, c))), ("br", (slice(sh - c, sh), slice(sw - c, sw)))):
        arrs["%s/corner_%s_u8" % (key, name)] = u8[sl].copy()
        arrs["%s/corner_%s_f32" % (key, name)] = np.asarray(canvas[sl], np.float32)
    arrs["%s/rowsum" % key] = u8.sum(axis=1, dtype=np.int64)          # [sH][3]
    arrs["%s/colsum" % key] = u8.sum(axis=0, dtype=np.int64)          # [sW][3]
    arrs["%s/lattice" % key] = u8[5::16, 7::16].copy()                # every 16th sample of the whole frame, both axes


def net_cases(ref, tmp, arrs, meta):
    from PIL import Image
    from oracle import uvoracle
    from upscale_video_amd.synth import synthetic_frame
    ncnn = sys.modules["ncnn_vulkan"].ncnn
    models = os.path.join(_ROOT, "models")

    def load(scale, model_file):
        proc = multiprocessing.current_process()
        saved, proc._identity = proc._identity, (1,)
        try:
            ref.init_worker([0], 0, models, model_file, scale, "input", "output")
        finally:
            proc._identity = saved

    def put(img, name):
        p = os.path.join(tmp, name)
        Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(p, compress_level=1)
        return p

    # upscale_image: the hard-coded 960 / 10 tiling (:489, :409-427) on frames that meet every border branch
    for key, model, s, h, w, seed in [("up2_965x970", "2x", 2, 965, 970, 11), ("up2_980x1000", "2x", 2, 980, 1000, 12),
                                      ("up2_1000x1940", "2x", 2, 1000, 1940, 13), ("up4_970x962", "4x", 4, 970, 962, 14)]:
        img = synthetic_frame(h, w, seed=seed)
        load(s, uvoracle.MODEL_FILES[model][1:])
        src = put(img, "7.extract.png")
        REC.reset()
        items = ref.upscale_image(src, os.path.join(tmp, "7.png"), s, 3, 7, 9, True)
        assert REC.canvas.dtype == np.float64
        record_frame(arrs, key, REC.canvas, REC.written, s)
        meta[key] = {"fn": "upscale_image", "model": model, "scale": s, "h": h, "w": w, "seed": seed, "input_sha16": sha16(img),
                     "output_sha16": sha16(REC.written), "log": [[lv, str(m)] for lv, m in items],
                     "calls": [list(c) for c in REC.calls], "tiles": [list(t.shape) for t in REC.tiles],
                     "input_removed": not os.path.exists(src)}
        print(key, meta[key]["output_sha16"], flush=True)

    # apply_model: the whole frame through the 1x net (:258-299)
    for key, h, w, seed in [("am1_360x480", 360, 480, 15), ("am1_1080x1920", 1080, 1920, 16)]:
        img = synthetic_frame(h, w, seed=seed)
        load(1, uvoracle.MODEL_FILES["1x"][1:])
        src = put(img, "7.extract.png")
        REC.reset()
        items = ref.apply_model(src, os.path.join(tmp, "7.anime.png"), True)
        assert REC.canvas.dtype == np.float32                            # out.transpose(1, 2, 0) * 255 (:284)
        record_frame(arrs, key, REC.canvas, REC.written, 1)
        arrs[key + "/window_u8"] = REC.written[100:164, 200:264].copy()
        arrs[key + "/window_f32"] = np.asarray(REC.canvas[100:164, 200:264], np.float32)
        meta[key] = {"fn": "apply_model", "model": "1x", "scale": 1, "h": h, "w": w, "seed": seed, "input_sha16": sha16(img),
                     "output_sha16": sha16(REC.written), "log": [[lv, str(m).replace(tmp + os.sep, "")] for lv, m in items],
                     "calls": [list(c) for c in REC.calls], "input_removed": not os.path.exists(src)}
        print(key, meta[key]["output_sha16"], flush=True)

    # the chain of config 3: apply_model 1x -> PNG (u8) -> upscale_image 2x (:905-948 order of stages)
    h, w, seed = 970, 990, 17
    img = synthetic_frame(h, w, seed=seed)
    load(1, uvoracle.MODEL_FILES["1x"][1:])
    src = put(img, "7.extract.png")
    ref.apply_model(src, os.path.join(tmp, "7.anime.png"), True)
    mid = REC.written.copy()
    load(2, uvoracle.MODEL_FILES["2x"][1:])
    REC.reset()
    ref.upscale_image(os.path.join(tmp, "7.anime.png"), os.path.join(tmp, "7.png"), 2, 0, 7, 9, True)
    record_frame(arrs, "chain_970x990", REC.canvas, REC.written, 2)
    arrs["chain_970x990/mid_lattice"] = mid[3::8, 5::8].copy()
    meta["chain_970x990"] = {"fn": "apply_model+upscale_image", "model": "1x,2x", "scale": 2, "h": h, "w": w, "seed": seed,
                             "input_sha16": sha16(img), "mid_sha16": sha16(mid), "output_sha16": sha16(REC.written)}
    print("chain", meta["chain_970x990"]["output_sha16"], flush=True)

    # the error path: a net whose extract raises (:454-459, :289-293)
    class Boom(Exception):
        pass

    def boom(self, name):
        raise Boom("device lost")
    keep = ncnn.Extractor.extract
    ncnn.Extractor.extract = boom
    try:
        src = put(synthetic_frame(32, 40, seed=1), "9.extract.png")
        REC.reset()
        items = ref.upscale_image(src, os.path.join(tmp, "9.png"), 2, 0, 9, 9, True)
        meta["error_upscale_image"] = {"log": [[lv, type(m).__name__ if isinstance(m, Exception) else str(m)] for lv, m in items],
                                       "destroyed": ("destroy_gpu_instance",) in REC.calls, "input_kept": os.path.exists(src),
                                       "output_written": os.path.exists(os.path.join(tmp, "9.png"))}
        load(1, uvoracle.MODEL_FILES["1x"][1:])
        REC.reset()
        items = ref.apply_model(src, os.path.join(tmp, "9.anime.png"), True)
        meta["error_apply_model"] = {"log": [[lv, type(m).__name__ if isinstance(m, Exception) else str(m)] for lv, m in items],
                                     "destroyed": ("destroy_gpu_instance",) in REC.calls, "input_kept": os.path.exists(src),
                                     "output_written": os.path.exists(os.path.join(tmp, "9.anime.png"))}
    finally:
        ncnn.Extractor.extract = keep


def small_cases(ref):
    """get_frames (:27-37), logging_callback (:40-51)"""
    out = {"get_frames": {s: ref.get_frames(s) for s in ("7", "1,4-6,9", "3-3", "10-12,1-2", "5,5")}}
    rows = []
    for items in ([["info", "a"], ["debug", "b"]], [["info", "ok"], ["error", "boom"], ["info", "after"]], [["error", "first"], ["error", "second"]], []):
        seen = []

        class H(logging.Handler):
            def emit(self, r):
                seen.append([r.levelname, r.getMessage()])
        hd = H()
        lg = logging.getLogger()
        old = lg.level
        lg.addHandler(hd)
        lg.setLevel(logging.DEBUG)
        try:
            ref.logging_callback(items)
            ex = None
        except SystemExit as e:
            ex = str(e)
        finally:
            lg.removeHandler(hd)
            lg.setLevel(old)
        rows.append({"items": items, "logged": seen, "exit": ex})
    out["logging_callback"] = rows
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--swap", action="store_true", help="ncnn_vulkan.ncnn = upscale_video_amd.ncnn (needs an MI355X and /root/reference)")
    args = ap.parse_args()
    import torch
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    if args.swap:
        from upscale_video_amd import ncnn as ours
        ref = import_reference(ours)
        raise SystemExit("--swap: run tests/test_ref_host.py::test_reference_module_runs_on_our_ncnn instead")
    ref = import_reference()
    arrs, meta = {}, {"reference_file": REFERENCE_FILE, "reference_sha16": hashlib.sha256(open(REFERENCE_FILE, "rb").read()).hexdigest()[:16],
                      "band": BAND, "corner": CORNER}
    meta.update(small_cases(ref))
    meta["init_worker"] = slot_table(ref)
    with tempfile.TemporaryDirectory() as tmp:
        meta["windows"] = window_cases(ref, tmp)
        net_cases(ref, tmp, arrs, meta)
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, "ref_host.npz"), **arrs)
    with open(os.path.join(GOLDEN, "ref_host.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(GOLDEN, "ref_host.npz"), os.path.getsize(os.path.join(GOLDEN, "ref_host.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
