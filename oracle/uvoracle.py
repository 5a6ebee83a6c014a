"""ctypes wrapper around oracle/_build/liboracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the
checker.  The product package (upscale_video_amd/) never imports this module.

PARITY UNPINNED (see oracle.c header): ncnn_vulkan / cv2 cannot run here and the reference
holds no golden vectors; the restatement is cross-checked against oracle/independent_check.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

F16_STORAGE = 1
F16_INPUT = 4        # with F16_STORAGE: the float route rounds the caller's normalised input to fp16
WINOGRAD_F23 = 2     # with F16_STORAGE: fused trunk pairs as 1-D Winograd F(2,3) with the HIP kernel's rounding points
PRELU_F16 = 8        # with WINOGRAD_F23: the PReLU behind those convolutions on fp16 values (trunkw_kernel's TW_ACT_F16 modes)


def product_flags(route="u8"):
    """The storage mode that restates the HIP path's rounding points: fp16 storage, and -- unless the product is run with
    UVA_TRUNK_WINO=0 (direct convolution, trunk2_kernel) or UVA_TRUNK_FUSION=0 -- the fused 64 -> 64 trunk pairs as
    Winograd F(2,3) (trunkw_kernel) with their PReLU on fp16 values (unless UVA_TW_ACT16=0).  No effect on the 24-feature net."""
    dbg = os.environ.get("UVA_DEBUG_SWITCHES") == "1"      # the library ignores its switches without the opt-in; so does this
    sw = (lambda name: os.environ.get(name, "1")) if dbg else (lambda name: "1")
    wino = sw("UVA_TRUNK_WINO") != "0" and sw("UVA_TRUNK_FUSION") != "0"
    act16 = wino and sw("UVA_TW_ACT16") != "0"
    return F16_STORAGE | (WINOGRAD_F23 if wino else 0) | (PRELU_F16 if act16 else 0) | (F16_INPUT if route == "f32" else 0)


def build(force=False):
    """Compile oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        c_p, c_i, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.uvo_load.restype = c_p
        L.uvo_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, c_i]
        L.uvo_free.argtypes = [c_p]
        for name in ("uvo_scale", "uvo_nf", "uvo_num_conv", "uvo_num_layers"):
            getattr(L, name).restype = c_i
            getattr(L, name).argtypes = [c_p]
        for name in ("uvo_bin_size", "uvo_bin_consumed"):
            getattr(L, name).restype = ctypes.c_size_t
            getattr(L, name).argtypes = [c_p]
        L.uvo_conv_info.restype = c_i
        L.uvo_conv_info.argtypes = [c_p, c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i),
                                    ctypes.POINTER(ctypes.c_uint32)]
        L.uvo_conv_weights.restype = ctypes.POINTER(c_f)
        L.uvo_conv_weights.argtypes = [c_p, c_i]
        L.uvo_conv_bias.restype = ctypes.POINTER(c_f)
        L.uvo_conv_bias.argtypes = [c_p, c_i]
        L.uvo_prelu_slopes.restype = ctypes.POINTER(c_f)
        L.uvo_prelu_slopes.argtypes = [c_p, c_i, ctypes.POINTER(c_i)]
        L.uvo_forward_f32.restype = c_i
        L.uvo_forward_f32.argtypes = [c_p, c_p, c_i, c_i, c_p, c_i, c_i]
        L.uvo_forward_tap.restype = c_i
        L.uvo_forward_tap.argtypes = [c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_i]
        L.uvo_from_pixels_normalize.argtypes = [c_p, c_i, c_i, ctypes.c_size_t, c_p]
        L.uvo_to_u8.argtypes = [c_p, c_i, c_i, c_p, ctypes.c_size_t]
        L.uvo_apply_model_u8.restype = c_i
        L.uvo_apply_model_u8.argtypes = [c_p, c_p, c_i, c_i, c_p, c_i, c_i]
        L.uvo_upscale_image_u8.restype = c_i
        L.uvo_upscale_image_u8.argtypes = [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i]
        L.uvo_round_f16.restype = c_f
        L.uvo_round_f16.argtypes = [c_f]
        L.uvo_max_threads.restype = c_i
        _lib = L
    return _lib


def cpu_quota():
    """CPUs this process may actually use: the affinity mask capped by the container's CFS quota (cgroup v2
    cpu.max / v1 cpu.cfs_quota_us).  The GPU boxes show 256 CPUs and grant 16."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:  # noqa: BLE001
            pass
    return n


def max_threads():
    """OpenMP threads for the oracle: what the host grants, not what it shows"""
    return max(1, min(lib().uvo_max_threads(), cpu_quota()))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Model:
    """Parsed .param graph + .bin weights (reference: upscale_processing.py:65-71)."""

    def __init__(self, param_path, bin_path):
        err = ctypes.create_string_buffer(256)
        self._h = lib().uvo_load(param_path.encode(), bin_path.encode(), err, 256)
        if not self._h:
            raise RuntimeError("oracle: " + err.value.decode())
        L = lib()
        self.scale = L.uvo_scale(self._h)
        self.nf = L.uvo_nf(self._h)
        self.num_conv = L.uvo_num_conv(self._h)
        self.num_layers = L.uvo_num_layers(self._h)
        self.bin_size = L.uvo_bin_size(self._h)
        self.bin_consumed = L.uvo_bin_consumed(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().uvo_free(self._h)
            self._h = None

    def conv(self, idx):
        """-> (weights OIHW f32, bias f32, tag)"""
        cin, cout, tag = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint32()
        if lib().uvo_conv_info(self._h, idx, cin, cout, tag):
            raise IndexError(idx)
        n = cin.value * cout.value * 9
        w = np.ctypeslib.as_array(lib().uvo_conv_weights(self._h, idx), (n,)).copy()
        b = np.ctypeslib.as_array(lib().uvo_conv_bias(self._h, idx), (cout.value,)).copy()
        return w.reshape(cout.value, cin.value, 3, 3), b, tag.value

    def prelu(self, idx):
        n = ctypes.c_int()
        p = lib().uvo_prelu_slopes(self._h, idx, n)
        if not p:
            raise IndexError(idx)
        return np.ctypeslib.as_array(p, (n.value,)).copy()

    def forward(self, x_chw, flags=0, threads=0):
        """ex.input/ex.extract: f32 [3,h,w] -> f32 [3,h*s,w*s]"""
        x = np.ascontiguousarray(x_chw, dtype=np.float32)
        _, h, w = x.shape
        out = np.empty((3, h * self.scale, w * self.scale), np.float32)
        rc = lib().uvo_forward_f32(self._h, _ptr(x), h, w, _ptr(out), flags, threads or max_threads())
        if rc:
            raise RuntimeError("oracle forward failed")
        return out

    def tap(self, x_chw, conv_idx, flags=0, threads=0):
        """activation after convolution #conv_idx (and its PReLU): f32 [cout,h,w]"""
        x = np.ascontiguousarray(x_chw, dtype=np.float32)
        _, h, w = x.shape
        wgt, _, _ = self.conv(conv_idx)
        out = np.empty((wgt.shape[0], h, w), np.float32)
        rc = lib().uvo_forward_tap(self._h, _ptr(x), h, w, conv_idx, _ptr(out), flags,
                                   threads or max_threads())
        if rc:
            raise RuntimeError("oracle tap failed")
        return out

    def apply_model(self, img_bgr, flags=0, threads=0):
        """apply_model (upscale_processing.py:258-299) minus file I/O: u8 HWC -> u8 HWC"""
        img = np.ascontiguousarray(img_bgr, dtype=np.uint8)
        h, w, _ = img.shape
        out = np.empty((h * self.scale, w * self.scale, 3), np.uint8)
        if lib().uvo_apply_model_u8(self._h, _ptr(img), h, w, _ptr(out), flags, threads or max_threads()):
            raise RuntimeError("oracle apply_model failed")
        return out

    def upscale_image(self, img_bgr, tile_size=960, border=10, flags=0, threads=0):
        """upscale_image/process_tile (upscale_processing.py:395-542) minus file I/O"""
        img = np.ascontiguousarray(img_bgr, dtype=np.uint8)
        h, w, _ = img.shape
        out = np.empty((h * self.scale, w * self.scale, 3), np.uint8)
        if lib().uvo_upscale_image_u8(self._h, _ptr(img), h, w, tile_size, border, _ptr(out), flags,
                                      threads or max_threads()):
            raise RuntimeError("oracle upscale_image failed")
        return out


def from_pixels_normalize(img_bgr):
    img = np.ascontiguousarray(img_bgr, dtype=np.uint8)
    h, w, _ = img.shape
    out = np.empty((3, h, w), np.float32)
    lib().uvo_from_pixels_normalize(_ptr(img), h, w, w * 3, _ptr(out))
    return out


def to_u8(chw):
    x = np.ascontiguousarray(chw, dtype=np.float32)
    _, h, w = x.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().uvo_to_u8(_ptr(x), h, w, _ptr(out), w * 3)
    return out


def round_f16(x):
    return lib().uvo_round_f16(float(x))


from upscale_video_amd.synth import synthetic_frame  # noqa: E402,F401  (the shared input generator)


MODELS_DIR = os.path.join(os.path.dirname(_HERE), "models")
MODEL_FILES = {
    "2x": "2x_Compact_Pretrain",
    "4x": "4x_Compact_Pretrain",
    "1x": "1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g",
}


def load_model(key):
    base = os.path.join(MODELS_DIR, MODEL_FILES[key])
    return Model(base + ".param", base + ".bin")
