"""Drop-in for the slice of `from ncnn_vulkan import ncnn` that davlee1972/upscale_video uses,
backed by the MI355X HIP engine (libuva.so, include/uva.h).

The reference touches exactly these names (upscale/upscale_processing.py:65-73, :265-281,
:292, :437-453, :458; test_gpus.py:47-67):

    ncnn.Net(); net.opt.use_vulkan_compute; net.set_vulkan_device(i)
    net.load_param(path); net.load_model(path)
    ncnn.Mat.from_pixels(arr, ncnn.Mat.PixelType.PIXEL_BGR, w, h)
    mat.substract_mean_normalize(mean_vals, norm_vals)
    ex = net.create_extractor(); ex.input(name, mat); ret, mat_out = ex.extract(name)
    np.array(mat_out)
    ncnn.destroy_gpu_instance(); ncnn.get_gpu_count(); ncnn.get_default_gpu_index()
    ncnn.get_gpu_info(i).type() / .device_name()

Same names, argument meaning and error behaviour (load_*/extract return 0 on success).  Mat is a
host-side f32 planar container like ncnn::Mat; all network arithmetic runs in HIP kernels.  There
is no CPU path: without an MI355X every extract raises.
"""
import ctypes

import numpy as np

from . import _lib


def get_gpu_count():
    """test_gpus.py:47"""
    return _lib.load().uva_get_gpu_count()


def get_default_gpu_index():
    """test_gpus.py:53"""
    return _lib.load().uva_get_default_gpu_index()


class GpuInfo:
    def __init__(self, type_, name):
        self._type, self._name = type_, name

    def type(self):
        return self._type

    def device_name(self):
        return self._name


def get_gpu_info(i):
    """test_gpus.py:59-66"""
    t = ctypes.c_int()
    name = ctypes.create_string_buffer(256)
    _lib.check(_lib.load().uva_get_gpu_info(i, t, name, 256))
    return GpuInfo(t.value, name.value.decode())


def destroy_gpu_instance():
    """upscale_processing.py:292, :458"""
    _lib.load().uva_destroy_gpu_instance()


def pinned_empty(shape, dtype=np.uint8):
    """numpy array in page-locked host memory (include/uva.h uva_host_alloc): frames held in such
    arrays are copied to / from the GPU without a staging copy by Net.submit_u8 / process_u8."""
    L = _lib.load()
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dt.itemsize
    ptr = L.uva_host_alloc(max(1, nbytes))
    if not ptr:
        raise _lib.UvaError(L.uva_last_error().decode(errors="replace"))
    buf = (ctypes.c_ubyte * max(1, nbytes)).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    import weakref
    weakref.finalize(buf, L.uva_host_free, ptr)     # freed when the last view of the buffer dies
    return arr


class PngWorkspace:
    """Page-locked memory the GPU's PNG encoder writes an h x w frame's deflate blocks into (include/uva.h
    uva_png_workspace_bytes); file_bytes() frames them as a PNG file on the host (zlib / PNG headers, Adler-32, chunk
    CRC: include/uva.h uva_png_assemble -- no GPU call, the GIL is released)."""

    def __init__(self, h, w):
        L = _lib.load()
        n = L.uva_png_workspace_bytes(h, w)
        if n == 0:
            raise ValueError("the GPU PNG encoder does not take %dx%d frames" % (w, h))
        self.h, self.w = h, w
        self.buf = pinned_empty((n,), np.uint8)
        self._out = None

    def file_bytes(self):
        """-> memoryview of the PNG file image (valid until the next file_bytes() of this workspace)."""
        L = _lib.load()
        n = ctypes.c_size_t(0)
        L.uva_png_assemble(self.buf.ctypes.data, self.h, self.w, None, 0, ctypes.byref(n))      # size query (fails by design)
        if n.value == 0:
            raise _lib.UvaError(L.uva_last_error().decode(errors="replace"))
        if self._out is None or len(self._out) < n.value:
            self._out = bytearray(n.value + n.value // 8)          # the next frame of the stream is about the same size
        dst = (ctypes.c_ubyte * len(self._out)).from_buffer(self._out)
        _lib.check(L.uva_png_assemble(self.buf.ctypes.data, self.h, self.w, dst, len(self._out), ctypes.byref(n)))
        del dst
        return memoryview(self._out)[:n.value]


def png_encode_u8(img_bgr, gpu=0, workspace=None):
    """cv2.imwrite's encoder for a frame in host memory, run on the GPU (include/uva.h uva_png_deflate_u8): -> bytes
    of the PNG file."""
    img = np.ascontiguousarray(img_bgr, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("frame must be u8 [h][w][3]")
    h, w, _ = img.shape
    ws = workspace or PngWorkspace(h, w)
    _lib.check(_lib.load().uva_png_deflate_u8(int(gpu), img.ctypes.data, h, w, w * 3, ws.buf.ctypes.data, ws.buf.nbytes))
    return bytes(ws.file_bytes())


class Ticket:
    """One frame in flight on the pipelined host route; keeps its buffers alive."""

    def __init__(self, tid, img, out):
        self.id, self._img, self.out = tid, img, out


class Mat:
    """Host f32 planar [c][h][w] image, the subset of ncnn::Mat the reference uses."""

    class PixelType:
        PIXEL_RGB = 1
        PIXEL_BGR = 2

    def __init__(self, array):
        self._a = np.ascontiguousarray(array, dtype=np.float32)
        self.c, self.h, self.w = self._a.shape

    @staticmethod
    def from_pixels(array, pixel_type, w, h):
        """ncnn mat_pixel.cpp from_pixels: u8 HWC -> f32 planar, byte k of a pixel -> plane k
        (PIXEL_BGR keeps BGR order: no swap).  upscale_processing.py:265-270, :437-442"""
        if pixel_type not in (Mat.PixelType.PIXEL_BGR, Mat.PixelType.PIXEL_RGB):
            raise ValueError("unsupported pixel type")
        a = np.asarray(array)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[0] != h or a.shape[1] != w or a.shape[2] != 3:
            raise ValueError("from_pixels expects a u8 [h][w][3] array")
        return Mat(a.transpose(2, 0, 1).astype(np.float32))

    def substract_mean_normalize(self, mean_vals, norm_vals):
        """ncnn mat.cpp: per channel (x - mean) * norm in fp32.  upscale_processing.py:271-273"""
        for c in range(self.c):
            if mean_vals:
                self._a[c] -= np.float32(mean_vals[c])
            if norm_vals:
                self._a[c] *= np.float32(norm_vals[c])

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def numpy(self):
        return self._a


class _Option:
    def __init__(self):
        self.use_vulkan_compute = False  # accepted for source compatibility; HIP is always used


def _param_blob_names(path):
    """(input blob, output blob) of an ncnn text .param: the Input layer's top and the last layer's top (what the
    reference passes as model_input / model_output, upscale_processing.py:72-73, always "input" / "output" there)."""
    try:
        with open(path) as f:
            lines = [ln.split() for ln in f.read().splitlines()[2:] if len(ln.split()) >= 4]
    except OSError:
        return None
    first_in = next((p for p in lines if p[0] == "Input"), None)
    if first_in is None or not lines:
        return None
    last = lines[-1]
    nin, nout = int(last[2]), int(last[3])
    if nout < 1 or int(first_in[3]) < 1:
        return None
    return first_in[4 + int(first_in[2])], last[4 + nin + nout - 1]


class Extractor:
    def __init__(self, net):
        self._net = net
        self._in = None

    def input(self, name, mat):
        """ex.input(model_input, mat): the name must be the loaded graph's input blob (:278, :450)"""
        if name != self._net.blob_names[0]:
            return -1
        self._in = mat
        return 0

    def extract(self, name):
        """-> (ret, Mat).  upscale_processing.py:280, :452; the name must be the loaded graph's output blob"""
        if name != self._net.blob_names[1] or self._in is None:
            return -1, None
        out = self._net._extract(np.asarray(self._in))
        return 0, Mat(out)


class Net:
    """ncnn.Net stand-in bound to one HIP device (upscale_processing.py:65-71)."""

    def __init__(self):
        self._L = _lib.load()
        self._h = self._L.uva_net_create()
        self.opt = _Option()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.uva_net_destroy(h)

    @property
    def device_index(self):
        """the HIP ordinal the net is bound to, as the library holds it (include/uva.h uva_net_device)"""
        return int(self._L.uva_net_device(self._h))

    def set_vulkan_device(self, device_index):
        _lib.check(self._L.uva_net_set_device(self._h, int(device_index)))

    blob_names = ("input", "output")     # of the loaded graph (load_param)

    def load_param(self, path):
        rc = self._L.uva_net_load_param(self._h, str(path).encode())
        if rc:
            self.last_error = self._L.uva_last_error().decode()
        else:
            self.blob_names = _param_blob_names(str(path)) or ("input", "output")
        return rc

    def load_model(self, path):
        rc = self._L.uva_net_load_model(self._h, str(path).encode())
        if rc:
            self.last_error = self._L.uva_last_error().decode()
        return rc

    def create_extractor(self):
        return Extractor(self)

    # --- facts -------------------------------------------------------------------------------
    @property
    def scale(self):
        return self._L.uva_net_scale(self._h)

    @property
    def num_features(self):
        return self._L.uva_net_num_features(self._h)

    @property
    def num_convs(self):
        return self._L.uva_net_num_convs(self._h)

    # --- compute -----------------------------------------------------------------------------
    def _extract(self, x_chw):
        x = np.ascontiguousarray(x_chw, dtype=np.float32)
        if x.ndim != 3 or x.shape[0] != 3:
            raise ValueError("input Mat must be [3][h][w]")
        s = self.scale
        if s <= 0:
            raise _lib.UvaError("net has no graph: load_param/load_model failed or were not called")
        _, h, w = x.shape
        out = np.empty((3, h * s, w * s), np.float32)
        _lib.check(self._L.uva_net_extract_f32(self._h, x.ctypes.data, h, w, out.ctypes.data))
        return out

    def process_u8(self, img_bgr, tile_size=0, border=0):
        """Fused device path for a whole frame: u8 HWC BGR -> u8 HWC BGR (include/uva.h
        uva_net_process_u8).  tile_size<=0: apply_model semantics; 960/10: upscale_image's."""
        img = np.ascontiguousarray(img_bgr, dtype=np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("frame must be u8 [h][w][3]")
        s = self.scale
        if s <= 0:
            raise _lib.UvaError("net has no graph: load_param/load_model failed or were not called")
        h, w, _ = img.shape
        out = np.empty((h * s, w * s, 3), np.uint8)
        _lib.check(self._L.uva_net_process_u8(self._h, img.ctypes.data, h, w, w * 3, out.ctypes.data,
                                              w * s * 3, int(tile_size), int(border)))
        return out

    def submit_u8(self, img_bgr, out=None, tile_size=0, border=0):
        """Pipelined process_u8 (include/uva.h uva_net_submit_u8): returns a Ticket at once; up to 3
        frames may be in flight and their H2D copy, kernels and D2H copy overlap.  `out`: optional
        preallocated u8 [h*s][w*s][3] array (pinned_empty avoids the staging copy)."""
        img = np.ascontiguousarray(img_bgr, dtype=np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("frame must be u8 [h][w][3]")
        s = self.scale
        if s <= 0:
            raise _lib.UvaError("net has no graph: load_param/load_model failed or were not called")
        h, w, _ = img.shape
        if out is None:
            out = np.empty((h * s, w * s, 3), np.uint8)
        if out.dtype != np.uint8 or out.shape != (h * s, w * s, 3) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous u8 [h*s][w*s][3] array")
        t = self._L.uva_net_submit_u8(self._h, img.ctypes.data, h, w, w * 3, out.ctypes.data, w * s * 3,
                                      int(tile_size), int(border))
        if t < 0:
            raise _lib.UvaError(self._L.uva_last_error().decode(errors="replace"))
        return Ticket(t, img, out)

    def collect_u8(self, ticket):
        """Waits for the frame of `ticket` and returns its u8 result array (submit_u8) or its PNG workspace
        (submit_u8_png)."""
        _lib.check(self._L.uva_net_collect_u8(self._h, ticket.id))
        return ticket.out

    def submit_u8_png(self, img_bgr, workspace=None, tile_size=0, border=0):
        """Pipelined like submit_u8, but the result frame stays on the GPU and is deflated there (include/uva.h
        uva_net_submit_u8_png): collect_u8(ticket) returns a PngWorkspace whose .file_bytes() is the PNG file
        cv2.imwrite would have been asked for (upscale_processing.py:288, :519).  `workspace`: a PngWorkspace of the
        result's size to reuse (they are page-locked; allocate a few and cycle them)."""
        img = np.ascontiguousarray(img_bgr, dtype=np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("frame must be u8 [h][w][3]")
        s = self.scale
        if s <= 0:
            raise _lib.UvaError("net has no graph: load_param/load_model failed or were not called")
        h, w, _ = img.shape
        if workspace is None:
            workspace = PngWorkspace(h * s, w * s)
        if (workspace.h, workspace.w) != (h * s, w * s):
            raise ValueError("PNG workspace is for %dx%d frames, the result is %dx%d" % (workspace.w, workspace.h, w * s, h * s))
        t = self._L.uva_net_submit_u8_png(self._h, img.ctypes.data, h, w, w * 3, workspace.buf.ctypes.data, workspace.buf.nbytes,
                                          int(tile_size), int(border))
        if t < 0:
            raise _lib.UvaError(self._L.uva_last_error().decode(errors="replace"))
        return Ticket(t, img, workspace)

    def process_u8_device(self, d_in, h, w, d_out, tile_size=0, border=0, in_stride=None, out_stride=None):
        """Asynchronous: raw device pointers (ints) of dense u8 HWC frames in this GPU's HBM."""
        s = self.scale
        _lib.check(self._L.uva_net_process_u8_device(
            self._h, ctypes.c_void_p(d_in), h, w, in_stride or w * 3, ctypes.c_void_p(d_out),
            out_stride or w * s * 3, int(tile_size), int(border)))

    def process_u8_device_batch(self, d_ins, h, w, d_outs, tile_size=0, border=0, in_stride=None, out_stride=None):
        """Asynchronous: several dense u8 HWC frames of ONE geometry in this GPU's HBM (lists of raw device pointers) in one
        call (include/uva.h uva_net_process_u8_device_batch).  The 1x net runs up to eight of them per kernel launch; the bytes
        are those of one process_u8_device call per frame."""
        n = len(d_ins)
        assert n == len(d_outs)
        s = self.scale
        ins = (ctypes.c_void_p * n)(*[int(p) for p in d_ins])
        outs = (ctypes.c_void_p * n)(*[int(p) for p in d_outs])
        _lib.check(self._L.uva_net_process_u8_device_batch(
            self._h, ins, outs, n, h, w, in_stride or w * 3, out_stride or w * s * 3, int(tile_size), int(border)))

    def denoise_u8_device(self, d_in, h, w, d_out, strength, after=None, in_stride=None, out_stride=None):
        """`-m n=K` on a frame in HBM, queued IN FRONT of this net (include/uva.h uva_denoise_u8_device): everything `after`
        (a net, or None) has been asked so far comes first, and whatever this net is asked from now on waits for the frame."""
        _lib.check(self._L.uva_denoise_u8_device(self.device_index, ctypes.c_void_p(d_in), h, w, in_stride or w * 3, ctypes.c_void_p(d_out),
                                                 out_stride or w * 3, float(strength), float(strength),
                                                 after._h if after is not None else None, self._h))

    def wait_for(self, producer):
        """Device-side ordering: work queued on `producer` so far finishes before this net's next work."""
        _lib.check(self._L.uva_net_wait_for(self._h, producer._h))

    def synchronize(self):
        _lib.check(self._L.uva_net_synchronize(self._h))

    def set_profiling(self, on):
        _lib.check(self._L.uva_net_set_profiling(self._h, 1 if on else 0))

    def kernel_stats(self, kind):
        n, ms = ctypes.c_longlong(), ctypes.c_double()
        _lib.check(self._L.uva_net_kernel_stats(self._h, kind, n, ms))
        return n.value, ms.value

    def debug_read_activation(self, conv_idx, h, w):
        out = np.empty((self.num_features, h, w), np.float32)
        _lib.check(self._L.uva_net_debug_read_activation(self._h, conv_idx, out.ctypes.data, h, w))
        return out

    def debug_packed_weights(self, conv_idx):
        need = ctypes.c_size_t()
        _lib.check(self._L.uva_net_debug_packed_weights(self._h, conv_idx, None, 0, need))
        buf = np.empty(need.value, np.uint16)
        _lib.check(self._L.uva_net_debug_packed_weights(self._h, conv_idx, buf.ctypes.data, need.value, need))
        return buf
