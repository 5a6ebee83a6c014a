"""A stand-in for the loaded libuva.so object (upscale_video_amd._lib._lib) -- TESTS ONLY, build container only.

tests/test_ref_host.py::test_reference_module_runs_on_our_ncnn runs the reference's module on upscale_video_amd.ncnn where
there is no GPU; the C entry points ncnn.Net / Mat / Extractor reach (include/uva.h) are answered here with the CPU oracle.
The product never imports this; with the real library the same calls run HIP kernels (tests/test_gpu_workers.py)."""
import ctypes

import numpy as np

from oracle import uvoracle

FAIL_EXTRACT = False


class _Net:
    def __init__(self):
        self.device, self.param, self.model = -1, None, None


def _arr(addr, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.frombuffer((ctypes.c_ubyte * n).from_address(int(addr)), dtype=dtype).reshape(shape)


class Lib:
    def __init__(self):
        self.nets, self.err = {}, b""

    def _fail(self, msg):
        self.err = msg.encode()
        return -1

    def uva_last_error(self):
        return self.err

    def uva_get_gpu_count(self):
        return 1

    def uva_get_default_gpu_index(self):
        return 0

    def uva_destroy_gpu_instance(self):
        pass

    def uva_net_create(self):
        h = len(self.nets) + 1
        self.nets[h] = _Net()
        return h

    def uva_net_destroy(self, h):
        self.nets.pop(h, None)

    def uva_net_set_device(self, h, i):
        if i != 0:
            return self._fail("no such device")
        self.nets[h].device = i
        return 0

    def uva_net_device(self, h):
        return self.nets[h].device

    def uva_net_load_param(self, h, path):
        try:
            open(path.decode()).close()
        except OSError as e:
            return self._fail(str(e))
        self.nets[h].param = path.decode()
        return 0

    def uva_net_load_model(self, h, path):
        n = self.nets[h]
        if n.param is None:
            return self._fail("load_param first")
        try:
            n.model = uvoracle.Model(n.param, path.decode())
        except Exception as e:  # noqa: BLE001
            return self._fail(str(e))
        return 0

    def uva_net_scale(self, h):
        m = self.nets[h].model
        return m.scale if m else 0

    def uva_net_num_features(self, h):
        return 0

    def uva_net_num_convs(self, h):
        return 0

    def uva_net_extract_f32(self, h, x, hh, w, out):
        if FAIL_EXTRACT:
            return self._fail("stand-in: extract failed")
        m = self.nets[h].model
        s = m.scale
        _arr(out, (3, hh * s, w * s), np.float32)[...] = m.forward(_arr(x, (3, hh, w), np.float32))
        return 0

    def uva_net_process_u8(self, h, src, hh, w, in_stride, dst, out_stride, tile_size, border):
        if FAIL_EXTRACT:
            return self._fail("stand-in: extract failed")
        m = self.nets[h].model
        s = m.scale
        assert in_stride == w * 3 and out_stride == w * s * 3
        img = _arr(src, (hh, w, 3), np.uint8)
        res = m.upscale_image(img, tile_size, border) if tile_size > 0 else m.apply_model(img)
        _arr(dst, (hh * s, w * s, 3), np.uint8)[...] = res
        return 0
