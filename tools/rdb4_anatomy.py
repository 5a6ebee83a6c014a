#!/usr/bin/env python3
"""Cycle anatomy of rdb4_kernel (a dense block's first four convolutions in one launch) from in-kernel s_memtime stamps.
UVA_RDB_STAMPS=1 UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr.so python tools/rdb4_anatomy.py"""
import ctypes, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UVA_RDB_STAMPS", "1")
from upscale_video_amd.synth import synthetic_weights  # noqa: E402
from upscale_video_amd import _lib, ncnn  # noqa: E402
param = os.path.join(ROOT, "models", "4x_Valar_v1.param")
with tempfile.TemporaryDirectory() as d:
    b = os.path.join(d, "v.bin")
    synthetic_weights(param, b, seed=1, gain=0.5)
    net = ncnn.Net()
    net.set_vulkan_device(0)
    assert net.load_param(param) == 0 and net.load_model(b) == 0
    img = np.random.default_rng(0).integers(0, 256, (970, 970, 3), dtype=np.uint8)
    net.process_u8(img, tile_size=0)
    buf = np.zeros(1024 * 16, np.uint64)
    _lib.check(_lib.load().uva_net_debug_rdb_stamps(net._h, buf.ctypes.data, 1024))
s = buf.reshape(1024, 4, 4).astype(np.int64)
s = s[(s[:, :, 0] > 0).all(axis=1)]
print("rdb4_kernel, 970x970 plane, workgroup 0: %d steps stamped" % len(s))
mid = s[12:-12]
period = np.diff(mid[:, 0, 0])
print("  step period (ticks): median %.0f min %d max %d" % (np.median(period), period.min(), period.max()))
print("  wave: start skew | first part | second part + epilogue | barrier wait | all of the step   [medians, ticks]")
for w in range(4):
    st, p1, br, ps = mid[:, w, 0], mid[:, w, 1], mid[:, w, 2], mid[:, w, 3]
    print("  %d: %6.0f | %6.0f | %6.0f | %6.0f | %6.0f" % (w, np.median(st - mid[:, :, 0].min(axis=1)), np.median(p1 - st), np.median(br - p1),
                                                        np.median(ps - br), np.median(ps - st)))
print("  (wave 0: conv1 18 k-steps | conv2 16; wave 1: conv2 11 + 1x1 | conv3 21; wave 2: conv3 15 | conv4 19; wave 3: conv4 26 | epilogue;"
      " one k-step = 6 MFMAs)")
