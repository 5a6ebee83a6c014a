#!/usr/bin/env python3
"""Cycle anatomy of pair24_kernel (two 24-feature trunk layers per launch) from in-kernel s_memtime stamps.
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr.so python tools/pair24_anatomy.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upscale_video_amd import _lib, ncnn  # noqa: E402
net = ncnn.Net()
net.set_vulkan_device(0)
base = os.path.join(ROOT, "models", "1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g")
assert net.load_param(base + ".param") == 0 and net.load_model(base + ".bin") == 0
img = np.random.default_rng(0).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
net.process_u8(img, tile_size=0)
cap = 256
buf = np.zeros(cap * 8, np.uint64)
n, ms = ctypes.c_int(), ctypes.c_float()
_lib.check(_lib.load().uva_net_debug_trunk_stamps(net._h, buf.ctypes.data, cap, n, 6, ms))
s = buf[:8 * n.value].reshape(-1, 8).astype(np.int64)
s = s[s[:, 5] > 0]
print(f"pair24_kernel: {ms.value * 1e3:.1f} us per launch, workgroup 0: {len(s)} tiles; loop {s[-1, 5] - s[0, 0]} ticks = {(s[-1,5]-s[0,0])/len(s):.0f} per tile")
for name, v in (("wait tile + barrier", s[:, 1] - s[:, 0]), ("stage A (decode, DMA issue, 3 fragments)", s[:, 2] - s[:, 1]), ("barrier (intermediate)", s[:, 3] - s[:, 2]),
                ("stage B k-loop", s[:, 4] - s[:, 3]), ("stage B epilogue + stores", s[:, 5] - s[:, 4]), ("tile period", s[1:, 0] - s[:-1, 0])):
    print(f"  {name:42s} median {np.median(v):8.1f}  min {v.min():6d}  max {v.max():6d}")
