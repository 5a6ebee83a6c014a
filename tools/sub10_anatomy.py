#!/usr/bin/env python3
"""Cycle anatomy of sub10_kernel (the whole 24-feature 1x net in one launch) from in-kernel s_memtime stamps.
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr.so python tools/sub10_anatomy.py"""
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
cap = 4096
buf = np.zeros(cap * 8, np.uint64)
n, ms = ctypes.c_int(), ctypes.c_float()
_lib.check(_lib.load().uva_net_debug_trunk_stamps(net._h, buf.ctypes.data, cap, n, 7, ms))
s = buf[:48 * n.value].reshape(-1, 12, 4).astype(np.int64)
s = s[s[:, 0, 0] > 0]
print(f"sub10_kernel: {ms.value * 1e3:.1f} us per launch, workgroup 0: {len(s)} steps; {(s[-1,:,2].max()-s[0,:,0].min())/len(s):.0f} ticks per step")
mid = s[30:-30]
period = np.diff(mid[:, 0, 0])
print(f"  step period: median {np.median(period):.0f} min {period.min()} max {period.max()}")
print("  wave: start skew | start -> at the barrier (the row's work) | barrier wait (next start - arrival)   [medians over the steady part]")
print("  (waves 0-7: trunk layers 1-8; S10_BAL builds: 8, 9 the two halves of the last layer, 10, 11 of the first; before: the other way round)")
for w in range(12):
    st = mid[:-1, w, 0]; br = mid[:-1, w, 2]; nx = mid[1:, w, 0]
    print(f"  {w:2d}: {np.median(st - mid[:-1, :, 0].min(axis=1)):6.0f} | {np.median(br - st):6.0f} | {np.median(nx - br):6.0f}")
