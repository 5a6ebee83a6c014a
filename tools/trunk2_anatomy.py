#!/usr/bin/env python3
"""Cycle anatomy of trunk2_kernel (two fused trunk layers) from in-kernel s_memtime stamps of workgroup 0.
Needs the instrumented build:  python -m upscale_video_amd.build --instrument
                               UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr.so python tools/trunk2_anatomy.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upscale_video_amd import _lib, ncnn  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
net = ncnn.Net()
net.set_vulkan_device(0)
base = os.path.join(ROOT, "models", "2x_Compact_Pretrain")
assert net.load_param(base + ".param") == 0 and net.load_model(base + ".bin") == 0
img = np.random.default_rng(0).integers(0, 256, (h, w, 3), dtype=np.uint8)
net.process_u8(img, tile_size=960, border=10)
cap = 1024
buf = np.zeros(cap * 8, np.uint64)
n, ms = ctypes.c_int(), ctypes.c_float()
_lib.check(_lib.load().uva_net_debug_trunk_stamps(net._h, buf.ctypes.data, cap, n, 5, ms))
niter = n.value
s = buf[:16 * niter].reshape(niter, 2, 8).astype(np.int64)
entry = int(buf[16 * niter])
print(f"trunk2_kernel: {ms.value * 1e3:.1f} us per launch; workgroup 0: {niter - 2} steps, {niter} iterations")
print(f"  entry -> first k-loop (A): {s[0, 0, 0] - entry} ticks;  whole loop: {s[-1, 0, 3] - s[0, 0, 0]} ticks "
      f"= {(s[-1, 0, 3] - s[0, 0, 0]) / (niter):.0f} per iteration; total/launch time: {(s[-1, 0, 3] - entry) / (ms.value * 1e3):.0f} ticks/us")
for g, name in ((0, "A (producer, layer i)"), (1, "B (consumer, layer i+1)")):
    lo, hi = (0, niter - 2) if g == 0 else (2, niter)
    v = s[lo:hi, g]
    rows = (("k-loop", v[:, 1] - v[:, 0]), ("wait at role swap", v[:, 2] - v[:, 1]), ("epilogue", v[:, 3] - v[:, 2]),
            ("wait at closing barrier", v[1:, 0] - v[:-1, 3]), ("iteration period", v[1:, 0] - v[:-1, 0]))
    print(f"  group {name}")
    for label, x in rows:
        print(f"    {label:24s} median {np.median(x):8.1f}  min {x.min():6d}  max {x.max():6d}")
