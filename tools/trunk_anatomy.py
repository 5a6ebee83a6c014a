#!/usr/bin/env python3
"""Cycle anatomy of the trunk kernel from in-kernel s_memtime stamps (debug aid, GPU only)."""
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
rng = np.random.default_rng(0)
img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
net.process_u8(img, tile_size=960, border=10)
cap = 256
for ablate, label in ((0, "real kernel"), (1, "memory only (no MFMA / LDS reads)"), (2, "compute only (L2-resident in, sink out)"),
                      (3, "TAIL kernel conv3x3_kernel<64,1,2> (stamps: 0 start, 1 k-loop done, 2 barrier, 3 epilogue done)")):
    buf = np.zeros(cap * 8, np.uint64)
    n = ctypes.c_int()
    ms = ctypes.c_float()
    _lib.check(_lib.load().uva_net_debug_trunk_stamps(net._h, buf.ctypes.data, cap, n, ablate, ms))
    s = buf[: n.value * 8].reshape(-1, 8).astype(np.int64)
    kloop, wait, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    gap = s[1:, 0] - s[:-1, 3]
    per = s[1:, 0] - s[:-1, 0]
    print(f"== {label}: {ms.value * 1e3:.1f} us per launch, {n.value} tiles per group (s_memtime ticks)")
    if ablate < 3:
        print(f"  prologue: entry -> schedule built {s[0, 7] - s[0, 5]}, -> first DMA issued {s[1, 7] - s[0, 7]}, -> landed + weights + barrier {s[0, 0] - s[1, 7]}")
    print(f"  workgroup 0: entry -> first k-loop {s[0, 0] - s[0, 5]} ticks, loop {s[-1, 3] - s[0, 0]}, drain {s[0, 6] - s[-1, 3]}, "
          f"total {s[0, 6] - s[0, 5]} ticks = {(s[0, 6] - s[0, 5]) / (ms.value * 1e3):.0f} ticks/us of the launch time")
    for name, v in (("k-loop", kloop), ("dma wait+barrier", wait), ("epilogue", epi), (" math+stage", s[:, 4] - s[:, 2]),
                    (" readback+st", s[:, 3] - s[:, 4]), ("end barrier+ovh", gap), ("tile period", per)):
        print(f"  {name:14s} median {np.median(v):8.1f}  min {v.min():6d}  max {v.max():6d}")
