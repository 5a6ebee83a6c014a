#!/usr/bin/env python3
"""What do the profiling events cost the frame?  The K route (frames resident in HBM) with uva_net_set_profiling on and off,
alternately, same process: ms per frame by wall clock around 300 frames each (GPU only)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
from upscale_video_amd import ncnn  # noqa: E402

h, w, n = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 300
net = ncnn.Net()
net.set_vulkan_device(0)
base = os.path.join(ROOT, "models", "2x_Compact_Pretrain")
assert net.load_param(base + ".param") == 0 and net.load_model(base + ".bin") == 0
rng = np.random.default_rng(0)
src = torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).cuda()
dst = torch.empty((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")


def run(k):
    for _ in range(k):
        net.process_u8_device(src.data_ptr(), h, w, dst.data_ptr(), tile_size=960, border=10)
    net.synchronize()


run(60)
for rnd in range(4):
    for mode in (1, 0):
        net.set_profiling(bool(mode))
        run(10)
        t0 = time.perf_counter()
        run(n)
        dt = time.perf_counter() - t0
        net.kernel_stats(1)
        net.set_profiling(False)
        print("profiling %s: %.4f ms per frame, %.1f frames/s" % ({1: "on ", 0: "off"}[mode], dt / n * 1e3, n / dt))
