"""BASELINE config 4 as named: 4x_Valar_v1 on 1920x1080 frames, one MI355X -- through the generic layer-by-layer
executor, with SYNTHETIC weights (the real .bin is a missing blob upstream: throughput only, no parity).
usage: python tools/valar_bench.py [frames=3] [height=1080] [width=1920]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from upscale_video_amd.synth import synthetic_weights  # noqa: E402
from upscale_video_amd import ncnn  # noqa: E402
from upscale_video_amd.synth import synthetic_frame  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
param = os.path.join(ROOT, "models", "4x_Valar_v1.param")
with tempfile.TemporaryDirectory() as d:
    b = os.path.join(d, "4x_Valar_v1.bin")
    synthetic_weights(param, b, seed=1, gain=0.5)
    net = ncnn.Net()
    net.set_vulkan_device(0)
    assert net.load_param(param) == 0 and net.load_model(b) == 0, getattr(net, "last_error", "")
    src = torch.from_numpy(synthetic_frame(h, w, seed=1)).cuda()
    out = torch.empty((4 * h, 4 * w, 3), dtype=torch.uint8, device="cuda")
    net.process_u8_device(src.data_ptr(), h, w, out.data_ptr(), tile_size=960, border=10)
    net.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net.process_u8_device(src.data_ptr(), h, w, out.data_ptr(), tile_size=960, border=10)
    net.synchronize()
    dt = (time.perf_counter() - t0) / n
    flop = 36_136_320 * h * w      # Valar 36 136 320 FLOP per input pixel (un-tiled frame)
    print("4x_Valar_v1 (synthetic weights), %dx%d -> %dx%d, reference tiling: %.3f s per frame = %.2f frames/s, %.0f TFLOP/s"
          % (w, h, 4 * w, 4 * h, dt, 1 / dt, flop / dt / 1e12))
    print("output finite and non-constant:", bool(out.float().std().item() > 0))
