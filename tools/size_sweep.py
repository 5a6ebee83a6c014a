"""Pixels/s of the 2x Compact net against frame size (whole-frame planes): does an activation
working set that fits the 256 MB Infinity Cache run faster per pixel?  (power-bound kernel: HBM
traffic costs energy, energy costs clock)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from upscale_video_amd import ncnn
from upscale_video_amd.synth import synthetic_frame
net = ncnn.Net(); net.set_vulkan_device(0)
b = "models/2x_Compact_Pretrain"
assert net.load_param(b + ".param") == 0 and net.load_model(b + ".bin") == 0
for rep in range(2):
    for (w, h) in [(1920, 1080), (1280, 720), (1600, 900), (960, 540), (2560, 1440), (1920, 1080)]:
        d = torch.from_numpy(synthetic_frame(h, w)).cuda()
        out = torch.empty((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
        n = int(400 * (1920 * 1080) / (w * h))
        for i in range(20): net.process_u8_device(d.data_ptr(), h, w, out.data_ptr(), tile_size=0)
        net.synchronize(); net.set_profiling(True)
        t0 = time.perf_counter()
        for i in range(n): net.process_u8_device(d.data_ptr(), h, w, out.data_ptr(), tile_size=0)
        net.synchronize(); dt = time.perf_counter() - t0
        nl, ms = net.kernel_stats(1); net.set_profiling(False)
        print(f"{w}x{h}: act {w*h*128/1e6:6.0f} MB/buffer  {n/dt:7.1f} fps  {n*w*h/dt/1e6:7.1f} Mpx/s   trunk {ms/nl*1e3:6.1f} us/launch "
              f"= {2*9*64*64*w*h/(ms/nl*1e-3)/1e12:6.1f} TFLOP/s")
