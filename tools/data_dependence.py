import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from upscale_video_amd import ncnn
net = ncnn.Net(); net.set_vulkan_device(0)
b = "models/2x_Compact_Pretrain"
assert net.load_param(b + ".param") == 0 and net.load_model(b + ".bin") == 0
h, w = 1080, 1920
from upscale_video_amd.synth import synthetic_frame
rng = np.random.default_rng(1)
kinds = {"zeros": np.zeros((h, w, 3), np.uint8), "smooth": synthetic_frame(h, w),
         "noise": rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "white255": np.full((h, w, 3), 255, np.uint8)}
out = torch.empty((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
for rep in range(2):
    for k, f in kinds.items():
        d = torch.from_numpy(f).cuda()
        for i in range(20): net.process_u8_device(d.data_ptr(), h, w, out.data_ptr(), tile_size=960, border=10)
        net.synchronize(); net.set_profiling(True)
        t0 = time.perf_counter()
        n = 400
        for i in range(n): net.process_u8_device(d.data_ptr(), h, w, out.data_ptr(), tile_size=960, border=10)
        net.synchronize(); dt = time.perf_counter() - t0
        nl, ms = net.kernel_stats(1); net.set_profiling(False)
        print(f"{k:9s} {n/dt:7.1f} fps   trunk {ms/nl*1e3:6.1f} us/launch")
