"""The 1x net (sub10_kernel) on frames of 1920 x (1080 k) pixels, k = 1, 2, 4: what a launch that carries k frames' rows would run at
(the pipeline's fill and drain and a segment's 20 warm-up rows are per workgroup and launch, not per row) -- a ceiling for
"k frames per launch", no such API exists.  1080p-frame equivalents per second."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from upscale_video_amd import ncnn
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
net = ncnn.Net(); net.set_vulkan_device(0)
base = os.path.join(ROOT, "models", "1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g")
assert net.load_param(base + ".param") == 0 and net.load_model(base + ".bin") == 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for rep in range(2):
    for k in (1, 2, 4):
        h = 1080 * k
        img = torch.randint(0, 256, (h, 1920, 3), dtype=torch.uint8, device="cuda")
        out = torch.empty_like(img)
        for _ in range(20): net.process_u8_device(img.data_ptr(), h, 1920, out.data_ptr())
        net.synchronize(); torch.cuda.synchronize(); t0 = time.time()
        for _ in range(N): net.process_u8_device(img.data_ptr(), h, 1920, out.data_ptr())
        net.synchronize(); dt = time.time() - t0
        print("1920 x %4d: %.4f ms per launch = %.0f 1080p frames/s" % (h, dt / N * 1e3, N * k / dt))
