"""Timing of the 1x net on one device-resident 1080p frame, no result checks: for kernel variants whose output is
knowingly wrong (timing ablations).  UVA_LIB_PATH selects the build."""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from upscale_video_amd import ncnn
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
net = ncnn.Net(); net.set_vulkan_device(0)
base = os.path.join(ROOT, "models", "1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g")
assert net.load_param(base + ".param") == 0 and net.load_model(base + ".bin") == 0
img = torch.randint(0, 256, (1080, 1920, 3), dtype=torch.uint8, device="cuda")
out = torch.empty_like(img)
for _ in range(20): net.process_u8_device(img.data_ptr(), 1080, 1920, out.data_ptr())
net.synchronize()
torch.cuda.synchronize(); t0 = time.time()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for _ in range(N): net.process_u8_device(img.data_ptr(), 1080, 1920, out.data_ptr())
net.synchronize(); dt = time.time() - t0
print("%.4f ms/frame  %.0f fps" % (dt / N * 1e3, N / dt))
