"""Two nets (two streams) on one GPU, frames resident in HBM, against one: how much do kernels of different frames overlap?
Usage: python tools/two_nets_probe.py [frames]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch
from upscale_video_amd import ncnn
from upscale_video_amd.synth import synthetic_frame

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
h, w = 1080, 1920
base = os.path.join(ROOT, "models", "2x_Compact_Pretrain")


def make():
    n = ncnn.Net()
    n.set_vulkan_device(0)
    assert n.load_param(base + ".param") == 0 and n.load_model(base + ".bin") == 0
    return n


frames = [torch.from_numpy(synthetic_frame(h, w, seed=5 + i)).cuda() for i in range(4)]
nets = [make(), make()]
outs = [torch.empty((2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda") for _ in nets]
torch.cuda.synchronize()


def run(k, total):
    for i in range(total):
        j = i % k
        nets[j].process_u8_device(frames[i % 4].data_ptr(), h, w, outs[j].data_ptr(), tile_size=960, border=10)
    for n in nets[:k]:
        n.synchronize()


for k in (1, 2, 1, 2):
    run(k, 20)
    t0 = time.perf_counter()
    run(k, N)
    dt = time.perf_counter() - t0
    print("%d net(s), alternating frames: %.1f frames/s" % (k, N / dt))
