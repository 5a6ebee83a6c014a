"""Soak test: the same frames through every net thousands of times back to back (sustained,
power-limited state, all kernels' ping-pong / LDS-DMA ordering under real timing); every result is
compared on the device with the first one.  A single differing byte is a failure.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upscale_video_amd import ncnn
from upscale_video_amd.synth import synthetic_frame, synthetic_weights

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
VALAR_ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, ITERS // 20)     # 85 ms per frame: the persistent dense-block kernels
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bad = 0
import tempfile
tmp = tempfile.TemporaryDirectory()
for stem, tile in (("2x_Compact_Pretrain", 960), ("4x_Compact_Pretrain", 960), ("1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g", 0),
                   ("2x_Compact_Pretrain", 0), ("4x_Valar_v1", 960)):
    net = ncnn.Net(); net.set_vulkan_device(0)
    base = os.path.join(ROOT, "models", stem)
    weights = base + ".bin"
    if stem == "4x_Valar_v1" and not os.path.exists(weights):       # a missing blob upstream: random-init weights
        weights = os.path.join(tmp.name, "v.bin")
        synthetic_weights(base + ".param", weights, seed=1, gain=0.5)
        ITERS = VALAR_ITERS
    assert net.load_param(base + ".param") == 0 and net.load_model(weights) == 0
    s = net.scale
    h, w = 1080, 1920
    frames = [torch.from_numpy(synthetic_frame(h, w, seed=7 + i, kind="random" if i & 1 else "smooth")).cuda() for i in range(3)]
    outs = [torch.empty((h * s, w * s, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    refs = []
    for f in frames:
        net.process_u8_device(f.data_ptr(), h, w, outs[0].data_ptr(), tile_size=tile, border=10 if tile else 0)
        net.synchronize()
        refs.append(outs[0].clone())
    t0 = time.perf_counter()
    mism = torch.zeros((), dtype=torch.int64, device="cuda")
    for i in range(ITERS):
        o = outs[i & 1]
        net.process_u8_device(frames[i % 3].data_ptr(), h, w, o.data_ptr(), tile_size=tile, border=10 if tile else 0)
        net.synchronize()
        mism += (o != refs[i % 3]).any().to(torch.int64)
    torch.cuda.synchronize()
    n = int(mism.item())
    bad += n
    print(f"{stem} tile={tile}: {ITERS} frames in {time.perf_counter() - t0:.1f} s, {n} differing results")
sys.exit(1 if bad else 0)
