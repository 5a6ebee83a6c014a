"""Soak of the host routes: random frame sizes and tilings through the pipelined submit/collect route
(pinned and pageable buffers mixed, three frames in flight) for a fixed time; every result is compared
with the synchronous call's.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from upscale_video_amd import ncnn
from upscale_video_amd.synth import synthetic_frame

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nets = {}
for key, stem in (("2x", "2x_Compact_Pretrain"), ("4x", "4x_Compact_Pretrain"), ("1x", "1x_HurrDeblur_SubCompact_nf24-nc8_244k_net_g")):
    n = ncnn.Net(); n.set_vulkan_device(0)
    base = os.path.join(ROOT, "models", stem)
    assert n.load_param(base + ".param") == 0 and n.load_model(base + ".bin") == 0
    nets[key] = n
rng = np.random.default_rng(5)
t_end = time.time() + SECONDS
frames = bad = 0
while time.time() < t_end:
    key = ("2x", "4x", "1x")[int(rng.integers(0, 3))]
    net = nets[key]
    batch = []
    for _ in range(int(rng.integers(1, 9))):
        h, w = int(rng.integers(1, 400)), int(rng.integers(1, 500))
        ts = int(rng.choice([0, 32, 64, 128, 960])) if key != "1x" else 0
        if ts and -(-h // ts) * -(-w // ts) > 64:      # the engine holds at most 64 planes (tiles) per frame
            ts = 0
        img = synthetic_frame(h, w, seed=int(rng.integers(0, 1 << 30)), kind="random" if rng.integers(0, 2) else "smooth")
        if rng.integers(0, 2):
            pin = ncnn.pinned_empty(img.shape); pin[...] = img; img = pin
        batch.append((img, ts))
    want = [net.process_u8(np.array(img), tile_size=ts, border=10 if ts else 0) for img, ts in batch]
    inflight, got = [], []
    for img, ts in batch:
        if len(inflight) == 3:
            got.append(net.collect_u8(inflight.pop(0)).copy())
        out = ncnn.pinned_empty((img.shape[0] * net.scale, img.shape[1] * net.scale, 3)) if rng.integers(0, 2) else None
        inflight.append(net.submit_u8(img, out=out, tile_size=ts, border=10 if ts else 0))
    while inflight:
        got.append(net.collect_u8(inflight.pop(0)).copy())
    for g, wnt in zip(got, want):
        frames += 1
        bad += not np.array_equal(g, wnt)
print(f"{frames} frames of random geometry through submit/collect in {SECONDS:.0f} s: {bad} differ from the synchronous route")
sys.exit(1 if bad else 0)
