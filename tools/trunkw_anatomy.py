#!/usr/bin/env python3
"""Cycle anatomy of trunkw_kernel (two fused trunk layers as Winograd F(2,3)) from in-kernel s_memtime stamps of workgroup 0.
Needs the instrumented build:  python -m upscale_video_amd.build --instrument
                               UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr.so python tools/trunkw_anatomy.py"""
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
img = np.random.default_rng(0).integers(0, 256, (h, w, 3), dtype=np.uint8)
net.process_u8(img, tile_size=960, border=10)
cap = 1024
buf = np.zeros(cap * 8, np.uint64)
n, ms = ctypes.c_int(), ctypes.c_float()
_lib.check(_lib.load().uva_net_debug_trunk_stamps(net._h, buf.ctypes.data, cap, n, 8, ms))
niter = n.value
s = buf[:16 * niter].reshape(niter, 2, 8).astype(np.int64)
entry = int(buf[16 * niter])
A, B = s[:, 0], s[:, 1]
print(f"trunkw_kernel: {ms.value * 1e3:.1f} us per launch; workgroup 0: {niter - 2} steps, {niter} iterations")
print(f"  entry -> first iteration (A): {A[0, 0] - entry} ticks;  whole loop: {A[-1, 3] - A[0, 0]} ticks "
      f"= {(A[-1, 3] - A[0, 0]) / niter:.0f} per iteration; ticks/us: {(A[-1, 3] - entry) / (ms.value * 1e3):.0f}")


def show(label, x):
    print(f"    {label:46s} median {np.median(x):8.1f}  min {x.min():6d}  max {x.max():6d}")


v = A[1:niter - 2]
print("  group A (producer, layer i)")
show("phase X: DMA issue + k-loop", v[:, 1] - v[:, 0])
show("wait at barrier 1 (incl. vmcnt)", v[:, 2] - v[:, 1])
show("phase Y: epilogue -> B-ring", v[:, 3] - v[:, 2])
show("wait at barrier 2", A[2:niter - 1, 0] - v[:, 3])
show("iteration period", A[2:niter - 1, 0] - v[:, 0])
v = B[3:niter - 3]
print("  group B (consumer, layer i+1)")
show("phase X: epilogue (HBM stores)", v[:, 1] - v[:, 0])
show("wait at barrier 1", v[:, 2] - v[:, 1])
# (stamp 4 sits between the raw-row transform and the k-loop: rounds 4's labels had the two swapped)
show("phase Y: raw rows -> A-ring (in front of the k-loop)", v[:, 4] - v[:, 2])
show("phase Y: k-loop (TW_RAW_INK: raw rows inside)", v[:, 3] - v[:, 4])
show("wait at barrier 2", B[4:niter - 2, 0] - v[:, 3])
