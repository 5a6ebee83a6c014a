"""A/B of the generic executor's kernels on 4x_Valar_v1 (synthetic weights): runs the same frames in child
processes under different environment settings and compares the results with the first one.
usage: python tools/sw_ab.py "UVA_GENERIC_SW=0" "UVA_GENERIC_SW=1" ..."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

CHILD = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from upscale_video_amd import ncnn
from upscale_video_amd.synth import synthetic_frame
net = ncnn.Net(); net.set_vulkan_device(0)
assert net.load_param(sys.argv[2]) == 0 and net.load_model(sys.argv[3]) == 0, net.last_error
outs = []
for h, w, t in ((70, 75, 32), (130, 200, 0), (97, 333, 64)):
    img = synthetic_frame(h, w, seed=h + w)
    outs.append(net.process_u8(img, tile_size=t, border=10))
x = (synthetic_frame(66, 140, seed=3).astype(np.float32) / 255.0).transpose(2, 0, 1).copy()
outs.append(net._extract(x))
np.savez(sys.argv[4], *outs)
"""

if __name__ == "__main__":
    from upscale_video_amd.synth import synthetic_weights  # noqa: E402
    param = os.path.join(ROOT, "models", "4x_Valar_v1.param")
    settings = sys.argv[1:] or ["UVA_GENERIC_SW=0", "UVA_GENERIC_SW=1"]
    with tempfile.TemporaryDirectory() as d:
        b = os.path.join(d, "v.bin")
        synthetic_weights(param, b, seed=7, gain=0.5)
        res = []
        for i, s in enumerate(settings):
            env = dict(os.environ)
            for kv in s.split(","):
                if kv:
                    k, v = kv.split("=")
                    env[k] = v
            f = os.path.join(d, "o%d.npz" % i)
            subprocess.check_call([sys.executable, "-c", CHILD, ROOT, param, b, f], env=env)
            res.append(np.load(f))
        for i in range(1, len(res)):
            for k in res[0].files:
                a0, a1 = res[0][k].astype(np.float64), res[i][k].astype(np.float64)
                print("%s vs %s  %s %s: max|diff| %.5g  (max|ref| %.4g, differing %.3f %%)" % (
                    settings[i], settings[0], k, a0.shape, np.abs(a0 - a1).max(), np.abs(a0).max(), 100 * (a0 != a1).mean()))
