"""trunk2_kernel's narrow-strip k-loop against the full one: identical results for every small width (run on the GPU box).
   python tools/narrow_check.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from upscale_video_amd import ncnn
net = ncnn.Net(); net.set_vulkan_device(0)
base = os.path.join(sys.argv[1], "models", "2x_Compact_Pretrain")
assert net.load_param(base + ".param") == 0 and net.load_model(base + ".bin") == 0
rng = np.random.default_rng(5)
out = {}
for h in (5, 23):
    for w in list(range(1, 48)) + [970, 974, 975]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out["%dx%d" % (h, w)] = net.process_u8(img, tile_size=0).copy()
img = rng.integers(0, 256, (70, 2000, 3), dtype=np.uint8)
out["tiled"] = net.process_u8(img, tile_size=960, border=10).copy()
np.savez(sys.argv[2], **out)
'''
import numpy as np
with tempfile.TemporaryDirectory() as d:
    files = []
    for v in ("1", "0"):
        f = os.path.join(d, "o%s.npz" % v)
        subprocess.check_call([sys.executable, "-c", CHILD, os.path.abspath(ROOT), f], env=dict(os.environ, UVA_T2_NARROW=v))
        files.append(np.load(f))
    bad = [k for k in files[0].files if not np.array_equal(files[0][k], files[1][k])]
    print("%d geometries, differing: %s" % (len(files[0].files), bad or "none"))
    sys.exit(1 if bad else 0)
