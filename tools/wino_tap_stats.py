"""Per-layer agreement of the HIP path with the oracle in the product's storage mode (run twice: UVA_TRUNK_WINO=1 / 0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import uvoracle as oracle
from upscale_video_amd import ncnn
from conftest import load_net
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (37, 70)
kind = sys.argv[3] if len(sys.argv) > 3 else "random"
net = load_net(ncnn, "2x")
om = oracle.load_model("2x")
img = oracle.synthetic_frame(h, w, kind=kind)
x = oracle.from_pixels_normalize(img)
net._extract(x)
print("UVA_TRUNK_WINO =", os.environ.get("UVA_TRUNK_WINO", "1"), "flags", oracle.product_flags(), h, w, kind)
for idx in range(net.num_convs - 1):
    got = net.debug_read_activation(idx, h, w)
    want = om.tap(x, idx, flags=oracle.product_flags('f32'))
    w32 = om.tap(x, idx, flags=0)
    d = np.abs(got - want)
    i = np.unravel_index(d.argmax(), d.shape)
    print("conv %2d scale %7.2f  max %.4f at %s (want %.3f)  mean %.2e  differ %.3f %%   vs fp32: gpu %.2e oracle16 %.2e" % (
        idx, np.abs(want).max(), d.max(), i, want[i], d.mean(), 100 * (d > 0).mean(), np.abs(got - w32).mean(), np.abs(want - w32).mean()))
# rounding direction of the stored fp16 values against the fp32 oracle: RNE -> about half are larger in magnitude
for idx in (0, 1, 2):
    got = net.debug_read_activation(idx, h, w)
    w32 = om.tap(x, idx, flags=0)
    nz = got != w32
    print("conv", idx, "share of |gpu| > |fp32|: %.3f" % float((np.abs(got[nz]) > np.abs(w32[nz])).mean()),
          " oracle16: %.3f" % float((np.abs(om.tap(x, idx, flags=1)[nz]) > np.abs(w32[nz])).mean()))
