"""`-m n=K` (cv2.fastNlMeansDenoisingColored(img, None, K, K, 5, 9)) on the MI355X: ms per 1080p frame, host to host
(uva_denoise_u8: H2D, Lab conversion, two NLM passes, Lab -> BGR, D2H, synchronous).  usage: python tools/denoise_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upscale_video_amd import upscale_processing as up  # noqa: E402
from upscale_video_amd.synth import synthetic_frame  # noqa: E402

img = synthetic_frame(1080, 1920, seed=2)
up.denoise_u8(img, 10, device=0)
t0 = time.perf_counter()
n = 20
for _ in range(n):
    out = up.denoise_u8(img, 10, device=0)
dt = (time.perf_counter() - t0) / n
print("denoise 1920x1080, K=10: %.2f ms per frame = %.0f frames/s (host to host, synchronous)" % (dt * 1e3, 1 / dt))
