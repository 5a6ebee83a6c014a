"""Seeded synthetic frames: the input generator shared by bench.py, the tools
and the tests.  No model arithmetic here."""
import numpy as np


def synthetic_frame(h, w, seed=20260928, kind="smooth"):
    """u8 BGR HWC frame.  'smooth': 127 + 100 sin(x/17 + c) cos(y/23) + N(0, 4) per channel c (smooth
    + grain, a photo-like spectrum); 'random': uniform white noise (worst case for fp16 rounding)."""
    rng = np.random.default_rng(seed)
    if kind == "random":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.empty((h, w, 3), np.float64)
    for c in range(3):
        img[..., c] = 127 + 100 * np.sin(x / 17 + c) * np.cos(y / 23) + rng.normal(0, 4, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
