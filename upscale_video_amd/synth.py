"""Seeded synthetic frames: the input generator shared by bench.py, the tools
and the tests -- and random-init weight files for graphs whose .bin is not available (4x_Valar_v1 is a missing blob
upstream, /root/reference/.MISSING_LARGE_BLOBS:1).  No model arithmetic here."""
import struct

import numpy as np

NCNN_FP16_FLAG = 0x01306B47


def synthetic_weights(param_path, bin_path, seed=1, gain=0.5):
    """A .bin for an ncnn .param graph with random weights (He-scaled by the fan-in times `gain`, so activations stay
    O(1) through hundreds of layers): per Convolution the fp16 flag, co*ci*k*k fp16 weights padded to 4 bytes, co fp32
    biases if bias_term; per PReLU the slopes.  Throughput runs only -- what such a net computes means nothing."""
    rng = np.random.default_rng(seed)
    with open(param_path) as f, open(bin_path, "wb") as out:
        for line in f:
            tok = line.split()
            if len(tok) < 4 or tok[0] not in ("Convolution", "PReLU"):
                continue
            kv = dict(t.split("=", 1) for t in tok[4 + int(tok[2]) + int(tok[3]):] if "=" in t and not t.startswith("-"))
            if tok[0] == "PReLU":
                out.write(rng.uniform(0.05, 0.3, int(kv["0"])).astype(np.float32).tobytes())
                continue
            co, k, wsize = int(kv["0"]), int(kv.get("1", 1)), int(kv["6"])
            ci = wsize // (co * k * k)
            w = rng.standard_normal(wsize).astype(np.float32) * np.float32(gain / np.sqrt(ci * k * k))
            raw = w.astype(np.float16).tobytes()
            out.write(struct.pack("<I", NCNN_FP16_FLAG) + raw + b"\0" * (-len(raw) % 4))
            if int(kv.get("5", 0)):
                out.write((rng.standard_normal(co).astype(np.float32) * np.float32(0.05)).tobytes())


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
