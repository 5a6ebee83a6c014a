"""Numerics of a 1-D Winograd F(2,3) trunk for the 2x Compact net -- TEST INFRASTRUCTURE / ANALYSIS ONLY (CPU, torch).

DESIGN.md section 8: the headline kernel (trunk2_kernel<64>) is bound by the package's power, 69 % of which are its MFMAs;
the one lever left is fewer MFMAs per pixel.  F(2,3) along x computes two output columns from four input columns with four
multiplications per (tap row, channel pair) instead of six: -33 % matrix work.  What it costs in accuracy is decided by
where fp16 enters: the MFMA's operands are fp16, so the TRANSFORMED inputs (d0 - d2, d1 + d2, d2 - d1, d1 - d3) and the
transformed weights (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2) are rounded to fp16 where the direct convolution rounds
d and g themselves.  This script runs the real net three ways on the same frames --
    fp32 everywhere                                   (the CPU oracle's arithmetic),
    direct, fp16 activations between the layers       (the shipped kernels' rounding points),
    F(2,3) on the sixteen 64 -> 64 layers, the same rounding points plus the two above
-- and prints PSNR / largest u8 difference of each against fp32, and of F(2,3) against direct.
usage: python oracle/experiments/winograd_f23_numerics.py [h=96] [w=128]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import independent_check as ic  # noqa: E402
from oracle import uvoracle  # noqa: E402


def h16(t):
    return t.to(torch.float16).to(torch.float32)


def conv_direct(x, w, b):
    return F.conv2d(x, h16(w), b, padding=1)


def conv_f23(x, w, b):
    """x [1, C, H, W] (fp16 values in fp32), W even.  Output columns (2j, 2j + 1) from input columns 2j - 1 .. 2j + 2."""
    _, c, hh, ww = x.shape
    xp = F.pad(x, (1, 1, 0, 0))                                   # columns -1 .. W
    d = [xp[..., k:k + ww:2] for k in range(4)]                   # d0..d3 at the W / 2 pair positions
    u = [h16(d[0] - d[2]), h16(d[1] + d[2]), h16(d[2] - d[1]), h16(d[1] - d[3])]
    g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]                  # [Co, Ci, 3 (dy)]
    gg = [h16(g0), h16((g0 + g1 + g2) * 0.5), h16((g0 - g1 + g2) * 0.5), h16(g2)]
    m = [F.conv2d(u[k], gg[k][..., None], None, padding=(1, 0)) for k in range(4)]      # fp32 accumulation
    y = torch.empty((1, w.shape[0], hh, ww))
    y[..., 0::2] = m[0] + m[1] + m[2]
    y[..., 1::2] = m[1] - m[2] - m[3]
    return y + b[None, :, None, None]


def forward(layers, params, x_chw, mode):
    blobs = {}
    nconv = sum(1 for t in layers if t[0] == "Convolution")
    ci = 0
    for typ, name, ins, outs, kv in layers:
        if typ == "Input":
            t = torch.from_numpy(np.ascontiguousarray(x_chw))[None]
            blobs[outs[0]] = t if mode == "fp32" else h16(t)
        elif typ == "Split":
            for o in outs:
                blobs[o] = blobs[ins[0]]
        elif typ == "Convolution":
            w, b, _ = params[name]
            w, b = torch.from_numpy(w), torch.from_numpy(b)
            x = blobs[ins[0]]
            if mode == "fp32":
                y = F.conv2d(x, w, b, padding=1)
            elif mode == "f23" and 0 < ci < nconv - 1:
                y = conv_f23(x, w, b)
            else:
                y = conv_direct(x, w, b)
            blobs[outs[0]] = y
            ci += 1
        elif typ == "PReLU":
            y = F.prelu(blobs[ins[0]], torch.from_numpy(params[name]))
            blobs[outs[0]] = y if mode == "fp32" else h16(y)          # activations are stored as fp16
        elif typ == "PixelShuffle":
            blobs[outs[0]] = F.pixel_shuffle(blobs[ins[0]], int(kv.get(0, 1)))
        elif typ == "Interp":
            s = float(kv.get(1, 1.0))
            t = blobs[ins[0]]
            blobs[outs[0]] = t if s == 1.0 else F.interpolate(t, scale_factor=s, mode="nearest")
        elif typ == "BinaryOp":
            blobs[outs[0]] = blobs[ins[0]] + blobs[ins[1]]
        else:
            raise ValueError(typ)
    out = blobs["output"][0].numpy()
    return np.clip(np.rint(out.transpose(1, 2, 0) * 255), 0, 255).astype(np.uint8)


def cmp(a, b):
    d = a.astype(int) - b.astype(int)
    mse = float((d * d).mean())
    return "PSNR %6.2f dB, max |diff| %d LSB, %5.2f %% of the samples differ" % (99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse),
                                                                                  int(np.abs(d).max()), 100.0 * float((d != 0).mean()))


def main():
    h = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    base = os.path.join(uvoracle.MODELS_DIR, uvoracle.MODEL_FILES["2x"])
    layers = ic.parse_param(base + ".param")
    params, used, size = ic.load_bin(layers, base + ".bin")
    assert used == size
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    for kind in ("smooth", "random"):
        img = uvoracle.synthetic_frame(h, w, kind=kind, seed=11)
        x = img.transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255.0)
        with torch.no_grad():
            ref, direct, f23 = (forward(layers, params, x, m) for m in ("fp32", "direct", "f23"))
        print("%dx%d %-6s  direct fp16 vs fp32: %s" % (w, h, kind, cmp(direct, ref)))
        print("%dx%d %-6s  F(2,3) fp16 vs fp32: %s" % (w, h, kind, cmp(f23, ref)))
        print("%dx%d %-6s  F(2,3) vs direct   : %s" % (w, h, kind, cmp(f23, direct)))


if __name__ == "__main__":
    main()
