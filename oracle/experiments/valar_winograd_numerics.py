"""Numerics of 1-D Winograd F(2,3) on 4x_Valar_v1's 192 -> 64 convolutions (a residual dense block's conv5: 42 % of the graph's
FLOPs) -- TEST INFRASTRUCTURE / ANALYSIS ONLY (CPU, numpy).  VERDICT r5 item 3 (i): "420 convolutions deep, the drift must be
shown, not assumed".  The graph is evaluated three ways on one frame with synthetic weights (the real ones are a missing blob
upstream): fp32; the executor's rounding points (every blob fp16, fp32 accumulation: generic_oracle's f16_storage); the same with
the 69 conv5 layers as F(2,3) along x -- transformed inputs d0-d2, d1+d2, d2-d1, d1-d3 and transformed weights g0, (g0+g1+g2)/2,
(g0-g1+g2)/2, g2 each rounded to fp16 (MFMA operands), fp32 accumulation, output transform in fp32, then the blob's fp16 rounding.
usage: python oracle/experiments/valar_winograd_numerics.py [h=16] [w=24] [gain ...]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import generic_oracle as go  # noqa: E402
from oracle import uvoracle  # noqa: E402

VALAR = os.path.join(ROOT, "models", "4x_Valar_v1.param")


def h16(a):
    return a.astype(np.float16).astype(np.float32)


def conv_f23(x, w, b):
    """x [C][H][W] fp16 values in fp32; w [Co][Ci][3][3]; pairs of output columns (2j, 2j+1) from input columns 2j-1 .. 2j+2"""
    c, h, wd = x.shape
    we = wd + (wd & 1)
    xp = np.zeros((c, h + 2, we + 2), np.float32)
    xp[:, 1:h + 1, 1:wd + 1] = x
    d = [xp[:, :, k:k + we:2] for k in range(4)]                       # [C][H+2][we/2]
    v = [h16(d[0] - d[2]), h16(d[1] + d[2]), h16(d[2] - d[1]), h16(d[1] - d[3])]
    g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]                        # [Co][Ci][3 = dy]
    u = [h16(g0), h16((g0 + g1 + g2) * np.float32(0.5)), h16((g0 - g1 + g2) * np.float32(0.5)), h16(g2)]
    m = []
    for k in range(4):
        rows = np.stack([v[k][:, dy:dy + h, :] for dy in range(3)], axis=1)          # [C][3][H][we/2]
        m.append(np.tensordot(u[k], rows, axes=([1, 2], [0, 1])).astype(np.float32))  # [Co][H][we/2]
    y = np.empty((w.shape[0], h, we), np.float32)
    y[:, :, 0::2] = (m[0] + m[1]) + m[2]
    y[:, :, 1::2] = (m[1] - m[2]) - m[3]
    return (y[:, :, :wd] + b[:, None, None]).astype(np.float32)


class WinoModel(go.Model):
    wino_layers = 0

    def forward(self, x, f16_storage=False):
        base = go.Model._conv

        def conv(xx, w, b, k):
            if k == 3 and w.shape[1] == 192:
                WinoModel.wino_layers += 1
                return conv_f23(xx, w, b)
            return base(xx, w, b, k)
        go.Model._conv = staticmethod(conv)
        try:
            return super().forward(x, f16_storage)
        finally:
            go.Model._conv = staticmethod(base)


def main():
    h = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    gains = [float(a) for a in sys.argv[3:]] or [0.5, 0.7]
    img = uvoracle.synthetic_frame(h, w, seed=5)
    x = img.transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255.0)
    print("4x_Valar_v1, %dx%d frame, synthetic weights (seed 11); distances relative to max|fp32 output|" % (w, h))
    for gain in gains:
        with tempfile.TemporaryDirectory() as td:
            b = os.path.join(td, "v.bin")
            go.write_synthetic_bin(VALAR, b, seed=11, gain=gain)
            t0 = time.time()
            ref = go.Model(VALAR, b).forward(x, f16_storage=False)
            dire = go.Model(VALAR, b).forward(x, f16_storage=True)
            WinoModel.wino_layers = 0
            wino = WinoModel(VALAR, b).forward(x, f16_storage=True)
            scale = float(np.abs(ref).max())
            u8 = lambda a: np.clip(np.rint(a * 255.0), 0, 255).astype(np.int32)       # noqa: E731
            def line(name, a, bb):
                dd = np.abs(a - bb)
                du = np.abs(u8(a) - u8(bb))
                print("  gain %.2f  %-44s max %.3e  mean %.3e   u8: max %d LSB, %.2f %% differ" % (
                    gain, name, dd.max() / scale, dd.mean() / scale, du.max(), 100.0 * (du > 0).mean()))
            print("  gain %.2f  max|out| %.3f, %d conv5 layers as F(2,3), %.0f s" % (gain, scale, WinoModel.wino_layers, time.time() - t0))
            line("direct fp16 storage  vs fp32", dire, ref)
            line("F(2,3) conv5         vs fp32", wino, ref)
            line("F(2,3) conv5         vs direct fp16", wino, dire)


if __name__ == "__main__":
    main()
