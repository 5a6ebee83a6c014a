"""nlm_oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by the product).

numpy restatement of what the reference's `-m n=K` stage computes,
    cv2.fastNlMeansDenoisingColored(img, None, K, K, 5, 9)      /root/reference/upscale/upscale_processing.py:350-361
from OpenCV's published algorithm (modules/photo/src/denoising.cpp: fastNlMeansDenoisingColored = LBGR2Lab,
fastNlMeansDenoising on L with h and on (a, b) with hColor, Lab2LBGR; fast_nlmeans_denoising_invoker.hpp:
reflect-101 border of 6, squared patch distance over the 5x5 template and the channels, `>> 5`, the integer
table round(F * exp(-d*1.28 / (h*h*cn))) with F = INT_MAX // (81*255) = 103969 and weights < F/1000 dropped, integer
accumulation over the 9x9 search window, rounded unsigned division).

PARITY UNPINNED: opencv-python is neither pinned by the reference nor installable here.  The integer NLM stage
follows OpenCV's source line by line; the Lab conversions use the CIE formulas in fp32, where OpenCV's 8-bit path
uses fixed-point tables that are not restated here (an output may differ from OpenCV's by an LSB).
"""
import numpy as np

T, S = 2, 4
B = T + S
SHIFT = 5
INT_MAX = 2 ** 31 - 1


def weight_table(h, cn):
    fixed = INT_MAX // (81 * 255)
    mult = 32.0 / 25.0
    almost_max = int(255 * 255 * cn / mult + 1)
    ad = np.arange(almost_max, dtype=np.float64)
    w = np.exp(-(ad * mult) / (float(np.float32(h)) * float(np.float32(h)) * cn))
    tab = np.rint(fixed * w).astype(np.int64)
    tab[tab < 0.001 * fixed] = 0
    return tab


def nlm_plane(img, h):
    """img: u8 [H][W] or [H][W][cn] -> same shape"""
    a = img[..., None] if img.ndim == 2 else img
    H, W, cn = a.shape
    tab = weight_table(h, cn)
    ext = np.pad(a.astype(np.int64), ((B, B), (B, B), (0, 0)), mode="reflect")       # BORDER_DEFAULT = reflect-101
    est = np.zeros((H, W, cn), np.int64)
    wsum = np.zeros((H, W), np.int64)
    c0 = ext[S:S + H + 2 * T, S:S + W + 2 * T]                                        # centre patches' support
    for sy in range(-S, S + 1):
        for sx in range(-S, S + 1):
            sh = ext[S + sy:S + sy + H + 2 * T, S + sx:S + sx + W + 2 * T]
            d2 = ((c0 - sh) ** 2).sum(axis=2)
            cs = np.pad(d2, ((1, 0), (1, 0))).cumsum(0).cumsum(1)
            k = 2 * T + 1
            dist = cs[k:, k:] - cs[:-k, k:] - cs[k:, :-k] + cs[:-k, :-k]              # 5x5 box sums, [H][W]
            wgt = tab[np.minimum(dist >> SHIFT, len(tab) - 1)]
            p = ext[B + sy:B + sy + H, B + sx:B + sx + W]
            est += wgt[..., None] * p
            wsum += wgt
    out = (est + (wsum // 2)[..., None]) // wsum[..., None]
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[..., 0] if img.ndim == 2 else out


def _f(t):
    return np.where(t > np.float32(0.008856), np.cbrt(t), np.float32(7.787) * t + np.float32(16.0 / 116.0)).astype(np.float32)


def bgr2lab(img):
    x = img.astype(np.float32) * np.float32(1 / 255.0)
    b, g, r = x[..., 0], x[..., 1], x[..., 2]
    X = (np.float32(0.412453) * r + np.float32(0.357580) * g + np.float32(0.180423) * b) / np.float32(0.950456)
    Y = np.float32(0.212671) * r + np.float32(0.715160) * g + np.float32(0.072169) * b
    Z = (np.float32(0.019334) * r + np.float32(0.119193) * g + np.float32(0.950227) * b) / np.float32(1.088754)
    fx, fy, fz = _f(X), _f(Y), _f(Z)
    L = np.where(Y > np.float32(0.008856), np.float32(116.0) * fy - np.float32(16.0), np.float32(903.3) * Y)
    a = np.float32(500.0) * (fx - fy)
    bb = np.float32(200.0) * (fy - fz)
    lab = np.stack([L * np.float32(2.55), a + np.float32(128.0), bb + np.float32(128.0)], axis=-1)
    return np.clip(np.rint(lab), 0, 255).astype(np.uint8)


def lab2bgr(lab):
    L = lab[..., 0].astype(np.float32) * np.float32(100.0 / 255.0)
    a = lab[..., 1].astype(np.float32) - np.float32(128.0)
    b = lab[..., 2].astype(np.float32) - np.float32(128.0)
    lo = L <= np.float32(8.0)
    Ylo = L / np.float32(903.3)
    fy = np.where(lo, np.float32(7.787) * Ylo + np.float32(16.0 / 116.0), (L + np.float32(16.0)) / np.float32(116.0)).astype(np.float32)
    Y = np.where(lo, Ylo, fy * fy * fy).astype(np.float32)
    fx, fz = fy + a / np.float32(500.0), fy - b / np.float32(200.0)
    ft = np.float32(7.787) * np.float32(0.008856) + np.float32(16.0 / 116.0)
    X = np.where(fx <= ft, (fx - np.float32(16.0 / 116.0)) / np.float32(7.787), fx * fx * fx) * np.float32(0.950456)
    Z = np.where(fz <= ft, (fz - np.float32(16.0 / 116.0)) / np.float32(7.787), fz * fz * fz) * np.float32(1.088754)
    R = np.float32(3.240479) * X - np.float32(1.53715) * Y - np.float32(0.498535) * Z
    G = np.float32(-0.969256) * X + np.float32(1.875991) * Y + np.float32(0.041556) * Z
    Bc = np.float32(0.055648) * X - np.float32(0.204043) * Y + np.float32(1.057311) * Z
    out = np.stack([Bc, G, R], axis=-1) * np.float32(255.0)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def denoise_colored(img, h, h_color):
    lab = bgr2lab(img)
    L = nlm_plane(lab[..., 0], h)
    ab = nlm_plane(np.ascontiguousarray(lab[..., 1:]), h_color)
    return lab2bgr(np.concatenate([L[..., None], ab], axis=-1))
