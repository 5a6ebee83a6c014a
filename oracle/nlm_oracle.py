"""nlm_oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by the product).

numpy restatement of what the reference's `-m n=K` stage computes,
    cv2.fastNlMeansDenoisingColored(img, None, K, K, 5, 9)      /root/reference/upscale/upscale_processing.py:350-361
from OpenCV's published algorithm (modules/photo/src/denoising.cpp: fastNlMeansDenoisingColored = LBGR2Lab,
fastNlMeansDenoising on L with h and on (a, b) with hColor, Lab2LBGR; fast_nlmeans_denoising_invoker.hpp:
reflect-101 border of 6, squared patch distance over the 5x5 template and the channels, `>> 5`, the integer
table round(F * exp(-d*1.28 / (h*h*cn))) with F = INT_MAX // (81*255) = 103969 and weights < F/1000 dropped, integer
accumulation over the 9x9 search window, rounded unsigned division).

PARITY UNPINNED: opencv-python is neither pinned by the reference nor installable here.  The integer NLM stage
follows OpenCV's source line by line; the Lab conversions follow the STRUCTURE of OpenCV's 8-bit fixed-point path
(tables and shifts, restated from memory -- see the section below for what that is worth).  The reference hands
the function a cv2.UMat: where OpenCV has an OpenCL device it takes its OpenCL branch instead
(ocl_fastNlMeansDenoisingColored, and an OpenCL cvtColor whose 8-bit Lab -> BGR goes through floats), whose results
need not equal the CPU branch's to the last bit -- "the reference's output" is itself device-dependent here.
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


# ---- OpenCV's 8-bit Lab conversions, the INTEGER path -----------------------------------------------------------------
# cvtColor(CV_8U, COLOR_LBGR2Lab / COLOR_Lab2LBGR) does not evaluate the CIE formulas: it runs fixed-point table code
# (modules/imgproc/src/color_lab.cpp: RGB2Lab_b, Lab2RGBinteger, initLabTabs).  Restated here FROM MEMORY of the
# OpenCV 4.x source -- this image has neither the wheel nor the source, so no constant below could be checked against
# it: lab_shift 12, gamma_shift 3, lab_shift2 15, LAB_CBRT_TAB_SIZE_B 256*3/2*8, base_shift 14, inv_gamma_shift 12,
# minABvalue -8145, the (5*a*53687 + 128) >> 13 and (b*41943 + 16) >> 9 divisions, the D65 matrices.  OpenCV builds the
# tables in softfloat (IEEE binary32, round to nearest even); numpy's float32 arithmetic is the same arithmetic, except
# for cv::cbrt, a polynomial approximation good to about one unit in the last place: a LabCbrtTab_b entry is
# round(32768 * cbrt(x)), so where OpenCV's value and the correctly rounded one straddle a .5 the entry -- and, for the
# pixels that hit it, rarely a Lab value -- can differ by one unit.  PARITY UNPINNED, in structure and in constants.
LAB_SHIFT, GAMMA_SHIFT = 12, 3
LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT
LAB_CBRT_TAB_SIZE_B = 256 * 3 // 2 * (1 << GAMMA_SHIFT)
BASE_SHIFT, INV_GAMMA_SHIFT = 14, 12
LAB_BASE = 1 << BASE_SHIFT
MIN_AB = -8145
_RGB2XYZ = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
_XYZ2RGB = np.array([[3.240479, -1.53715, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
_D65 = np.array([0.950456, 1.0, 1.088754])


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def lab_cbrt_tab_b():
    f32 = np.float32
    x = (f32(1) / (f32(255) * f32(1 << GAMMA_SHIFT))) * np.arange(LAB_CBRT_TAB_SIZE_B, dtype=np.float32)
    lthresh, lscale, lbias = f32(216) / f32(24389), f32(841) / f32(108), f32(16) / f32(116)
    lin = (x.astype(np.float64) * np.float64(lscale) + np.float64(lbias)).astype(np.float32)      # mulAdd: one rounding
    cb = np.cbrt(x.astype(np.float64)).astype(np.float32)
    return np.rint(f32(1 << LAB_SHIFT2) * np.where(x < lthresh, lin, cb)).astype(np.int64)


def lab_fwd_coeffs():
    """[3][3] ints: row = X/whiteX, Y, Z/whiteZ from (R, G, B), scaled by 1 << lab_shift (softdouble arithmetic)"""
    return np.rint((1 << LAB_SHIFT) * _RGB2XYZ / _D65[:, None]).astype(np.int64)


def lab_inv_coeffs():
    """[3][3] ints: row = R, G, B from (x, y, z) with the white point folded in"""
    return np.rint((1 << LAB_SHIFT) * _XYZ2RGB * _D65[None, :]).astype(np.int64)


def lab_to_yf_b():
    f32 = np.float32
    y, ify = np.zeros(256, np.int64), np.zeros(256, np.int64)
    for i in range(256):
        if i <= 20:
            y[i] = np.rint(f32(i * LAB_BASE * 20 * 9) / f32(17 * 29 * 29 * 29))
            ify[i] = np.rint(f32(LAB_BASE) * (f32(16) / f32(116) + f32(i * 5) / f32(3 * 17 * 29)))
        else:
            fy = f32(i * 100 * LAB_BASE) / f32(255 * 116) + f32(16 * LAB_BASE) / f32(116)
            ify[i] = np.rint(fy)
            y[i] = np.rint(fy * fy * fy / f32(LAB_BASE * LAB_BASE))
    return y, ify


def _trunc_div(a, b):
    """C's integer division (towards zero) on int64 arrays"""
    return np.sign(a) * (np.abs(a) // b)


def ab_to_xz(v):
    v = np.asarray(v, np.int64)
    lin = _trunc_div(v * 108, 841) - (LAB_BASE * 16 // 116 * 108 // 841)
    cube = _trunc_div(_trunc_div(v * v, LAB_BASE) * v, LAB_BASE)
    return np.where(v <= 3390, lin, cube)


def bgr2lab(img):
    """COLOR_LBGR2Lab on CV_8U (RGB2Lab_b, linear gamma table i << 3)"""
    tab, C = lab_cbrt_tab_b(), lab_fwd_coeffs()
    B, G, R = (img[..., k].astype(np.int64) << GAMMA_SHIFT for k in range(3))
    f = [tab[_descale(R * C[k, 0] + G * C[k, 1] + B * C[k, 2], LAB_SHIFT)] for k in range(3)]
    Lscale = (116 * 255 + 50) // 100
    Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) // 100)
    L = _descale(Lscale * f[1] + Lshift, LAB_SHIFT2)
    a = _descale(500 * (f[0] - f[1]) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)
    b = _descale(200 * (f[1] - f[2]) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)
    return np.clip(np.stack([L, a, b], axis=-1), 0, 255).astype(np.uint8)


def lab2bgr(lab):
    """COLOR_Lab2LBGR on CV_8U (Lab2RGBinteger, linear inverse gamma table (v * 255) >> 12)"""
    ytab, fytab = lab_to_yf_b()
    C = lab_inv_coeffs()
    LL, aa, bb = (lab[..., k].astype(np.int64) for k in range(3))
    y, ify = ytab[LL], fytab[LL]
    adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * LAB_BASE // 500
    bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * LAB_BASE // 200 + 1
    x, z = ab_to_xz(ify + adiv), ab_to_xz(ify - bdiv)
    shift = LAB_SHIFT + (BASE_SHIFT - INV_GAMMA_SHIFT)
    out = []
    for k in (2, 1, 0):                                   # B, G, R
        v = np.clip(_descale(C[k, 0] * x + C[k, 1] * y + C[k, 2] * z, shift), 0, (1 << INV_GAMMA_SHIFT) - 1)
        out.append((v * 255) >> INV_GAMMA_SHIFT)
    return np.stack(out, axis=-1).astype(np.uint8)


# ---- the CIE formulas in fp32 (what round 2 compared with; kept to measure how far the table code is from them) -------
def _f(t):
    return np.where(t > np.float32(0.008856), np.cbrt(t), np.float32(7.787) * t + np.float32(16.0 / 116.0)).astype(np.float32)


def bgr2lab_cie(img):
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


def lab2bgr_cie(lab):
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
