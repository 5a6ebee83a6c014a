"""nlm_float_check.py -- TEST INFRASTRUCTURE ONLY: a SECOND, independent evaluation of the reference's `-m n=K` stage,
    cv2.fastNlMeansDenoisingColored(img, None, K, K, 5, 9)      /root/reference/upscale/upscale_processing.py:350-361

oracle/nlm_oracle.py restates OpenCV's INTEGER implementation (fixed-point Lab tables, `>> 5` distance bins, integer weight
table) and the HIP kernels are bit-exact against it -- but it is single-sourced and restated from memory.
This file shares no code and no approximation with it.  It evaluates what the function is DOCUMENTED to compute, in float64:

  * BGR (linear: COLOR_LBGR2Lab) -> CIE L*a*b* by the formulas of OpenCV's colour-conversion documentation (D65, the 0.412453..
    matrix, f(t) = t^(1/3) above 0.008856 and 7.787 t + 16/116 below, L = 116 f(Y) - 16 or 903.3 Y), 8-bit as the
    documentation states it (L * 255/100, a + 128, b + 128), rounded to nearest;
  * non-local means as Buades et al. define it and OpenCV's documentation parametrises it: for every pixel the weighted mean
    over the 9 x 9 search window, weight exp(-d / (h^2 cn)) with d the MEAN squared difference of the 5 x 5 patches (summed
    over the channels), weights below 0.001 dropped (OpenCV's WEIGHT_THRESHOLD), reflect-101 border -- with exact distances and
    exact exponentials, where the integer code bins the distance (`>> 5`, i.e. down to a multiple of 1.28) and rounds the
    weights to 1/103969;
  * back through the inverse formulas, rounded to nearest.

It will NOT agree with the integer path bit for bit -- the point is a stated bound: on the committed fixture
(tests/golden/nlm_float.npz, written by `python oracle/nlm_float_check.py --write-golden`) the two agree within the levels
printed by this script and asserted by tests/test_denoise.py, stage by stage and end to end.  It cannot pin OpenCV either
(parity stays UNPINNED); it takes the stage from "agrees with itself" to "agrees with two things", and the known-answer
vectors it writes (tests/golden/nlm_kat.json) are numbers anyone with a cv2 can check offline in a minute.
"""
import json
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))

M_RGB2XYZ = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float64)
WHITE = np.array([0.950456, 1.0, 1.088754], np.float64)


def lab_float(bgr_u8):
    """u8 BGR (linear) -> float64 (L in 0..100, a, b)"""
    rgb = bgr_u8[..., ::-1].astype(np.float64) / 255.0
    xyz = rgb @ M_RGB2XYZ.T / WHITE
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    Y = xyz[..., 1]
    L = np.where(Y > 0.008856, 116.0 * f[..., 1] - 16.0, 903.3 * Y)
    return np.stack([L, 500.0 * (f[..., 0] - f[..., 1]), 200.0 * (f[..., 1] - f[..., 2])], axis=-1)


def lab8(bgr_u8):
    lab = lab_float(bgr_u8)
    v = np.stack([lab[..., 0] * 255.0 / 100.0, lab[..., 1] + 128.0, lab[..., 2] + 128.0], axis=-1)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def bgr_from_lab8(lab_u8):
    lab = lab_u8.astype(np.float64)
    L, a, b = lab[..., 0] * 100.0 / 255.0, lab[..., 1] - 128.0, lab[..., 2] - 128.0
    fy = np.where(L > 903.3 * 0.008856, (L + 16.0) / 116.0, 7.787 * (L / 903.3) + 16.0 / 116.0)
    fx, fz = fy + a / 500.0, fy - b / 200.0

    def finv(t):
        return np.where(t ** 3 > 0.008856, t ** 3, (t - 16.0 / 116.0) / 7.787)
    xyz = np.stack([finv(fx), finv(fy), finv(fz)], axis=-1) * WHITE
    rgb = xyz @ np.linalg.inv(M_RGB2XYZ).T
    return np.clip(np.rint(rgb[..., ::-1] * 255.0), 0, 255).astype(np.uint8)


def _reflect101(n, idx):
    """BORDER_REFLECT_101 index map (gfedcb|abcdefgh|gfedcba), written out instead of np.pad's"""
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    m = np.mod(idx, period)
    return np.where(m >= n, period - m, m)


def nlm_float(plane_u8, h, template=5, search=9):
    """u8 [H][W] or [H][W][cn] -> float64 result before rounding, and the rounded u8"""
    a = plane_u8[..., None] if plane_u8.ndim == 2 else plane_u8
    H, W, cn = a.shape
    t, s = template // 2, search // 2
    x = a.astype(np.float64)
    ys, xs = np.arange(H), np.arange(W)
    num = np.zeros((H, W, cn))
    den = np.zeros((H, W))
    for dy in range(-s, s + 1):
        for dx in range(-s, s + 1):
            d = np.zeros((H, W))
            for ty in range(-t, t + 1):
                for tx in range(-t, t + 1):
                    p = x[_reflect101(H, ys + ty)][:, _reflect101(W, xs + tx)]
                    q = x[_reflect101(H, ys + dy + ty)][:, _reflect101(W, xs + dx + tx)]
                    d += ((p - q) ** 2).sum(axis=2)
            w = np.exp(-(d / (template * template)) / (float(h) * float(h) * cn))
            w[w < 0.001] = 0.0
            q0 = x[_reflect101(H, ys + dy)][:, _reflect101(W, xs + dx)]
            num += w[..., None] * q0
            den += w
    out = num / den[..., None]
    u8 = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return (out[..., 0], u8[..., 0]) if plane_u8.ndim == 2 else (out, u8)


def denoise_colored_float(img, h, h_color):
    lab = lab8(img)
    _, L = nlm_float(lab[..., 0], h)
    _, ab = nlm_float(np.ascontiguousarray(lab[..., 1:]), h_color)
    return bgr_from_lab8(np.concatenate([L[..., None], ab], axis=-1)), lab, L, ab


def known_answers():
    """numbers a reader can check offline: cv2.cvtColor(np.uint8([[[b, g, r]]]), cv2.COLOR_LBGR2Lab) for the colours below (OpenCV's
    8-bit result may sit one level off the rounded formula value: its table code, not a different definition); the end points of
    LabCbrtTab_b = round(2^15 f(i / 2040)); the end points of the integer weight table round(103969 exp(-1.28 ad / (h^2 cn)))"""
    colours = {"black": (0, 0, 0), "white": (255, 255, 255), "blue": (255, 0, 0), "green": (0, 255, 0), "red": (0, 0, 255),
               "grey64": (64, 64, 64), "grey128": (128, 128, 128), "grey192": (192, 192, 192), "yellow": (0, 255, 255),
               "cyan": (255, 255, 0), "magenta": (255, 0, 255)}
    kat = {"lab8_of_bgr_by_the_cie_formulas": {}, "lab_float_of_bgr": {}}
    for name, bgr in colours.items():
        px = np.array([[bgr]], np.uint8)
        kat["lab8_of_bgr_by_the_cie_formulas"][name] = {"bgr": list(bgr), "lab8": [int(v) for v in lab8(px)[0, 0]]}
        kat["lab_float_of_bgr"][name] = [round(float(v), 4) for v in lab_float(px)[0, 0]]

    def f(t):
        return t ** (1.0 / 3.0) if t > 216.0 / 24389.0 else (841.0 / 108.0) * t + 16.0 / 116.0
    n = 256 * 3 // 2 * 8
    kat["LabCbrtTab_b"] = {"size": n, "scale": "round(32768 * f(i / (255 * 8)))", "first": int(round(32768 * f(0.0))),
                           "at_2040_is_one": int(round(32768 * f(1.0))), "last": int(round(32768 * f((n - 1) / 2040.0)))}
    fixed = (2 ** 31 - 1) // (81 * 255)
    wt = {}
    for h in (1, 3, 10, 30):
        for cn in (1, 2):
            # last index with a weight >= fixed / 1000: exp(-1.28 ad / (h^2 cn)) >= 0.001  <=>  ad <= ln(1000) h^2 cn / 1.28 (up to the rounding)
            ad = np.arange(int(255 * 255 * cn / 1.28 + 1), dtype=np.float64)
            tab = np.rint(fixed * np.exp(-1.28 * ad / (h * h * cn)))
            tab[tab < 0.001 * fixed] = 0
            nz = np.nonzero(tab)[0]
            wt[f"h={h},cn={cn}"] = {"size": int(len(tab)), "at_0": int(tab[0]), "at_1": int(tab[1]), "last_nonzero_index": int(nz[-1]),
                                    "last_nonzero_value": int(tab[nz[-1]])}
    kat["weight_table"] = {"fixed_point_mult": fixed, "entries": wt}
    return kat


def fixture():
    """a 40 x 56 frame with flat areas, edges and grain (noise is what the stage is for), K = 3 and 10"""
    rng = np.random.default_rng(2026)
    yy, xx = np.mgrid[0:40, 0:56]
    base = np.stack([90 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 128 + 50 * np.cos(xx / 11.0 + 1), 100 + 70 * (xx > 28) * (yy > 15)], axis=-1)
    return np.clip(np.rint(base + rng.normal(0, 5, base.shape)), 0, 255).astype(np.uint8)


def main():
    from oracle import nlm_oracle as no
    write = "--write-golden" in sys.argv
    img = fixture()
    gold = {"in": img}
    for K in (3, 10):
        f_out, f_lab, f_L, f_ab = denoise_colored_float(img, K, K)
        i_lab = no.bgr2lab(img)
        i_L, i_ab = no.nlm_plane(i_lab[..., 0], K), no.nlm_plane(np.ascontiguousarray(i_lab[..., 1:]), K)
        i_out = no.denoise_colored(img, K, K)
        # stage by stage on the SAME stage input (the integer path's), so that a stage's distance is its own
        _, fL_on_i = nlm_float(i_lab[..., 0], K)
        _, fab_on_i = nlm_float(np.ascontiguousarray(i_lab[..., 1:]), K)
        fback_on_i = bgr_from_lab8(np.concatenate([i_L[..., None], i_ab], axis=-1))
        d = lambda a, b: np.abs(a.astype(int) - b.astype(int))     # noqa: E731
        print(f"K={K}: Lab8 table code vs formulas: max {d(i_lab, f_lab).reshape(-1, 3).max(0)}  | NLM L max {d(i_L, fL_on_i).max()} "
              f"({(d(i_L, fL_on_i) > 0).mean():.3%} differ)  ab max {d(i_ab, fab_on_i).max()} ({(d(i_ab, fab_on_i) > 0).mean():.3%})  | "
              f"Lab8->BGR max {d(i_out, fback_on_i).max()} (mean {d(i_out, fback_on_i).mean():.3f})  | end to end max {d(i_out, f_out).max()} "
              f"mean {d(i_out, f_out).mean():.3f}, {(d(i_out, f_out) > 1).mean():.3%} beyond one level")
        gold[f"K{K}_float_u8"] = f_out
        gold[f"K{K}_float_lab8"] = f_lab
        gold[f"K{K}_float_nlm_L_on_integer_lab"] = fL_on_i
        gold[f"K{K}_float_nlm_ab_on_integer_lab"] = fab_on_i
    kat = known_answers()
    print(json.dumps(kat["LabCbrtTab_b"]), json.dumps(kat["weight_table"]["entries"]["h=3,cn=1"]))
    if write:
        out = os.path.join(os.path.dirname(_HERE), "tests", "golden", "nlm_float.npz")
        np.savez_compressed(out, **gold)
        print("wrote", out, os.path.getsize(out), "bytes")
        out = os.path.join(os.path.dirname(_HERE), "tests", "golden", "nlm_kat.json")
        with open(out, "w") as fh:
            json.dump(kat, fh, indent=1, sort_keys=True)
        print("wrote", out)


if __name__ == "__main__":
    main()
