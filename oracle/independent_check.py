"""Independent torch-CPU evaluation of the ncnn graphs -- TEST INFRASTRUCTURE ONLY.

Purpose: the C oracle (oracle.c) cannot be pinned against ncnn itself (absent, see its
header), so it is cross-checked against a second implementation that shares no code with it:
its own .param/.bin parser (numpy) and torch.nn.functional ops.  Also generates the small
golden fixtures under tests/golden/ (python oracle/independent_check.py --write-golden).

Reference anchors: models/*.param graphs; upscale/upscale_processing.py:263-288 pre/post.
"""
import argparse
import os
import struct
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))


def parse_param(path):
    toks = open(path).read().split("\n")
    assert toks[0].strip() == "7767517"
    nl, _ = map(int, toks[1].split())
    layers = []
    for line in toks[2:2 + nl]:
        p = line.split()
        typ, name, nin, nout = p[0], p[1], int(p[2]), int(p[3])
        ins = p[4:4 + nin]
        outs = p[4 + nin:4 + nin + nout]
        kv = {}
        for t in p[4 + nin + nout:]:
            k, v = t.split("=")
            kv[int(k)] = v
        layers.append((typ, name, ins, outs, kv))
    return layers


def load_bin(layers, path):
    raw = open(path, "rb").read()
    off = 0
    params = {}
    for typ, name, _, _, kv in layers:
        if typ == "Convolution":
            cout, n = int(kv[0]), int(kv[6])
            (tag,) = struct.unpack_from("<I", raw, off)
            off += 4
            if tag == 0x01306B47:
                w = np.frombuffer(raw, "<f2", n, off).astype(np.float32)
                off += (n * 2 + 3) & ~3
            elif tag == 0:
                w = np.frombuffer(raw, "<f4", n, off).copy()
                off += n * 4
            else:
                raise ValueError(hex(tag))
            b = np.frombuffer(raw, "<f4", cout, off).copy()
            off += cout * 4
            cin = n // (cout * 9)
            params[name] = (w.reshape(cout, cin, 3, 3), b, tag)
        elif typ == "PReLU":
            n = int(kv[0])
            params[name] = np.frombuffer(raw, "<f4", n, off).copy()
            off += n * 4
    return params, off, len(raw)


def forward(layers, params, x_chw):
    import torch
    import torch.nn.functional as F

    blobs = {}
    for typ, name, ins, outs, kv in layers:
        if typ == "Input":
            blobs[outs[0]] = torch.from_numpy(np.ascontiguousarray(x_chw))[None]
        elif typ == "Split":
            for o in outs:
                blobs[o] = blobs[ins[0]]
        elif typ == "Convolution":
            w, b, _ = params[name]
            blobs[outs[0]] = F.conv2d(blobs[ins[0]], torch.from_numpy(w), torch.from_numpy(b),
                                      padding=int(kv.get(4, 0)))
        elif typ == "PReLU":
            blobs[outs[0]] = F.prelu(blobs[ins[0]], torch.from_numpy(params[name]))
        elif typ == "PixelShuffle":
            blobs[outs[0]] = F.pixel_shuffle(blobs[ins[0]], int(kv.get(0, 1)))
        elif typ == "Interp":
            assert int(kv[0]) == 1
            s = float(kv.get(1, 1.0))
            t = blobs[ins[0]]
            blobs[outs[0]] = t if s == 1.0 else F.interpolate(t, scale_factor=s, mode="nearest")
        elif typ == "BinaryOp":
            blobs[outs[0]] = blobs[ins[0]] + blobs[ins[1]]
        else:
            raise ValueError(typ)
    return blobs["output"][0].numpy()


def run_model(key, img_bgr):
    """u8 HWC BGR -> (f32 CHW pre-quantisation output, u8 HWC output)"""
    import torch
    from oracle import uvoracle

    torch.set_num_threads(max(1, os.cpu_count() or 1))
    base = os.path.join(uvoracle.MODELS_DIR, uvoracle.MODEL_FILES[key])
    layers = parse_param(base + ".param")
    params, used, size = load_bin(layers, base + ".bin")
    assert used == size, (used, size)
    x = img_bgr.transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255.0)
    with torch.no_grad():
        out = forward(layers, params, x)
    q = out.transpose(1, 2, 0) * 255  # float32, as upscale_processing.py:284
    u8 = np.clip(np.rint(q), 0, 255).astype(np.uint8)  # convertTo(CV_8U): half-even + saturate
    return out, u8


GOLDEN_CASES = [  # (model key, h, w, kind)
    ("2x", 32, 32, "smooth"), ("2x", 48, 64, "random"),
    ("4x", 32, 32, "smooth"), ("4x", 40, 24, "random"),
    ("1x", 32, 32, "smooth"), ("1x", 48, 64, "random"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write-golden", action="store_true")
    args = ap.parse_args()
    from oracle import uvoracle

    gold = {}
    for key, h, w, kind in GOLDEN_CASES:
        img = uvoracle.synthetic_frame(h, w, kind=kind)
        f_ind, u_ind = run_model(key, img)
        m = uvoracle.load_model(key)
        f_c = m.forward(uvoracle.from_pixels_normalize(img))
        u_c = m.apply_model(img)
        err = float(np.abs(f_ind - f_c).max())
        ndiff = int((u_ind != u_c).sum())
        print(f"{key} {h}x{w} {kind}: max|f32 diff|={err:.3e} u8 mismatches={ndiff}/{u_c.size} "
              f"range=[{f_c.min():.3f},{f_c.max():.3f}]")
        assert err < 1e-4
        tag = f"{key}_{h}x{w}_{kind}"
        gold[tag + "_in"] = img
        gold[tag + "_f32"] = f_ind.astype(np.float32)
        gold[tag + "_u8"] = u_ind
    # BASELINE config 1: one 256x256 frame, 2x Compact (u8 result only, to keep the fixture small)
    img = uvoracle.synthetic_frame(256, 256)
    _, u_ind = run_model("2x", img)
    u_c = uvoracle.load_model("2x").upscale_image(img)       # 256 < 960: a single tile, no border
    print(f"2x 256x256 smooth (config 1): u8 mismatches={int((u_ind != u_c).sum())}/{u_c.size}")
    assert np.abs(u_ind.astype(int) - u_c.astype(int)).max() <= 1
    gold["config1_2x_256x256_in"] = img
    gold["config1_2x_256x256_u8"] = u_ind
    if args.write_golden:
        out = os.path.join(os.path.dirname(_HERE), "tests", "golden", "independent_torch.npz")
        np.savez_compressed(out, **gold)
        print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
