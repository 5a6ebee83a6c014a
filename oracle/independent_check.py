"""Independent torch-CPU evaluation of the ncnn graphs -- TEST INFRASTRUCTURE ONLY.

Purpose: the C oracle (oracle.c) and the numpy restatement for generic graphs (generic_oracle.py) cannot be pinned
against ncnn itself (absent, see their headers), so they are cross-checked against a second implementation that shares no
code with them: its own .param/.bin parser (numpy) and torch.nn.functional ops -- including the ops only 4x_Valar_v1 has
(Concat, Eltwise with coefficients, LeakyReLU fused into a convolution, bias-less 1x1 convolution, nearest x2 Interp:
/root/reference/models/4x_Valar_v1.param:6-21,1203-1208), its own tiling loop, and the 1x -> 2x chain.  Also generates the small
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


def _channels(layers):
    """input channel count of every Convolution, by walking the graph (the .param does not state it)"""
    ch, cin = {}, {}
    for typ, name, ins, outs, kv in layers:
        if typ == "Input":
            ch[outs[0]] = 3
        elif typ == "Convolution":
            cin[name] = ch[ins[0]]
            ch[outs[0]] = int(kv[0])
        elif typ == "Concat":
            ch[outs[0]] = sum(ch[i] for i in ins)
        elif typ == "PixelShuffle":
            f = int(kv.get(0, 1))
            ch[outs[0]] = ch[ins[0]] // (f * f)
        else:
            for o in outs:
                ch[o] = ch[ins[0]]
    return cin


def load_bin(layers, path):
    raw = open(path, "rb").read()
    off = 0
    params = {}
    cins = _channels(layers)
    for typ, name, _, _, kv in layers:
        if typ == "Convolution":
            cout, n, k = int(kv[0]), int(kv[6]), int(kv.get(1, 1))
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
            if int(kv.get(5, 0)):
                b = np.frombuffer(raw, "<f4", cout, off).copy()
                off += cout * 4
            else:
                b = None
            cin = cins[name]
            assert n == cout * cin * k * k, name
            params[name] = (w.reshape(cout, cin, k, k), b, tag)
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
            y = F.conv2d(blobs[ins[0]], torch.from_numpy(w), None if b is None else torch.from_numpy(b), padding=int(kv.get(4, 0)))
            if int(kv.get(9, 0)) == 2:          # activation_type 2 = LeakyReLU, slope in -23310=1,slope
                y = F.leaky_relu(y, float(kv[-23310].split(",")[1]))
            else:
                assert int(kv.get(9, 0)) == 0
            blobs[outs[0]] = y
        elif typ == "Concat":                   # axis 0 of the CHW blob = channels
            assert int(kv.get(0, 0)) == 0
            blobs[outs[0]] = torch.cat([blobs[i] for i in ins], dim=1)
        elif typ == "Eltwise":                  # op_type 1 = SUM with coefficients (-23301=n,c0,c1)
            assert int(kv[0]) == 1
            c = [float(v) for v in kv[-23301].split(",")[1:]] if -23301 in kv else [1.0] * len(ins)
            assert len(c) == len(ins)
            acc = blobs[ins[0]] * c[0]
            for i, ci in zip(ins[1:], c[1:]):
                acc = acc + blobs[i] * ci
            blobs[outs[0]] = acc
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
            assert int(kv.get(0, 0)) == 0       # ADD
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


def run_graph(param_path, bin_path, img_bgr):
    """any graph this file knows the layers of: u8 HWC BGR -> (f32 CHW output, u8 HWC output)"""
    import torch
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    layers = parse_param(param_path)
    params, used, size = load_bin(layers, bin_path)
    assert used == size, (used, size)
    x = img_bgr.transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255.0)
    with torch.no_grad():
        out = forward(layers, params, x)
    q = out.transpose(1, 2, 0) * 255
    return out, np.clip(np.rint(q), 0, 255).astype(np.uint8)


def tiled(run, scale, img, tile_size, border):
    """upscale_image / process_tile (upscale/upscale_processing.py:395-519) restated here a second time: the tile grid,
    the four border rules, the crop of the scaled tile; `run` maps a u8 tile to its u8 result."""
    h, w, _ = img.shape
    out = np.zeros((h * scale, w * scale, 3), np.uint8)
    for ty in range(-(-h // tile_size)):
        for tx in range(-(-w // tile_size)):
            y0, x0 = ty * tile_size, tx * tile_size
            y1, x1 = min(y0 + tile_size, h), min(x0 + tile_size, w)
            t, l = (border if y0 >= border else 0), (border if x0 >= border else 0)
            b, r = (border if y1 <= h - border else 0), (border if x1 <= w - border else 0)
            res = run(np.ascontiguousarray(img[y0 - t:y1 + b, x0 - l:x1 + r]))
            out[y0 * scale:y1 * scale, x0 * scale:x1 * scale] = res[t * scale:(t + y1 - y0) * scale, l * scale:(l + x1 - x0) * scale]
    return out


VALAR_SEED, VALAR_GAIN = 11, 0.5      # upscale_video_amd.synth.synthetic_weights(param, bin, seed, gain): the .bin is a missing blob upstream


def valar_fixture():
    """4x_Valar_v1 with seeded synthetic weights (the ops the Compact graphs do not have: Concat, Eltwise with
    coefficients, LeakyReLU fused into a convolution, the bias-less 1x1 convolution, nearest x2 Interp) on two frames."""
    import tempfile
    from oracle import uvoracle
    from upscale_video_amd import synth
    param = os.path.join(uvoracle.MODELS_DIR, "4x_Valar_v1.param")
    gold = {"valar_seed_gain": np.array([VALAR_SEED, VALAR_GAIN], np.float64)}
    with tempfile.TemporaryDirectory() as d:
        wb = os.path.join(d, "v.bin")
        synth.synthetic_weights(param, wb, seed=VALAR_SEED, gain=VALAR_GAIN)
        for h, w in ((12, 20), (70, 75)):
            img = uvoracle.synthetic_frame(h, w, seed=5)
            f, u = run_graph(param, wb, img)
            print(f"4x_Valar_v1 (synthetic weights) {w}x{h}: output range [{f.min():.3f}, {f.max():.3f}]")
            gold[f"valar_{h}x{w}_in"] = img
            gold[f"valar_{h}x{w}_f32"] = f.astype(np.float32)
            gold[f"valar_{h}x{w}_u8"] = u
    return gold


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
    # the tiled path (75x70, tile 32, border 10: all four border branches) and BASELINE config 3's chain (1x -> u8 -> 2x),
    # through this file's own tiling loop
    img = uvoracle.synthetic_frame(70, 75, seed=9)
    u_t = tiled(lambda t: run_model("2x", t)[1], 2, img, 32, 10)
    u_c = uvoracle.load_model("2x").upscale_image(img, tile_size=32, border=10)
    print(f"2x 75x70 tiled 32/10: u8 mismatches={int((u_t != u_c).sum())}/{u_c.size}")
    assert np.abs(u_t.astype(int) - u_c.astype(int)).max() <= 1
    gold["tiled_2x_70x75_t32_in"], gold["tiled_2x_70x75_t32_u8"] = img, u_t
    img = uvoracle.synthetic_frame(48, 64, seed=10)
    u_ch = tiled(lambda t: run_model("2x", t)[1], 2, run_model("1x", img)[1], 32, 10)
    u_cc = uvoracle.load_model("2x").upscale_image(uvoracle.load_model("1x").apply_model(img), tile_size=32, border=10)
    print(f"chain 1x -> 2x 64x48 tiled 32/10: u8 mismatches={int((u_ch != u_cc).sum())}/{u_cc.size}")
    assert np.abs(u_ch.astype(int) - u_cc.astype(int)).max() <= 1
    gold["chain_1x_2x_48x64_t32_in"], gold["chain_1x_2x_48x64_t32_u8"] = img, u_ch
    # round 5: a tiled frame on which BOTH layers of every fused trunk pair see tile seams, strip seams and
    # narrow last strips -- 190 columns x 200 rows, tile 64, border 10: planes 74 / 84 / 72 columns wide (two 30-column strips +
    # a last strip of 14 / 24 / 12 columns) and 74 / 84 / 84 / 18 rows high.  fp32 throughout, like every fixture of this file:
    # the bar for the fp16 Winograd product path against it is the fp32 one (<= 2 LSB, >= 50 dB).
    img = uvoracle.synthetic_frame(200, 190, seed=12)
    u_t = tiled(lambda t: run_model("2x", t)[1], 2, img, 64, 10)
    u_c = uvoracle.load_model("2x").upscale_image(img, tile_size=64, border=10)
    print(f"2x 190x200 tiled 64/10: u8 mismatches={int((u_t != u_c).sum())}/{u_c.size}")
    assert np.abs(u_t.astype(int) - u_c.astype(int)).max() <= 1
    gold["wino_seams_2x_200x190_t64_in"], gold["wino_seams_2x_200x190_t64_u8"] = img, u_t
    vgold = valar_fixture()
    if args.write_golden:
        out = os.path.join(os.path.dirname(_HERE), "tests", "golden", "independent_torch.npz")
        np.savez_compressed(out, **gold)
        print("wrote", out, os.path.getsize(out), "bytes")
        out = os.path.join(os.path.dirname(_HERE), "tests", "golden", "valar_synthetic.npz")
        np.savez_compressed(out, **vgold)
        print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
