"""CPU tests of the oracle (oracle/oracle.c): loader known-answer tests, per-op semantics, the
independent-torch golden vectors, and the reference's tiling logic.  The reference has no tests
or golden vectors for this path: parity is unpinned at the ncnn boundary and
these are the pins the build created."""
import os

import numpy as np
import pytest

from conftest import ROOT, psnr_u8

BIN_SIZES = {"2x": 1205752, "4x": 2485768, "1x": 87316}


@pytest.mark.parametrize("key", ["2x", "4x", "1x"])
def test_loader_consumes_every_byte(oracle_models, key):
    m = oracle_models[key]
    assert m.bin_size == BIN_SIZES[key]
    assert m.bin_consumed == m.bin_size


def test_graph_facts(oracle_models):
    m2, m4, m1 = oracle_models["2x"], oracle_models["4x"], oracle_models["1x"]
    assert (m2.scale, m2.nf, m2.num_conv, m2.num_layers) == (2, 64, 18, 40)
    assert (m4.scale, m4.nf, m4.num_conv, m4.num_layers) == (4, 64, 18, 40)
    assert (m1.scale, m1.nf, m1.num_conv, m1.num_layers) == (1, 24, 10, 24)


def test_weight_known_answers(oracle_models):
    # tags: fp16 payload for 2x / 1x, raw fp32 for 4x
    for i in range(18):
        assert oracle_models["2x"].conv(i)[2] == 0x01306B47
        assert oracle_models["4x"].conv(i)[2] == 0
    for i in range(10):
        assert oracle_models["1x"].conv(i)[2] == 0x01306B47
    w0, b0, _ = oracle_models["2x"].conv(0)
    assert w0.shape == (64, 3, 3, 3) and b0.shape == (64,)
    assert w0.min() == pytest.approx(-41.75) and w0.max() == pytest.approx(32.72, abs=0.01)
    s = oracle_models["2x"].prelu(16)   # PRelu_33
    assert s.shape == (64,)
    assert s.min() == pytest.approx(-0.0001, abs=1e-4) and s.max() == pytest.approx(0.7056, abs=1e-4)
    assert oracle_models["2x"].conv(17)[0].shape == (12, 64, 3, 3)
    assert oracle_models["4x"].conv(17)[0].shape == (48, 64, 3, 3)
    assert oracle_models["1x"].conv(9)[0].shape == (3, 24, 3, 3)
    # fp16-tagged weights are exactly representable in fp16
    w = oracle_models["1x"].conv(3)[0]
    assert np.array_equal(w, w.astype(np.float16).astype(np.float32))


def test_round_f16_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.normal(0, 30, 2000), rng.normal(0, 1e-5, 500), [0.0, 65504.0, 65519.9, 65520.0, 1e-8,
                         2.0 ** -24, 2.0 ** -25, 3 * 2.0 ** -25, 1 + 2.0 ** -11, 1 + 3 * 2.0 ** -11]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).astype(np.float32)
    got = np.array([oracle.round_f16(float(x)) for x in xs], np.float32)
    assert np.array_equal(got, want)


def test_from_pixels_keeps_bgr_order_and_scales(oracle):
    img = np.zeros((2, 3, 3), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 10, 20, 255
    x = oracle.from_pixels_normalize(img)
    n = np.float32(1 / 255.0)
    assert x.shape == (3, 2, 3)
    assert np.all(x[0] == np.float32(10) * n) and np.all(x[1] == np.float32(20) * n)
    assert np.all(x[2] == np.float32(255) * n)


def test_quantise_is_round_half_even_saturating(oracle):
    vals = np.array([0.5, 1.5, 2.5, 3.5, -3.0, 254.5, 255.5, 300.0, 127.49999], np.float32) / np.float32(255)
    # make the products land exactly on the intended values where possible
    chw = np.zeros((3, 1, len(vals)), np.float32)
    chw[0, 0] = vals
    out = oracle.to_u8(chw)[0, :, 0]
    prod = chw[0, 0] * np.float32(255)
    want = np.clip(np.rint(prod), 0, 255).astype(np.uint8)
    assert np.array_equal(out, want)
    # explicit ties (exactly representable inputs): k + 0.5 -> even
    for k in (0, 1, 2, 3, 100, 253):
        v = np.float32(k + 0.5)
        chw[:] = 0
        # find x with x*255 == k+0.5 exactly is not always possible; test the rounding rule on the product
        assert np.rint(v) == (k if k % 2 == 0 else k + 1)


def test_pixelshuffle_interp_add_semantics(oracle_models, oracle):
    """2x graph on a constant image: with zero trunk contribution impossible to isolate, so check
    the structural identity instead: output(y, x) - input(y//2, x//2) is the shuffled conv term,
    and the torch golden (F.pixel_shuffle / nearest) agrees -- covered by test_golden; here check
    nearest-upsample + add on the 1x model, whose PixelShuffle/Interp are identities."""
    m = oracle_models["1x"]
    img = oracle.synthetic_frame(16, 24, kind="random")
    x = oracle.from_pixels_normalize(img)
    out = m.forward(x)
    tail = m.tap(x, 9)            # last conv (no PReLU follows): [3, h, w]
    assert np.allclose(out, tail + x, atol=0, rtol=0)


@pytest.mark.parametrize("key", ["2x", "4x"])
def test_pixelshuffle_order(oracle_models, oracle, key):
    m = oracle_models[key]
    r = m.scale
    img = oracle.synthetic_frame(12, 20, kind="random")
    x = oracle.from_pixels_normalize(img)
    out = m.forward(x)
    tail = m.tap(x, 17)           # [3*r*r, h, w]
    h, w = 12, 20
    shuf = tail.reshape(3, r, r, h, w).transpose(0, 3, 1, 4, 2).reshape(3, h * r, w * r)
    up = np.repeat(np.repeat(x, r, axis=1), r, axis=2)
    assert np.array_equal(out, shuf + up)


def test_golden_independent_torch(oracle_models, oracle):
    g = np.load(os.path.join(ROOT, "tests", "golden", "independent_torch.npz"))
    tags = sorted({k[:-3] for k in g.files if k.endswith("_in") and k.split("_")[0] in ("1x", "2x", "4x")})
    assert len(tags) == 6
    # the tiled path (all four border branches) and BASELINE config 3's chain through the independent evaluation's OWN tiling loop
    t = oracle_models["2x"].upscale_image(g["tiled_2x_70x75_t32_in"], tile_size=32, border=10)
    assert np.abs(t.astype(int) - g["tiled_2x_70x75_t32_u8"].astype(int)).max() <= 1 and (t != g["tiled_2x_70x75_t32_u8"]).mean() <= 1e-3
    c = oracle_models["2x"].upscale_image(oracle_models["1x"].apply_model(g["chain_1x_2x_48x64_t32_in"]), tile_size=32, border=10)
    assert np.abs(c.astype(int) - g["chain_1x_2x_48x64_t32_u8"].astype(int)).max() <= 1 and (c != g["chain_1x_2x_48x64_t32_u8"]).mean() <= 1e-3
    # BASELINE config 1 (256x256, 2x Compact): u8 result of the independent evaluation
    ws = oracle_models["2x"].upscale_image(g["wino_seams_2x_200x190_t64_in"], tile_size=64, border=10)   # round 5: strip + tile seams
    assert np.abs(ws.astype(int) - g["wino_seams_2x_200x190_t64_u8"].astype(int)).max() <= 1 and (ws != g["wino_seams_2x_200x190_t64_u8"]).mean() <= 1e-3
    # ... and the oracle's own Winograd / fp16 modes (the rounding points of the product's trunk kernel) stay inside the fp32 bar
    from oracle import uvoracle
    wp = oracle_models["2x"].upscale_image(g["wino_seams_2x_200x190_t64_in"], tile_size=64, border=10,
                                           flags=uvoracle.F16_STORAGE | uvoracle.WINOGRAD_F23 | uvoracle.PRELU_F16)
    dp = np.abs(wp.astype(int) - g["wino_seams_2x_200x190_t64_u8"].astype(int))
    assert dp.max() <= 2 and 10 * np.log10(255.0 ** 2 / (dp.astype(float) ** 2).mean()) >= 50
    c1 = oracle_models["2x"].upscale_image(g["config1_2x_256x256_in"])
    assert np.abs(c1.astype(int) - g["config1_2x_256x256_u8"].astype(int)).max() <= 1
    assert (c1 != g["config1_2x_256x256_u8"]).mean() <= 1e-3
    for tag in tags:
        key = tag.split("_")[0]
        m = oracle_models[key]
        img = g[tag + "_in"]
        f = m.forward(oracle.from_pixels_normalize(img))
        assert np.abs(f - g[tag + "_f32"]).max() < 1e-4, tag
        u = m.apply_model(img)
        mism = (u != g[tag + "_u8"])
        assert mism.mean() <= 1e-3, tag                      # only exact .5 ties may flip
        assert np.abs(u.astype(int) - g[tag + "_u8"].astype(int)).max() <= 1, tag


def test_tiling_all_border_branches(oracle_models, oracle):
    """upscale_image/process_tile with a 32-px tile and 10-px border on a 75x70 image exercises
    every branch of upscale_processing.py:409-427 on both axes; compare with a literal numpy
    re-enactment of :398-477 built on the oracle's forward()."""
    m = oracle_models["2x"]
    h, w, ts, b, s = 70, 75, 32, 10, 2
    img = oracle.synthetic_frame(h, w, kind="random")
    got = m.upscale_image(img, tile_size=ts, border=b)
    canvas = np.zeros((h * s, w * s, 3))
    import math
    for ty in range(math.ceil(h / ts)):
        for tx in range(math.ceil(w / ts)):
            y0, x0 = ty * ts, tx * ts
            y1, x1 = min(y0 + ts, h), min(x0 + ts, w)
            by0 = -b if y0 >= b else 0
            by1 = b if y1 <= h - b else 0
            bx0 = -b if x0 >= b else 0
            bx1 = b if x1 <= w - b else 0
            tile = img[y0 + by0:y1 + by1, x0 + bx0:x1 + bx1].copy()
            out = m.forward(oracle.from_pixels_normalize(tile)).transpose(1, 2, 0) * 255
            canvas[y0 * s:y1 * s, x0 * s:x1 * s] = out[-by0 * s:(y1 - y0 - by0) * s, -bx0 * s:(x1 - x0 - bx0) * s]
    want = np.clip(np.rint(canvas), 0, 255).astype(np.uint8)
    assert np.array_equal(got, want)
    # and tiling is a (tiny) approximation of the whole-frame result, as the survey measured
    # (on photo-like content; on white noise the 10-px border is visibly short of the 18-px
    # receptive field, which is the reference's behaviour, not ours to fix)
    smooth = oracle.synthetic_frame(h, w)
    tiled, whole = m.upscale_image(smooth, tile_size=ts, border=b), m.apply_model(smooth)
    assert psnr_u8(tiled, whole) > 45


def test_full_size_tile_shapes():
    """1080p with the reference's 960/10 tiling -> (970,970)x2 + (130,970)x2  (SURVEY.md section 5)."""
    from upscale_video_amd.upscale_processing import tile_window
    shapes = []
    for y in range(2):
        for x in range(2):
            (y0, y1, x0, x1), (t, bo, l, r) = tile_window(960, y, x, 1080, 1920)
            shapes.append((y1 - y0 + t + bo, x1 - x0 + l + r))
    assert shapes == [(970, 970), (970, 970), (130, 970), (130, 970)]


def test_f16_storage_mode_is_close_to_fp32(oracle_models, oracle):
    m = oracle_models["2x"]
    img = oracle.synthetic_frame(40, 48)
    a = m.apply_model(img)
    b = m.apply_model(img, flags=oracle.F16_STORAGE)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1
    assert psnr_u8(a, b) > 55


def test_chain_1x_then_2x_is_quantised_between(oracle_models, oracle):
    """'-m a' then upscale: the 1x output is written as an 8-bit PNG before the 2x pass reads it
    (upscale_processing.py:888-909), so the chain re-quantises to u8 in between."""
    img = oracle.synthetic_frame(24, 40)
    mid = oracle_models["1x"].apply_model(img)
    assert mid.dtype == np.uint8 and mid.shape == img.shape
    out = oracle_models["2x"].upscale_image(mid)
    assert out.shape == (48, 80, 3)


@pytest.mark.parametrize("key", ["2x", "4x"])
def test_prelu_on_halves_mode(oracle_models, oracle, key):
    """UVO_PRELU_F16 (the product's trunkw_kernel applies PReLU to the sum rounded to fp16): against the one-rounding PReLU of
    the same Winograd convolution, an activation changes by at most one fp16 step, and only where it is the product
    slope * x (x < 0): the first layer of the first fused pair, whose input is the same in both modes, shows it directly --
    the 4x net has slopes above 1 (min instead of max) among them."""
    m = oracle_models[key]
    x = oracle.from_pixels_normalize(oracle.synthetic_frame(20, 24, seed=5))
    base = oracle.F16_STORAGE | oracle.WINOGRAD_F23
    a = m.tap(x, 2, flags=base)                       # taps at the second layer of a pair: both layers ran as Winograd
    pre1 = m.tap(x, 1, flags=oracle.F16_STORAGE)      # (layer 1 alone, direct: the same input for both modes' layer 2)
    b = m.tap(x, 2, flags=base | oracle.PRELU_F16)
    assert a.shape == b.shape == pre1.shape
    ulp = np.spacing(np.abs(a).astype(np.float16)).astype(np.float32)
    # two layers deep a one-step difference of layer 1 has passed through layer 2's sums: a few steps, not more
    assert np.abs(a - b).max() <= 8 * ulp.max()
    assert np.mean(a != b) < 0.5
    # and the two modes agree closely as images of the net
    out_a, out_b = m.forward(x, flags=base), m.forward(x, flags=base | oracle.PRELU_F16)
    assert np.abs(out_a - out_b).max() < 2e-2
