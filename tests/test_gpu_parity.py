"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (include/uva.h via the
ctypes shim), against the CPU oracle on the same seeded inputs, the committed golden vectors, and
size-independent properties at the full BASELINE sizes.

Stated tolerances (north_star: within 0.5 dB PSNR / fp32 tolerance of the reference):
  * vs the fp32 oracle: u8 output PSNR >= 50 dB and max |diff| <= 2 LSB; f32 pre-quantisation
    output max |diff| <= 6e-3 (1.5 LSB).  The HIP path stores activations and weights in fp16
    (fp32 accumulate) -- the 2x/1x weights are fp16 on disk, ncnn-vulkan itself defaults to fp16
    storage -- so bit-exactness with an fp32 CPU run is not defined for this floating-point path.
  * vs the oracle in the product's own storage mode (oracle.product_flags(): fp16 storage, and the fused trunk
    pairs as Winograd F(2,3) with the kernel's rounding points unless UVA_TRUNK_WINO=0 -- differences only
    from fp32 summation order): per-layer activations within 2e-3 * max|act| (+1 fp16 ulp),
    u8 output equal except <= 5 % of samples (8 % with the Winograd trunk; white-noise inputs) off by 1 LSB.
"""
import os

import numpy as np
import pytest

try:  # torch first: it must bring up its own HIP runtime before libuva.so pulls in /opt/rocm's
    import torch
except Exception:  # noqa: BLE001
    torch = None

from conftest import ROOT, load_net, psnr_u8
from parity_report import check_f32, check_u8, fp32_bar, record, slack

pytestmark = pytest.mark.gpu
FP32 = "fp32 oracle"
PRODUCT = "oracle, product rounding mode"


@pytest.fixture(scope="module")
def nets(uva):
    assert uva.get_gpu_count() > 0, "no HIP device: the HIP path cannot run (no CPU fallback exists)"
    return {k: load_net(uva, k) for k in ("2x", "4x", "1x")}


def test_device_enumeration(uva):
    assert uva.get_gpu_count() >= 1
    assert uva.get_default_gpu_index() == 0
    info = uva.get_gpu_info(0)
    assert info.type() == 0 and "gfx950" in info.device_name()


def _wino(oracle):
    return bool(oracle.product_flags() & oracle.WINOGRAD_F23)


def U8_DIFFER(oracle, key, route="whole"):
    """share of u8 samples that may differ (by one level) from the oracle in the product's rounding mode: the MEASURED maximum
    of the committed sweep for this model and route plus one point (tests/golden/parity_slack.json, tools/parity_slack.py);
    without an entry, the round numbers of round 4 (5 %; 8 % where the trunk runs as Winograd F(2,3))"""
    return slack(key, route, "u8_differ_share", 8e-2 if _wino(oracle) and key != "1x" else 5e-2)


@pytest.mark.parametrize("key", ["2x", "1x"])
def test_per_layer_activations(nets, oracle_models, oracle, key):
    """Every layer of the trunk against the oracle's tap of the same layer, in the product's storage mode (float route:
    the input blob is rounded to fp16 too).  Rounding flips caused by the fp32 summation order grow from layer to layer
    on white noise (by conv 8 most values differ in their last bit, in either mode), so the bar has two parts: the head
    and the FIRST fused pair -- where the kernel's arithmetic shows undisturbed -- agree except for single-ulp flips on
    a few per cent of the values, and every layer stays within 2e-3 (direct) / 3e-3 (Winograd F(2,3): its transforms
    double the rounding noise of a layer) of the layer's range."""
    net, om = nets[key], oracle_models[key]
    h, w = 37, 70          # ragged: 5 tile rows, 3 tile columns, partial tiles on both axes
    img = oracle.synthetic_frame(h, w, kind="random")
    x = oracle.from_pixels_normalize(img)
    net._extract(x)        # sets the replay source
    flags = oracle.product_flags("f32")
    rel = slack(key, "float", "layer_rel", 3e-3 if (_wino(oracle) and key != "1x") else 2e-3)
    worst = []
    for idx in range(net.num_convs - 1):
        got = net.debug_read_activation(idx, h, w)
        want = om.tap(x, idx, flags=flags)
        scale = float(np.abs(want).max())
        err = check_f32(f"{key} conv {idx} activations 70x37", got, want, vs=PRODUCT, max_abs=rel * scale + 2e-3, scale=scale,
                        model=key, route="float", what="layer_rel")
        worst.append((idx, err, scale))
        if idx <= 2:
            flips = float(((got - want) != 0).mean())        # single-bit rounding flips from the fp32 summation order
            assert flips <= 0.05 and err <= 1e-3 * scale, (key, idx, flips, worst)


@pytest.mark.parametrize("key,h,w", [("2x", 37, 70), ("4x", 21, 45), ("1x", 50, 33), ("2x", 8, 32), ("2x", 1, 1),
                                     ("1x", 3, 2), ("4x", 9, 33), ("2x", 16, 64)])
def test_extract_f32_matches_oracle(nets, oracle_models, oracle, key, h, w):
    net, om = nets[key], oracle_models[key]
    img = oracle.synthetic_frame(h, w, kind="random", seed=h * 1000 + w)
    x = oracle.from_pixels_normalize(img)
    got = net._extract(x)
    want32 = om.forward(x)
    want16 = om.forward(x, flags=oracle.product_flags("f32"))
    assert got.shape == want32.shape
    check_f32(f"{key} extract {w}x{h}", got, want32, vs=FP32, max_abs=6e-3, model=key, route="float")
    check_f32(f"{key} extract {w}x{h}", got, want16, vs=PRODUCT, model=key, route="float",
              max_abs=slack(key, "float", "f32_abs", 4e-3 if _wino(oracle) and key != "1x" else 3e-3))


@pytest.mark.parametrize("key,h,w,kind", [("2x", 37, 70, "random"), ("2x", 64, 96, "smooth"), ("4x", 21, 45, "random"),
                                          ("1x", 50, 33, "random"), ("1x", 40, 72, "smooth"), ("2x", 1, 1, "random"),
                                          ("4x", 2, 35, "smooth")])
def test_process_u8_whole_frame_matches_oracle(nets, oracle_models, oracle, key, h, w, kind):
    """apply_model arithmetic (upscale_processing.py:263-288), fused on the device."""
    net, om = nets[key], oracle_models[key]
    img = oracle.synthetic_frame(h, w, kind=kind, seed=7 * h + w)
    got = net.process_u8(img, tile_size=0)
    want32 = om.apply_model(img)
    want16 = om.apply_model(img, flags=oracle.product_flags())
    assert got.shape == want32.shape and got.dtype == np.uint8
    check_u8(f"{key} apply_model {w}x{h} {kind}", got, want32, vs=FP32, model=key, route="whole", **fp32_bar(key, "whole"))
    check_u8(f"{key} apply_model {w}x{h} {kind}", got, want16, vs=PRODUCT, max_lsb=1, max_share=U8_DIFFER(oracle, key, "whole"),
             model=key, route="whole")


def test_golden_vectors(nets):
    """Committed fixtures generated by oracle/independent_check.py (independent torch evaluation)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "independent_torch.npz"))
    tags = sorted({k[:-3] for k in g.files if k.endswith("_in") and k.split("_")[0] in ("1x", "2x", "4x")})
    # the tiled path (75x70, 32/10: all four border branches) and config 3's chain, from the independent evaluation's own tiling loop
    GOLD = "golden fixture (independent torch fp32)"
    t = nets["2x"].process_u8(g["tiled_2x_70x75_t32_in"], tile_size=32, border=10)
    check_u8("golden tiled_2x_70x75_t32", t, g["tiled_2x_70x75_t32_u8"], vs=GOLD, max_lsb=2, min_psnr=50, model="2x", route="tiled")
    c = nets["2x"].process_u8(nets["1x"].process_u8(g["chain_1x_2x_48x64_t32_in"], tile_size=0), tile_size=32, border=10)
    check_u8("golden chain_1x_2x_48x64_t32", c, g["chain_1x_2x_48x64_t32_u8"], vs=GOLD, max_lsb=2, min_psnr=50, model="chain", route="tiled")
    c1 = nets["2x"].process_u8(g["config1_2x_256x256_in"], tile_size=960, border=10)   # BASELINE config 1
    check_u8("golden config1_2x_256x256", c1, g["config1_2x_256x256_u8"], vs=GOLD, max_lsb=2, min_psnr=50, model="2x", route="tiled")
    if "wino_seams_2x_200x190_t64_in" in g.files:
        # the Winograd path where BOTH layers of a fused pair see tile seams and strip seams: 190 columns = 7 strips per
        # plane, 64-px tiles with the 10-px border = 9 planes of up to 84 columns (3 strips each), segment cuts in between
        ws = nets["2x"].process_u8(g["wino_seams_2x_200x190_t64_in"], tile_size=64, border=10)
        check_u8("golden wino_seams_2x_200x190_t64", ws, g["wino_seams_2x_200x190_t64_u8"], vs=GOLD, max_lsb=2, min_psnr=50,
                 model="2x", route="tiled")
    for tag in tags:
        net = nets[tag.split("_")[0]]
        got = net.process_u8(g[tag + "_in"], tile_size=0)
        check_u8("golden " + tag, got, g[tag + "_u8"], vs=GOLD, max_lsb=2, min_psnr=50, model=tag.split("_")[0], route="whole")
        x = g[tag + "_in"].transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255.0)
        check_f32("golden " + tag, net._extract(x), g[tag + "_f32"], vs=GOLD, max_abs=6e-3, model=tag.split("_")[0], route="float")


@pytest.mark.parametrize("key,h,w,ts", [("2x", 70, 75, 32), ("4x", 45, 50, 32), ("2x", 150, 140, 64), ("2x", 41, 20, 32)])
def test_tiled_frame_matches_oracle_tiling(nets, oracle_models, oracle, key, h, w, ts):
    """upscale_image/process_tile arithmetic (upscale_processing.py:395-519) with every tile as a
    plane of one launch; small tiles exercise all four border branches on both axes."""
    net, om = nets[key], oracle_models[key]
    img = oracle.synthetic_frame(h, w, kind="random", seed=ts + h)
    got = net.process_u8(img, tile_size=ts, border=10)
    want16 = om.upscale_image(img, tile_size=ts, border=10, flags=oracle.product_flags())
    want32 = om.upscale_image(img, tile_size=ts, border=10)
    check_u8(f"{key} upscale_image {w}x{h} t{ts}", got, want16, vs=PRODUCT, max_lsb=1, max_share=U8_DIFFER(oracle, key, "tiled"),
             model=key, route="tiled")
    check_u8(f"{key} upscale_image {w}x{h} t{ts}", got, want32, vs=FP32, model=key, route="tiled", **fp32_bar(key, "tiled"))


@pytest.mark.parametrize("key", ["2x", "4x"])
def test_sign_carry_between_launches_changes_no_bit(uva, nets, oracle, key, monkeypatch):
    """trunkw_kernel computes channels whose PReLU slope exceeds 1 negated; between two of its launches they stay negated in
    HBM and the next launch's weights take the sign back (uva_api.hip run_graph, `carry`).  Sign flips are exact: the frame must
    equal, byte for byte, the one a net built with UVA_TW_CARRY=0 (every launch restores the signs) gives -- tiled and whole,
    u8 and float route"""
    img = oracle.synthetic_frame(131, 94, seed=77)
    monkeypatch.setenv("UVA_TW_CARRY", "0")
    plain = load_net(uva, key)            # (the switches are read when a net's device side is built)
    a = plain.process_u8(img, tile_size=64, border=10)
    b = plain.process_u8(img, tile_size=0)
    monkeypatch.delenv("UVA_TW_CARRY")
    assert np.array_equal(nets[key].process_u8(img, tile_size=64, border=10), a)
    assert np.array_equal(nets[key].process_u8(img, tile_size=0), b)
    x = oracle.from_pixels_normalize(img)
    assert np.array_equal(nets[key]._extract(x), plain._extract(x))


@pytest.mark.parametrize("key", ["2x", "4x"])
def test_folded_last_strips_change_no_bit(uva, nets, oracle, key, monkeypatch):
    """trunkw_kernel walks the narrow last strips of two planes of one size in ONE pass (pairs 0..7: the first plane, 8..15: the
    second; csrc/uva_api.hip build_trunkw_schedule).  The arithmetic per pixel is untouched, so the frame must equal, BYTE FOR
    BYTE, the one a net with UVA_TW_FOLD=0 gives -- the bar that catches what the LSB / dB bars let through (the first version
    folded strips of 13 and 14 columns, whose last producer pair reads two raw columns of the other plane: one wrong column per
    plane, inside every tolerance).  Frames of 2T x 2T pixels with tile T give four planes of (T + 10)^2: last strips of 1, 5,
    11, 12 columns (folded), 13 and 14 (not), and the 1080p reference tiling (10 columns, 970 and 130 rows)."""
    monkeypatch.setenv("UVA_TW_FOLD", "0")
    plain = load_net(uva, key)            # (the schedule is built, and the switch read, when a geometry is first seen)
    frames = [(oracle.synthetic_frame(2 * t, 2 * t, kind="random", seed=900 + t), t) for t in (32, 51, 55, 61, 33, 34)]
    frames.append((oracle.synthetic_frame(1080, 1920, seed=77), 960))
    rng = np.random.default_rng(5)            # ... and a seeded sweep: 3 x 2, 2 x 3, 3 x 3 tile grids (interior tiles with borders on every side)
    for _ in range(int(os.environ.get("UVA_SWEEP_FACTOR", "1")) * 6):
        t = int(rng.integers(21, 70))
        ny, nx = int(rng.integers(2, 4)), int(rng.integers(2, 4))
        frames.append((oracle.synthetic_frame(ny * t - int(rng.integers(0, 9)), nx * t - int(rng.integers(0, 9)), kind="random", seed=t), t))
    want = [plain.process_u8(img, tile_size=t, border=10) for img, t in frames]
    monkeypatch.delenv("UVA_TW_FOLD")
    for (img, t), w in zip(frames, want):
        got = nets[key].process_u8(img, tile_size=t, border=10)
        assert np.array_equal(got, w), (key, t, int(np.abs(got.astype(int) - w.astype(int)).max()), float((got != w).mean()))


def test_structure_detector_catches_the_fold14_schedule(uva, nets, oracle, oracle_models, monkeypatch):
    """VERDICT r5 item 1b.  Round 5's first folded-strip schedule (last strips of up to 14 columns share a walk: the last
    producer pair then reads two raw columns of the OTHER plane) put one wrong column into every 74-wide plane and stayed inside
    <= 2 LSB / >= 50 dB against the fp32 oracle.  UVA_TW_FOLD=14 (debug opt-in) rebuilds exactly that schedule; every full-size
    comparison now also asks that no row or column of |HIP - oracle| stands out of its neighbourhood (parity_report.structure_u8),
    and that question must FAIL on the bad schedule and pass on the shipped one -- against the fp32 oracle, which knows nothing
    of the kernel's rounding."""
    import parity_report
    monkeypatch.setenv("UVA_TW_FOLD", "14")
    bad = load_net(uva, "2x")             # (the schedule is built, and the switch read, when a geometry is first seen)
    cases = [(oracle.synthetic_frame(128, 128, seed=31), 64), (oracle.synthetic_frame(384, 128, seed=32), 64),
             (oracle.synthetic_frame(128, 128, seed=33, kind="random"), 64)]
    bad_out = [bad.process_u8(img, tile_size=t, border=10) for img, t in cases]
    monkeypatch.delenv("UVA_TW_FOLD")
    for (img, t), b in zip(cases, bad_out):
        want = oracle_models["2x"].upscale_image(img, tile_size=t, border=10)
        good = nets["2x"].process_u8(img, tile_size=t, border=10)
        assert not np.array_equal(good, b), "UVA_TW_FOLD=14 did not change the frame: the known-bad schedule is not being built"
        st_good = check_u8(f"2x {img.shape[1]}x{img.shape[0]} t{t}: shipped schedule", good, want, vs=FP32, model="2x", route="tiled",
                           **fp32_bar("2x", "tiled"))
        d = np.abs(b.astype(np.int16) - want.astype(np.int16))
        st = parity_report.structure_u8(d)
        record(f"2x {img.shape[1]}x{img.shape[0]} t{t}: fold-14 schedule (KNOWN BAD, must be caught)", kind="u8", vs=FP32, model=None, route=None,
               samples=int(d.size), max_lsb=int(d.max()), psnr_db=parity_report.psnr_u8(b, want), differ_share=float((d > 0).mean()),
               bar_max_lsb=None, bar_min_psnr_db=None, bar_max_share=None, structure=st, bar_structure_z=parity_report.STRUCTURE_Z)
        assert st["col_z"] > 2 * parity_report.STRUCTURE_Z, ("the detector does not see the wrong columns", st)
        with pytest.raises(AssertionError, match="structured error"):
            check_u8("fold-14 through the bars (recorded again on purpose)", b, want, vs=FP32 + " [expected to fail]", model=None, route=None,
                     **fp32_bar("2x", "tiled"))


@pytest.mark.parametrize("key", ["2x", "4x"])
def test_winograd_trunk_against_the_direct_trunk_1080p(uva, nets, oracle, key, monkeypatch):
    """the oracle's product mode follows the kernel's rounding points, so the drift of the Winograd F(2,3) trunk
    (PReLU on halves) against the DIRECT fp16 trunk (trunk2_kernel, UVA_TRUNK_WINO=0: round 2/3's product) is pinned here on its own,
    on a whole 1080p frame with the reference tiling: both are within one level of the fp32 oracle, so at most two apart; the share
    of samples that differ and the PSNR between the two go into the parity report (expected below 1 % and around 69-70 dB: the
    fp16 roundings of the transformed operands), the bars are 2 % and 66 dB."""
    monkeypatch.setenv("UVA_TRUNK_WINO", "0")
    direct = load_net(uva, key)
    img = oracle.synthetic_frame(1080, 1920, seed=2026)
    a = direct.process_u8(img, tile_size=960, border=10)
    monkeypatch.delenv("UVA_TRUNK_WINO")
    b = nets[key].process_u8(img, tile_size=960, border=10)
    check_u8(f"{key} 1080p: Winograd F(2,3) trunk against the direct trunk", b, a, vs="trunk2_kernel (direct fp16 trunk, UVA_TRUNK_WINO=0)",
             max_lsb=2, min_psnr=66.0, max_share=0.02, model=key, route="tiled")


def test_fused_route_equals_float_route(nets, oracle):
    """The fused u8 device call against the reference-shaped float route (from_pixels ->
    normalize -> extract -> *255 -> convertTo) run tile by tile through the same kernels."""
    from upscale_video_amd import upscale_processing as up
    net = nets["2x"]
    img = oracle.synthetic_frame(70, 75, seed=3)
    fused = net.process_u8(img, tile_size=32, border=10)
    up.net = net
    canvas = np.zeros((140, 150, 3))
    items = []
    for y in range(3):
        for x in range(3):
            assert up.process_tile(img, 32, 2, y, x, 70, 75, canvas, items) == 0
    floaty = np.clip(np.rint(canvas), 0, 255).astype(np.uint8)
    d = np.abs(fused.astype(int) - floaty.astype(int))
    # the float route rounds x/255 to fp16 at the head, the u8 route feeds exact integers
    assert d.max() <= 1 and (d > 0).mean() <= 5e-2, (d.max(), (d > 0).mean())


def test_chain_1x_then_2x(nets, oracle_models, oracle):
    """BASELINE config 3: HurrDeblur 1x -> u8 -> 2x Compact."""
    img = oracle.synthetic_frame(48, 80, seed=11)
    mid = nets["1x"].process_u8(img, tile_size=0)
    out = nets["2x"].process_u8(mid, tile_size=960, border=10)
    omid = oracle_models["1x"].apply_model(img)
    want = oracle_models["2x"].upscale_image(omid)
    check_u8("chain: 1x stage 80x48", mid, omid, vs=FP32, model="1x", route="whole", **fp32_bar("1x", "whole"))
    check_u8("chain: 1x -> u8 -> 2x 80x48", out, want, vs=FP32 + " chain", model="chain", route="tiled", **fp32_bar("chain", "tiled"))


def test_row_strides_and_repeatability(nets, uva, oracle):
    from upscale_video_amd import _lib
    net = nets["2x"]
    h, w = 20, 36
    img = oracle.synthetic_frame(h, w, seed=5)
    ref = net.process_u8(img, tile_size=0)
    assert np.array_equal(ref, net.process_u8(img, tile_size=0))
    padded_in = np.zeros((h, w * 3 + 10), np.uint8)
    padded_in[:, :w * 3] = img.reshape(h, w * 3)
    out = np.full((h * 2, w * 6 + 7), 0xAB, np.uint8)
    _lib.check(_lib.load().uva_net_process_u8(net._h, padded_in.ctypes.data, h, w, padded_in.shape[1],
                                              out.ctypes.data, out.shape[1], 0, 0))
    assert np.array_equal(out[:, :w * 6].reshape(h * 2, w * 2, 3), ref)
    assert (out[:, w * 6:] == 0xAB).all()
    with pytest.raises(_lib.UvaError):
        _lib.check(_lib.load().uva_net_process_u8(net._h, padded_in.ctypes.data, h, w, w * 3 - 1,
                                                  out.ctypes.data, out.shape[1], 0, 0))


@pytest.mark.parametrize("key", ["2x", "4x", "1x"])
def test_full_size_frame_properties(nets, oracle_models, oracle, key):
    """BASELINE size (1920x1080), where the CPU oracle is too slow for a whole frame:
       (a) locality: a conv stack of n 3x3 layers has an n-px receptive radius, so any output
           window equals the oracle run on the window plus n px of context -- checked at the
           corners, a tile seam and the frame centre;
       (b) the reference tiling (960/10) differs from the whole-frame result only near seams;
       (c) determinism."""
    net, om = nets[key], oracle_models[key]
    h, w, s = 1080, 1920, net.scale
    rad = net.num_convs
    img = oracle.synthetic_frame(h, w, seed=20260928)
    whole = net.process_u8(img, tile_size=0)
    assert whole.shape == (h * s, w * s, 3)
    assert np.array_equal(whole, net.process_u8(img, tile_size=0))
    win = 24
    for (y0, x0) in [(0, 0), (0, w - win), (h - win, 0), (h - win, w - win), (h // 2, w // 2), (948, 948), (1056, 1880)]:
        cy0, cx0 = max(0, y0 - rad), max(0, x0 - rad)
        cy1, cx1 = min(h, y0 + win + rad), min(w, x0 + win + rad)
        crop = np.ascontiguousarray(img[cy0:cy1, cx0:cx1])
        want = om.apply_model(crop)[(y0 - cy0) * s:(y0 - cy0 + win) * s, (x0 - cx0) * s:(x0 - cx0 + win) * s]
        got = whole[y0 * s:(y0 + win) * s, x0 * s:(x0 + win) * s]
        check_u8(f"{key} 1080p whole frame, window ({y0},{x0})", np.ascontiguousarray(got), np.ascontiguousarray(want), vs=FP32,
                 model=key, route="whole", **fp32_bar(key, "whole"))
    if s > 1:
        tiled = net.process_u8(img, tile_size=960, border=10)
        d = np.abs(tiled.astype(int) - whole.astype(int))
        assert d.max() <= 2
        ys, xs = np.nonzero(d.max(axis=2))
        near_seam = (np.abs(ys - 960 * s) <= (rad + 2) * s) | (np.abs(xs - 960 * s) <= (rad + 2) * s)
        assert near_seam.all()
        assert psnr_u8(tiled, whole) >= 60


def test_huge_frame_is_split_into_several_trunk_launches(nets, oracle_models, oracle):
    """A frame with more 4-row work tiles than one launch's per-workgroup LDS schedule can hold
    (> 145 920 tiles, i.e. > 18.7 Mpx) goes through several trunk launches per layer (tile_base);
    checked by locality windows against the oracle, as in the full-size test."""
    net, om = nets["2x"], oracle_models["2x"]
    h, w, s = 4400, 4352, 2
    rad = net.num_convs
    rng = np.random.default_rng(99)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    out = net.process_u8(img, tile_size=0)
    assert out.shape == (h * s, w * s, 3)
    win = 16
    for (y0, x0) in [(0, 0), (h - win, w - win), (4292, 2000), (2200, 4330), (4000, 100)]:
        cy0, cx0 = max(0, y0 - rad), max(0, x0 - rad)
        cy1, cx1 = min(h, y0 + win + rad), min(w, x0 + win + rad)
        want = om.apply_model(np.ascontiguousarray(img[cy0:cy1, cx0:cx1]))[
            (y0 - cy0) * s:(y0 - cy0 + win) * s, (x0 - cx0) * s:(x0 - cx0 + win) * s]
        got = out[y0 * s:(y0 + win) * s, x0 * s:(x0 + win) * s]
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 2 and psnr_u8(got, want) >= 48, ((y0, x0), d.max(), psnr_u8(got, want))


def test_1x_frame_too_large_for_the_fused_kernel_takes_the_pair_path(nets, oracle_models, oracle):
    """sub10_kernel (the whole 1x net in one launch) copies its row list into LDS: 640 rows per workgroup, which a
    2700 x 3840 frame exceeds (64 strips x 2700 rows / 256 workgroups).  The call must fall back to the per-pair
    kernels, not fail -- checked by locality windows against the oracle; a full-size 2160 x 3840 frame, which still
    fits, is checked the same way."""
    net, om = nets["1x"], oracle_models["1x"]
    rad = net.num_convs
    rng = np.random.default_rng(123)
    for h, w in ((2700, 3840), (2160, 3840)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out = net.process_u8(img, tile_size=0)
        assert out.shape == (h, w, 3)
        win = 24
        for (y0, x0) in [(0, 0), (h - win, w - win), (h // 2, 59), (1077, w - 61), (h - win, 1000)]:
            cy0, cx0 = max(0, y0 - rad), max(0, x0 - rad)
            cy1, cx1 = min(h, y0 + win + rad), min(w, x0 + win + rad)
            want = om.apply_model(np.ascontiguousarray(img[cy0:cy1, cx0:cx1]))[y0 - cy0:y0 - cy0 + win, x0 - cx0:x0 - cx0 + win]
            got = out[y0:y0 + win, x0:x0 + win]
            d = np.abs(got.astype(int) - want.astype(int))
            assert d.max() <= 1 and psnr_u8(got, want) >= 55, ((h, w), (y0, x0), d.max(), psnr_u8(got, want))


def test_device_chain_stays_on_gpu(nets, oracle_models, oracle):
    """Config 3 without the PNG hop: 1x net -> u8 frame in HBM -> 2x net, ordered on the device with
    uva_net_wait_for; identical to the host-route chain (the u8 hop carries the same information)."""
    if torch is None or not torch.cuda.is_available():
        pytest.skip("torch cannot see the GPU in this process")
    img = oracle.synthetic_frame(48, 80, seed=21)
    h, w = img.shape[:2]
    d_in = torch.from_numpy(img).cuda()
    d_mid = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    d_out = torch.empty((h * 2, w * 2, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    pre, net = nets["1x"], nets["2x"]
    for _ in range(3):
        pre.wait_for(net)
        pre.process_u8_device(d_in.data_ptr(), h, w, d_mid.data_ptr(), tile_size=0)
        net.wait_for(pre)
        net.process_u8_device(d_mid.data_ptr(), h, w, d_out.data_ptr(), tile_size=960, border=10)
    pre.synchronize()
    net.synchronize()
    host = net.process_u8(pre.process_u8(img, tile_size=0), tile_size=960, border=10)
    assert np.array_equal(d_out.cpu().numpy(), host)
    want = oracle_models["2x"].upscale_image(oracle_models["1x"].apply_model(img))
    assert np.abs(host.astype(int) - want.astype(int)).max() <= 3 and psnr_u8(host, want) >= 48


def test_pipelined_host_route_equals_synchronous(nets, uva, oracle):
    """uva_net_submit_u8 / uva_net_collect_u8 (3 frames in flight, pinned and pageable buffers, different
    frame sizes interleaved, out-of-order collection) give exactly what uva_net_process_u8 gives."""
    net = nets["2x"]
    frames = [oracle.synthetic_frame(72 + 8 * i, 96 + 16 * (i % 3), seed=100 + i) for i in range(7)]
    want = [net.process_u8(f, tile_size=32, border=10) for f in frames]
    pinned = []
    for f in frames[:3]:
        p = uva.pinned_empty(f.shape)
        p[...] = f
        pinned.append(p)
    tickets = []
    got = {}
    for i, f in enumerate(frames):
        if len(tickets) == 3:
            j, t = tickets.pop(1 if len(tickets) > 1 else 0)   # not the oldest: out-of-order collection
            got[j] = net.collect_u8(t).copy()
        src = pinned[i] if i < 3 else f
        out = uva.pinned_empty((f.shape[0] * 2, f.shape[1] * 2, 3)) if i % 2 == 0 else None
        tickets.append((i, net.submit_u8(src, out=out, tile_size=32, border=10)))
    for j, t in tickets:
        got[j] = net.collect_u8(t).copy()
    for i in range(len(frames)):
        assert np.array_equal(got[i], want[i]), i


def test_pipeline_rejects_a_fourth_frame_in_flight(nets, uva, oracle):
    net = nets["1x"]
    f = oracle.synthetic_frame(40, 48)
    ts = [net.submit_u8(f) for _ in range(3)]
    with pytest.raises(Exception, match="in flight"):
        net.submit_u8(f)
    outs = [net.collect_u8(t) for t in ts]
    assert all(np.array_equal(o, outs[0]) for o in outs)
    with pytest.raises(Exception, match="not in flight"):
        net.collect_u8(ts[0])
    net.synchronize()


def test_workspace_cache_eviction_and_reuse(nets, oracle):
    """More frame geometries than the per-net workspace cache holds (6), revisited in a different
    order: results must not depend on what was evicted and rebuilt in between, nor on
    destroy_gpu_instance() having dropped every device allocation."""
    from upscale_video_amd import ncnn
    net = nets["2x"]
    shapes = [(40 + 4 * i, 64 + 16 * (i % 4)) for i in range(9)]
    frames = [oracle.synthetic_frame(h, w, seed=300 + i) for i, (h, w) in enumerate(shapes)]
    first = [net.process_u8(f, tile_size=32, border=10) for f in frames]
    for i in (8, 0, 4, 1, 7, 2):
        assert np.array_equal(net.process_u8(frames[i], tile_size=32, border=10), first[i]), i
    ncnn.destroy_gpu_instance()              # upscale_processing.py:292: nets stay usable afterwards
    for i in (3, 0, 8):
        assert np.array_equal(net.process_u8(frames[i], tile_size=32, border=10), first[i]), i
    # the other nets of this module lost their device state too; touch them so later tests start clean
    for k in ("4x", "1x"):
        nets[k].process_u8(frames[0])


@pytest.mark.parametrize("key", ["2x", "4x", "1x"])
def test_random_geometries_match_oracle(nets, oracle_models, oracle, key):
    """Seeded sweep over frame sizes, tile sizes and row strides (partial tiles on both axes, planes
    narrower than a 16-pixel MFMA column block, 1-row and 1-column frames, tile seams at odd places):
    the fused u8 route against the oracle's upscale_image / apply_model."""
    import ctypes
    net, om = nets[key], oracle_models[key]
    s = net.scale
    rng = np.random.default_rng({"2x": 11, "4x": 12, "1x": 13}[key])
    n_cases = (14 if key == "2x" else 7) * int(os.environ.get("UVA_SWEEP_FACTOR", "1"))   # (a longer sweep on demand: profiles/r02_l_parity_sweep.txt)
    for case in range(n_cases):
        h = int(rng.integers(1, 90)); w = int(rng.integers(1, 150))
        if case == 0: h, w = 1, 17
        if case == 1: h, w = 5, 1
        ts = int(rng.choice([0, 24, 32, 48, 64]))
        img = oracle.synthetic_frame(h, w, kind="random" if case % 2 else "smooth", seed=1000 * case + h)
        want = om.upscale_image(img, tile_size=ts, border=10) if ts else om.apply_model(img)
        # padded row strides on both sides, through the raw C entry point
        in_stride, out_stride = w * 3 + int(rng.integers(0, 9)), w * s * 3 + int(rng.integers(0, 9))
        src = np.zeros((h, in_stride), np.uint8); src[:, :w * 3] = img.reshape(h, w * 3)
        dst = np.full((h * s, out_stride), 0xA5, np.uint8)
        rc = net._L.uva_net_process_u8(net._h, src.ctypes.data, h, w, ctypes.c_size_t(in_stride), dst.ctypes.data,
                                       ctypes.c_size_t(out_stride), ts, 10 if ts else 0)
        assert rc == 0, net._L.uva_last_error()
        got = dst[:, :w * s * 3].reshape(h * s, w * s, 3)
        assert (dst[:, w * s * 3:] == 0xA5).all(), (key, case, "row padding was written")
        check_u8(f"{key} sweep case {case}: {w}x{h} t{ts}", np.ascontiguousarray(got), want, vs=FP32,
                 model=key, route="tiled" if ts else "whole", **fp32_bar(key, "tiled" if ts else "whole"))


@pytest.mark.parametrize("key", ["2x", "1x"])
def test_persistent_head_changes_no_bit(nets, oracle, key, monkeypatch):
    """headp_kernel (persistent, software-pipelined; the default) against head_kernel (one workgroup per tile): same bytes, on a
    frame with partial tiles on both axes, whole and tiled (the 1x net's tiled route runs the 24-feature instantiation)."""
    net = nets[key]
    img = oracle.synthetic_frame(203, 331, kind="random", seed=77)
    for ts in (0, 64):
        monkeypatch.setenv("UVA_HEAD_PERSIST", "1")
        a = net.process_u8(img, tile_size=ts, border=10 if ts else 0)
        monkeypatch.setenv("UVA_HEAD_PERSIST", "0")
        b = net.process_u8(img, tile_size=ts, border=10 if ts else 0)
        assert np.array_equal(a, b), (key, ts)


def test_frames_of_4_gb_and_more_are_refused(nets):
    """the tail kernels address residual and output bytes with 32-bit offsets from the frame's base: the API says no up front"""
    import ctypes
    import torch
    net = nets["2x"]
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    rc = net._L.uva_net_process_u8_device(net._h, ctypes.c_void_p(buf.data_ptr()), 8, 16, ctypes.c_size_t(1 << 29),
                                          ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(16 * 2 * 3), 0, 0)
    assert rc != 0 and b"4 GB" in net._L.uva_last_error()


@pytest.mark.parametrize("key", ["2x", "4x"])
def test_whole_1080p_frame_against_the_oracle(nets, oracle_models, oracle, key):
    """BASELINE config 2's frame (and config 4's runnable stand-in, 4x Compact), every output sample: the
    fused u8 route with the reference's 960/10 tiling against the oracle's upscale_image on the host
    cores (~15 s each on the GPU box)."""
    if oracle.cpu_quota() < 12:
        pytest.skip("the fp32 oracle needs a many-core host for a whole 1080p frame")
    s = nets[key].scale
    img = oracle.synthetic_frame(1080, 1920, seed=2026)
    got = nets[key].process_u8(img, tile_size=960, border=10)
    want = oracle_models[key].upscale_image(img, tile_size=960, border=10)
    assert got.shape == (1080 * s, 1920 * s, 3)
    # bars: the measured distance of this frame (2x: 71.3 dB, 0.48 % of the samples one level apart; 4x: 69.9 dB, 0.66 %) less a
    # margin, and no row or column standing out of its neighbourhood (parity_report.structure_u8)
    worst, psnr, share = check_u8(f"{key} WHOLE 1080p frame, reference tiling 960/10, every sample", got, want, vs=FP32, max_lsb=1,
                                  min_psnr={"2x": 69.3, "4x": 67.9}[key], max_share=0.02, model=key, route="tiled")
    print(f"whole 1080p frame, {key}: max |diff| {worst} LSB, PSNR {psnr:.2f} dB, {100 * share:.3f} % of the samples differ")
