"""Generic ncnn graphs (`-m r`, models/4x_Valar_v1.param): host loader on CPU, the HIP
executor against the numpy restatement (oracle/generic_oracle.py) under -m gpu.  The Valar weights are a
missing blob upstream, so every comparison here uses synthetic weights: functional coverage, no parity claim."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, load_net, model_paths, psnr_u8
from upscale_video_amd import _lib

VALAR = os.path.join(ROOT, "models", "4x_Valar_v1.param")

MINI = """7767517
19 24
Input            input      0 1 input
Convolution      c0         1 1 input a 0=64 1=3 4=1 5=1 6=1728
Split            s0         1 5 a a0 a1 a2 a3 a4
Convolution      c1         1 1 a4 b 0=32 1=3 4=1 5=1 6=18432 9=2 -23310=1,2.000000e-01
Split            s1         1 2 b b0 b1
Concat           cat1       2 1 a3 b1 c
Convolution      c2         1 1 c d 0=32 1=3 4=1 5=1 6=27648 9=2 -23310=1,2.000000e-01
Convolution      c3         1 1 a2 e 0=32 1=1 6=2048
BinaryOp         add1       2 1 d e f
Concat           cat2       3 1 a1 b0 f g
Convolution      c4         1 1 g h 0=64 1=3 4=1 5=1 6=73728
Eltwise          sum1       2 1 h a0 i 0=1 -23301=2,2.000000e-01,1.000000e+00
Interp           up1        1 1 i j 0=1 1=2.000000e+00 2=2.000000e+00
Convolution      c5         1 1 j k 0=64 1=3 4=1 5=1 6=36864 9=2 -23310=1,2.000000e-01
Interp           up2        1 1 k l 0=1 1=2.000000e+00 2=2.000000e+00
Convolution      c6         1 1 l m 0=48 1=3 4=1 5=1 6=27648
PReLU            p6         1 1 m n 0=48
Convolution      c7         1 1 n o 0=64 1=3 4=1 5=1 6=27648 9=2 -23310=1,2.000000e-01
Convolution      c8         1 1 o output 0=3 1=3 4=1 5=1 6=1728
"""


@pytest.fixture(scope="module")
def mini(tmp_path_factory):
    from oracle import generic_oracle as go
    d = tmp_path_factory.mktemp("mini")
    p, b = str(d / "4x_mini.param"), str(d / "4x_mini.bin")
    open(p, "w").write(MINI)
    go.write_synthetic_bin(p, b, seed=3)
    return p, b, go.Model(p, b)


def test_numpy_restatement_agrees_with_the_c_oracle_on_the_real_compact_weights(oracle, oracle_models):
    """The generic numpy oracle is itself checked where a second implementation with REAL weights exists."""
    from oracle import generic_oracle as go
    for key in ("2x", "1x"):
        p, b = model_paths(key)
        m = go.Model(p, b)
        img = oracle.synthetic_frame(13, 17, kind="random", seed=4)
        x = oracle.from_pixels_normalize(img)
        want = oracle_models[key].forward(x)
        got = m.forward(x)
        assert got.shape == want.shape and float(np.abs(got - want).max()) <= 2e-5
        d = np.abs(m.apply_u8(img).astype(int) - oracle_models[key].apply_model(img).astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3          # exact .5 ties may fall either way


def test_executor_plan_for_valar_and_the_mini_graph(uva, mini):
    """The plan the executor derives from the graph alone (csrc/uva_generic.h plan_concat_groups): Valar's 23 RRDBs x 3
    dense blocks become 69 chains of four Concats each in arrays of 64 + 4 x 32 channels -- all free, the block's input x
    being written into the array by the sum (or the convolution with the sum in its epilogue) that produces it, the
    very first block's by the 3-channel head convolution -- and all but the 1x1 convolutions take the LDS-tiled kernel."""
    import ctypes
    from upscale_video_amd import _lib
    L = _lib.load()
    net = uva.Net()
    assert net.load_param(VALAR) == 0, getattr(net, "last_error", "")
    info = (ctypes.c_int * 8)()
    assert L.uva_net_debug_generic_plan(net._h, info) == 0, L.uva_last_error()
    groups, concats, free, first, lds_convs, widest = list(info)[:6]
    assert (groups, concats, free, first, widest) == (69, 276, 276, 0, 192)
    assert lds_convs == 420 - 69                         # the one 1x1 convolution of every dense block stays on the plain kernel
    p, b, _ = mini
    net2 = uva.Net()
    assert net2.load_param(p) == 0 and net2.load_model(b) == 0
    assert L.uva_net_debug_generic_plan(net2._h, info) == 0
    assert info[0] >= 1 and info[2] + info[3] == info[1]         # the mini RRDB graph exercises the shared arrays too
    compact = uva.Net()
    assert compact.load_param(model_paths("2x")[0]) == 0
    assert L.uva_net_debug_generic_plan(compact._h, info) != 0   # an SRVGGNetCompact graph never reaches this executor


def test_loader_accepts_valar_and_checks_the_weight_stream(uva, tmp_path, mini):
    from oracle import generic_oracle as go
    net = uva.Net()
    assert net.load_param(VALAR) == 0, getattr(net, "last_error", "")
    assert (net.scale, net.num_convs, net.num_features) == (4, 420, 192)     # 64 + 4 x 32 channels at the widest Concat
    assert len(go.conv_shapes(go.parse_param(VALAR))) == 420
    # the reference's .bin is a missing blob: a file of another graph must not fit
    assert net.load_model(model_paths("4x")[1]) != 0
    p, b, _ = mini
    net2 = uva.Net()
    assert net2.load_param(p) == 0 and net2.load_model(b) == 0, getattr(net2, "last_error", "")
    assert (net2.scale, net2.num_convs) == (4, 9)
    raw = open(b, "rb").read()
    (tmp_path / "short.bin").write_bytes(raw[:-3])
    (tmp_path / "long.bin").write_bytes(raw + b"\0\0\0\0")
    assert net2.load_model(str(tmp_path / "short.bin")) != 0
    assert net2.load_model(str(tmp_path / "long.bin")) != 0 and "unread bytes" in net2.last_error
    # layer types outside the executor's list are still refused by name
    bad = MINI.replace("Eltwise          sum1       2 1 h a0 i 0=1 -23301=2,2.000000e-01,1.000000e+00", "Softmax          sum1       1 1 h i")
    (tmp_path / "bad.param").write_text(bad)
    n3 = uva.Net()
    assert n3.load_param(str(tmp_path / "bad.param")) != 0 and "unsupported layer type 'Softmax'" in n3.last_error


def _close(got, want, rel):
    scale = float(np.abs(want).max())
    return float(np.abs(got - want).max()) <= rel * scale + 1e-3, (float(np.abs(got - want).max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", [(9, 14), (16, 16), (21, 37), (1, 1)])
def test_mini_graph_matches_the_restatement(uva, mini, oracle, h, w):
    p, b, om = mini
    net = uva.Net()
    net.set_vulkan_device(0)
    assert net.load_param(p) == 0 and net.load_model(b) == 0
    img = oracle.synthetic_frame(h, w, kind="random", seed=h * 100 + w)
    x = oracle.from_pixels_normalize(img)
    got = net._extract(x)
    want16, want32 = om.forward(x, f16_storage=True), om.forward(x)
    assert got.shape == (3, 4 * h, 4 * w)
    ok, info = _close(got, want16, 4e-3)
    assert ok, info
    ok, info = _close(got, want32, 2e-2)
    assert ok, info
    u8 = net.process_u8(img, tile_size=0)
    ref = om.apply_u8(img, f16_storage=True)
    d = np.abs(u8.astype(int) - ref.astype(int))
    assert d.max() <= 2 and (d > 0).mean() < 0.1, (int(d.max()), float((d > 0).mean()))


@pytest.mark.gpu
def test_mini_graph_with_reference_tiling(uva, mini, oracle):
    """upscale_image's tile loop (upscale_processing.py:499-516) on the generic path: every tile a plane, core pasted."""
    from upscale_video_amd import upscale_processing as up
    p, b, om = mini
    net = uva.Net()
    net.set_vulkan_device(0)
    assert net.load_param(p) == 0 and net.load_model(b) == 0
    h, w, ts = 40, 53, 16
    img = oracle.synthetic_frame(h, w, seed=77)
    got = net.process_u8(img, tile_size=ts, border=10)
    want = np.zeros((4 * h, 4 * w, 3), np.uint8)
    for ty in range(-(-h // ts)):
        for tx in range(-(-w // ts)):
            (y0, y1, x0, x1), (top, bottom, left, right) = up.tile_window(ts, ty, tx, h, w)
            tile = om.apply_u8(np.ascontiguousarray(img[y0 - top:y1 + bottom, x0 - left:x1 + right]), f16_storage=True)
            want[4 * y0:4 * y1, 4 * x0:4 * x1] = tile[4 * top:4 * (top + y1 - y0), 4 * left:4 * (left + x1 - x0)]
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 2 and (d > 0).mean() < 0.1, (int(d.max()), float((d > 0).mean()))


def _valar_fixture(tmp_path):
    """tests/golden/valar_synthetic.npz (oracle/independent_check.py: torch ops, own parser) and the weights it was made with"""
    from upscale_video_amd import synth
    g = np.load(os.path.join(ROOT, "tests", "golden", "valar_synthetic.npz"))
    seed, gain = g["valar_seed_gain"]
    b = str(tmp_path / "4x_Valar_v1.bin")
    synth.synthetic_weights(VALAR, b, seed=int(seed), gain=float(gain))
    return g, b


def test_numpy_restatement_agrees_with_the_independent_evaluation_of_valar(tmp_path):
    """the ops only 4x_Valar_v1 has (Concat, Eltwise with coefficients, LeakyReLU fused into a convolution,
    the bias-less 1x1 convolution, nearest x2 Interp) were restated exactly once; the committed fixture is a second,
    independent evaluation (torch.nn.functional, its own .param / .bin parser) of all 1206 layers."""
    from oracle import generic_oracle as go
    g, b = _valar_fixture(tmp_path)
    om = go.Model(VALAR, b)
    img = g["valar_12x20_in"]
    x = img.transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255.0)
    f = om.forward(x)
    want = g["valar_12x20_f32"]
    assert f.shape == want.shape == (3, 48, 80)
    assert float(np.abs(f - want).max()) <= 2e-5 * float(np.abs(want).max()) + 1e-6
    u = om.apply_u8(img)
    assert np.abs(u.astype(int) - g["valar_12x20_u8"].astype(int)).max() <= 1 and (u != g["valar_12x20_u8"]).mean() <= 1e-3


@pytest.mark.gpu
def test_valar_graph_against_the_independent_fixture(uva, tmp_path):
    """The HIP executor (rdb4_kernel / g_conv3_sw on the 75-wide plane, the layer-by-layer kernels on the 20-wide one) against
    the independent torch evaluation: f32 within 4e-3 of max|out| (fp16 storage against fp32), u8 within 2 LSB."""
    g, b = _valar_fixture(tmp_path)
    net = uva.Net()
    net.set_vulkan_device(0)
    assert net.load_param(VALAR) == 0 and net.load_model(b) == 0, getattr(net, "last_error", "")
    for tag in ("valar_12x20", "valar_70x75"):
        img, want = g[tag + "_in"], g[tag + "_f32"]
        got = net._extract(img.transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255.0))
        ok, info = _close(got, want, 4e-3 if tag == "valar_70x75" else 2e-2)
        assert ok, (tag, info)
        u8 = net.process_u8(img, tile_size=0)
        assert np.abs(u8.astype(int) - g[tag + "_u8"].astype(int)).max() <= 2 and psnr_u8(u8, g[tag + "_u8"]) >= 45, tag


@pytest.mark.gpu
def test_valar_graph_with_synthetic_weights(uva, oracle, tmp_path):
    """All 1206 layers of 4x_Valar_v1 (23 RRDBs: Concat up to 192 channels, 1x1 convolutions, Eltwise(0.2, 1.0),
    two nearest x2 Interps) on a small frame, synthetic weights, against the numpy restatement."""
    from oracle import generic_oracle as go
    b = str(tmp_path / "4x_Valar_v1.bin")
    go.write_synthetic_bin(VALAR, b, seed=11, gain=0.5)
    om = go.Model(VALAR, b)
    net = uva.Net()
    net.set_vulkan_device(0)
    assert net.load_param(VALAR) == 0 and net.load_model(b) == 0, getattr(net, "last_error", "")
    img = oracle.synthetic_frame(12, 20, seed=5)
    x = oracle.from_pixels_normalize(img)
    got = net._extract(x)
    want = om.forward(x, f16_storage=True)
    assert got.shape == (3, 48, 80) and np.isfinite(got).all()
    ok, info = _close(got, want, 2e-2)
    assert ok, info
    u8 = net.process_u8(img, tile_size=960, border=10)
    assert u8.shape == (48, 80, 3)
    # the worker layer reaches it by name like the reference does (model_file = "x_Valar_v1", :913-916)
    from upscale_video_amd import upscale_processing as up
    os.symlink(VALAR, tmp_path / "4x_Valar_v1.param")
    up.init_worker([0], 0, str(tmp_path), "x_Valar_v1", 4, "input", "output")
    assert up.net is not None, up.init_error
    assert np.array_equal(up.net.process_u8(img, tile_size=960, border=10), u8)


@pytest.mark.gpu
def test_valar_graph_multi_tile_frame_against_the_restatement(uva, oracle, tmp_path):
    """The as-named graph at the bar of the mini graph: all 1206 layers on a frame that takes the reference's tile loop
    (32-pixel tiles, 10-pixel borders: nine planes, 42..52 pixels wide -- rdb4_kernel for the dense blocks' first four
    convolutions, g_conv3_sw for the 192 -> 64 and 64 -> 64 ones) and on one 75-pixel-wide plane through the float route,
    synthetic weights, against the numpy restatement in fp16-storage mode: f32 within 4e-3 of max|out|, u8 within 2 LSB."""
    from oracle import generic_oracle as go
    from upscale_video_amd import upscale_processing as up
    b = str(tmp_path / "4x_Valar_v1.bin")
    go.write_synthetic_bin(VALAR, b, seed=11, gain=0.5)
    om = go.Model(VALAR, b)
    net = uva.Net()
    net.set_vulkan_device(0)
    assert net.load_param(VALAR) == 0 and net.load_model(b) == 0, getattr(net, "last_error", "")
    x = oracle.from_pixels_normalize(oracle.synthetic_frame(34, 75, seed=21))
    got = net._extract(x)
    want = om.forward(x, f16_storage=True)
    ok, info = _close(got, want, 4e-3)
    assert ok, info
    h, w, ts = 70, 75, 32
    img = oracle.synthetic_frame(h, w, seed=9)
    u8 = net.process_u8(img, tile_size=ts, border=10)
    ref = np.zeros((4 * h, 4 * w, 3), np.uint8)
    for ty in range(-(-h // ts)):
        for tx in range(-(-w // ts)):
            (y0, y1, x0, x1), (top, bottom, left, right) = up.tile_window(ts, ty, tx, h, w)
            tile = om.apply_u8(np.ascontiguousarray(img[y0 - top:y1 + bottom, x0 - left:x1 + right]), f16_storage=True)
            ref[4 * y0:4 * y1, 4 * x0:4 * x1] = tile[4 * top:4 * (top + y1 - y0), 4 * left:4 * (left + x1 - x0)]
    d = np.abs(u8.astype(int) - ref.astype(int))
    assert d.max() <= 2 and (d > 0).mean() < 0.1, (int(d.max()), float((d > 0).mean()))


_FUSE_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from upscale_video_amd import ncnn
from upscale_video_amd.synth import synthetic_frame
net = ncnn.Net(); net.set_vulkan_device(0)
assert net.load_param(sys.argv[2]) == 0 and net.load_model(sys.argv[3]) == 0
outs = [net.process_u8(synthetic_frame(h, w, seed=h + w), tile_size=t, border=10) for h, w, t in ((12, 20, 0), (37, 45, 0), (50, 70, 32), (40, 150, 64))]
np.savez(sys.argv[4], *outs)
"""


@pytest.mark.gpu
def test_sums_done_in_the_convolution_epilogue_change_nothing(tmp_path):
    """Valar's Eltwise / BinaryOp sums that follow a convolution are done while its result leaves the kernel
    (GConvArgs::res); with UVA_GENERIC_FUSE_ADD=0 every sum is a launch of its own.  Same bytes, also with the reference
    tiling and ragged tiles."""
    import subprocess
    import sys
    from oracle import generic_oracle as go
    b = str(tmp_path / "4x_Valar_v1.bin")
    go.write_synthetic_bin(VALAR, b, seed=7, gain=0.5)
    res = []
    for v in ("1", "0"):
        f = str(tmp_path / ("o%s.npz" % v))
        subprocess.check_call([sys.executable, "-c", _FUSE_CHILD, ROOT, VALAR, b, f], env=dict(os.environ, UVA_GENERIC_FUSE_ADD=v))
        res.append(np.load(f))
    for k in res[0].files:
        assert np.array_equal(res[0][k], res[1][k]), k
        assert res[0][k].std() > 0


_GRID_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "gpu", os.path.join(sys.argv[1], "tests", "test_generic_graph.py"), "-k", "multi_tile_frame_against or graph_with_synthetic_weights", "-p", "no:cacheprovider"]))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("sk", ["0", "1", "direct"])
def test_long_segments_against_the_restatement(sk):
    """The persistent kernels' loops over the blocks / rows of a segment, and several segments per workgroup: with one
    workgroup per CU a frame small enough for the numpy restatement gives every workgroup at most one block, so the two
    parity tests are run again on EIGHT workgroups (UVA_GENERIC_GRID=8: 6-7 four-row blocks and 17 rows per workgroup),
    with g_conv3_sww (the default since round 6: Winograd F(2,3)), with g_conv3_sw<6, 1> (UVA_GENERIC_WINO=0, "direct") and with the
    k-split 32x32x16 kernel (UVA_GENERIC_SK=1) for the 192 -> 64 convolutions."""
    import subprocess
    import sys
    if os.environ.get("UVA_GENERIC_GRID"):
        pytest.skip("already the child")
    env = dict(os.environ, UVA_GENERIC_GRID="8", UVA_GENERIC_SK="1" if sk == "1" else "0", UVA_GENERIC_WINO="0" if sk == "direct" else "1")
    r = subprocess.run([sys.executable, "-c", _GRID_CHILD, ROOT], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "2 passed" in r.stdout, r.stdout[-500:]


@pytest.mark.gpu
def test_winograd_conv5_against_the_direct_kernel(tmp_path):
    """Round 6: a dense block's last convolution (192 -> 64) runs as 1-D Winograd F(2,3) (g_conv3_sww, csrc/uva_sww.hip.h);
    UVA_GENERIC_WINO=0 keeps g_conv3_sw<6, 1>.  In exact arithmetic the same convolution; what moves is where fp16 rounds (the
    transformed inputs and weights are the MFMA operands) -- 69 such layers deep the two must stay within ONE u8 level of each
    other and well above 50 dB, whole frames and the reference tiling with ragged tiles (the planes narrower than a strip take
    the layer-by-layer kernel either way: identical there)."""
    import subprocess
    import sys
    from oracle import generic_oracle as go
    b = str(tmp_path / "4x_Valar_v1.bin")
    go.write_synthetic_bin(VALAR, b, seed=7, gain=0.5)
    res = []
    for v in ("1", "0"):
        f = str(tmp_path / ("o%s.npz" % v))
        subprocess.check_call([sys.executable, "-c", _FUSE_CHILD, ROOT, VALAR, b, f], env=dict(os.environ, UVA_GENERIC_WINO=v))
        res.append(np.load(f))
    some_differ = False
    for k in res[0].files:
        a, d = res[0][k], res[1][k]
        diff = np.abs(a.astype(int) - d.astype(int))
        assert diff.max() <= 1 and psnr_u8(a, d) >= 55, (k, int(diff.max()), psnr_u8(a, d))
        some_differ |= bool(diff.max() > 0)
        assert a.std() > 0
    assert some_differ, "UVA_GENERIC_WINO changed nothing: is g_conv3_sww the kernel that ran?"


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["UVA_GENERIC_BATCH", "UVA_GENERIC_FUSE_INTERP", "UVA_GENERIC_FUSE_OUT"])
def test_plane_batches_and_the_folded_interp_change_nothing(tmp_path, switch):
    """A frame's reference tiles go through the graph together (one rdb4 / g_conv3_sw launch per layer for all planes,
    UVA_GENERIC_BATCH=0: one plane after the other) and the nearest 2x Interp is folded into the next convolution's row
    DMA (UVA_GENERIC_FUSE_INTERP=0: a launch of its own): the same bytes either way, ragged tiles included."""
    import subprocess
    import sys
    from oracle import generic_oracle as go
    b = str(tmp_path / "4x_Valar_v1.bin")
    go.write_synthetic_bin(VALAR, b, seed=11, gain=0.5)
    res = []
    for v in ("1", "0"):
        f = str(tmp_path / ("o%s.npz" % v))
        subprocess.check_call([sys.executable, "-c", _FUSE_CHILD, ROOT, VALAR, b, f], env=dict(os.environ, **{switch: v}))
        res.append(np.load(f))
    for k in res[0].files:
        assert np.array_equal(res[0][k], res[1][k]), k
        assert res[0][k].std() > 0


@pytest.mark.gpu
def test_compact_graphs_through_the_generic_executor(uva, oracle, oracle_models, monkeypatch):
    """Cross-check with REAL weights: the generic executor (UVA_GENERIC=1) on the 2x / 1x Compact graphs against
    the C oracle and against the fused kernels' result."""
    fused = {k: load_net(uva, k) for k in ("2x", "1x")}
    monkeypatch.setenv("UVA_GENERIC", "1")
    for key in ("2x", "1x"):
        net = load_net(uva, key)
        img = oracle.synthetic_frame(30, 41, kind="random", seed=8)
        got = net.process_u8(img, tile_size=0)
        want = oracle_models[key].apply_model(img)
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 2 and psnr_u8(got, want) >= 50, (key, int(d.max()), psnr_u8(got, want))
        d2 = np.abs(got.astype(int) - fused[key].process_u8(img, tile_size=0).astype(int))
        assert d2.max() <= 1


def _segments(kind, h, w, grid=256):
    L = _lib.load()
    need = ctypes.c_size_t(0)
    L.uva_debug_generic_segments(kind, h, w, grid, None, 0, ctypes.byref(need), None)
    words = (ctypes.c_int32 * need.value)()
    sbeg = (ctypes.c_int * (grid + 1))()
    assert L.uva_debug_generic_segments(kind, h, w, grid, words, need.value, ctypes.byref(need), sbeg) == 0, L.uva_last_error()
    segs = np.frombuffer(words, np.int32).reshape(-1, 8)
    return segs, list(sbeg)


@pytest.mark.parametrize("h,w", [(970, 970), (130, 970), (1080, 1920), (52, 42), (7, 45), (20, 12), (333, 97), (2160, 3840)])
def test_rdb4_work_list_covers_every_pixel_once(h, w):
    """rdb4_kernel's segments (a dense block's first four convolutions in one launch): every plane pixel owned by exactly
    one segment; a segment's own columns lie where its 48 computed columns are still right after four convolutions (three
    columns in from a cut edge, up to the plane's edge otherwise); at most one segment per workgroup unless there are
    more strips than workgroups."""
    segs, sbeg = _segments(0, h, w)
    cover = np.zeros((h, w), np.int32)
    for c0, yb, ye, own0, own1 in segs[:, :5]:
        assert 0 <= yb < ye <= h and 0 <= own0 < own1 <= w and c0 >= 0
        assert own0 >= (c0 + 3 if c0 > 0 else 0) and own1 <= (c0 + 45 if c0 + 48 < w else w)
        cover[yb:ye, own0:own1] += 1
    assert (cover == 1).all()
    assert sbeg[0] == 0 and sbeg[-1] == len(segs) and all(b >= a for a, b in zip(sbeg, sbeg[1:]))
    per_wg = np.diff(sbeg)
    nstrips = len({int(c) for c in segs[:, 0]})
    assert per_wg.max() == (1 if nstrips <= 256 else -(-nstrips // 256))
    rows = np.array([sum(int(s[2] - s[1]) for s in segs[sbeg[g]:sbeg[g + 1]]) for g in range(256)])
    busy = rows[rows > 0]
    assert busy.max() - busy.min() <= 1 or nstrips > 256


@pytest.mark.parametrize("kind,cols", [(1, 32), (2, 64)])
@pytest.mark.parametrize("h,w", [(970, 970), (130, 970), (3880, 3880), (75, 64), (9, 200)])
def test_conv3_sw_work_list_covers_every_pixel_once(kind, cols, h, w):
    """g_conv3_sw's segments: strips of 32 / 64 columns cut into runs of 4-row blocks, every pixel in exactly one, runs
    of equal length (+-1 block) per workgroup."""
    segs, sbeg = _segments(kind, h, w)
    cover = np.zeros((h, w), np.int32)
    for c0, y0, y1 in segs[:, :3]:
        assert c0 % cols == 0 and y0 % 4 == 0 and (y1 % 4 == 0 or y1 == h) and 0 <= y0 < y1 <= h
        cover[y0:y1, c0:min(w, c0 + cols)] += 1
    assert (cover == 1).all()
    blocks = np.array([sum(-(-int(s[2] - s[1]) // 4) for s in segs[sbeg[g]:sbeg[g + 1]]) for g in range(256)])
    assert blocks.max() - blocks.min() <= 1


@pytest.mark.parametrize("h,w,tile,bound", [(1080, 1920, 960, 0), (2160, 3840, 960, 0), (2160, 3840, 960, 4200000), (70, 75, 32, 0),
                                              (40, 150, 64, 0), (100, 500, 32, 0), (1080, 1920, 0, 0), (4320, 7680, 960, 0)])
def test_plane_batches(h, w, tile, bound):
    """generic_plan_batches: every plane (reference tile) of the frame in exactly one batch; a batch's planes of one width
    class, at most 16 of them and at most `bound` input pixels (default 2.2 M: the planes of a 1080p frame are ONE batch);
    batches numbered in running order."""
    L = _lib.load()
    need = ctypes.c_size_t(0)
    L.uva_debug_generic_batches(h, w, tile, 10, bound, None, 0, ctypes.byref(need))
    words = (ctypes.c_int32 * need.value)()
    assert L.uva_debug_generic_batches(h, w, tile, 10, bound, words, need.value, ctypes.byref(need)) == 0, L.uva_last_error()
    pl = np.frombuffer(words, np.int32).reshape(-1, 4)
    if tile > 0:
        assert len(pl) == -(-h // tile) * -(-w // tile)
        assert sum(int(p[0]) * int(p[1]) for p in pl) >= h * w           # (borders overlap)
    else:
        assert len(pl) == 1 and tuple(pl[0][:2]) == (h, w)
    limit = bound or 2200000
    seen = []
    for k in sorted(set(int(b) for b in pl[:, 2])):
        members = pl[pl[:, 2] == k]
        assert len(set(int(c) for c in members[:, 3])) == 1 and len(members) <= 16
        assert len(members) == 1 or sum(int(p[0]) * int(p[1]) for p in members) <= limit
        seen.append(k)
    assert seen == list(range(len(seen)))
    first = [int(np.argmax(pl[:, 2] == k)) for k in seen]
    assert first == sorted(first)
    if (h, w, tile, bound) == (1080, 1920, 960, 0):
        assert len(seen) == 1 and len(pl) == 4


def test_product_side_weight_writer_matches_the_restatements(tmp_path):
    """bench.py / tools write random-init weights for 4x_Valar_v1 (a missing blob upstream) with
    upscale_video_amd.synth.synthetic_weights; the tests use the numpy restatement's writer.  Same bytes."""
    from oracle import generic_oracle as go
    from upscale_video_amd.synth import synthetic_weights
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    synthetic_weights(VALAR, a, seed=3, gain=0.5)
    go.write_synthetic_bin(VALAR, b, seed=3, gain=0.5)
    assert open(a, "rb").read() == open(b, "rb").read()


def _segments_planes(kind, dims, grid=256):
    L = _lib.load()
    flat = (ctypes.c_int * (2 * len(dims)))(*[v for d in dims for v in d])
    need = ctypes.c_size_t(0)
    L.uva_debug_generic_segments_planes(kind, flat, len(dims), grid, None, 0, ctypes.byref(need), None)
    words = (ctypes.c_int32 * need.value)()
    sbeg = (ctypes.c_int * (grid + 1))()
    assert L.uva_debug_generic_segments_planes(kind, flat, len(dims), grid, words, need.value, ctypes.byref(need), sbeg) == 0, L.uva_last_error()
    return np.frombuffer(words, np.int32).reshape(-1, 8), list(sbeg)


# the planes of a 1080p frame cut into the reference's tiles (upscale_processing.py:499-516), and odd mixes
PLANE_SETS = [[(970, 970), (970, 970), (130, 970), (130, 970)], [(970, 960), (970, 970), (120, 960), (120, 970)],
              [(64, 64), (64, 33), (17, 64), (17, 33)], [(300, 200)] * 16, [(7, 45), (1080, 1920)]]


@pytest.mark.parametrize("dims", PLANE_SETS)
def test_rdb4_work_list_of_a_plane_batch(dims):
    """One rdb4 launch for several planes: every pixel of every plane owned by exactly one segment, the own-column rule
    of the one-plane list, STEPS (a segment's rows + 9 of pipeline fill) dealt evenly: every workgroup but the last one
    used is within one fill of the largest budget, and that budget is within a fill of the mean."""
    segs, sbeg = _segments_planes(0, dims)
    cover = [np.zeros(d, np.int32) for d in dims]
    for c0, yb, ye, own0, own1, pl in segs[:, :6]:
        h, w = dims[pl]
        assert 0 <= yb < ye <= h and 0 <= own0 < own1 <= w and c0 >= 0
        assert own0 >= (c0 + 3 if c0 > 0 else 0) and own1 <= (c0 + 45 if c0 + 48 < w else w)
        cover[pl][yb:ye, own0:own1] += 1
    assert all((c == 1).all() for c in cover)
    assert sbeg[0] == 0 and sbeg[-1] == len(segs) and all(b >= a for a, b in zip(sbeg, sbeg[1:]))
    steps = np.array([sum(int(s[2] - s[1]) + 9 for s in segs[sbeg[g]:sbeg[g + 1]]) for g in range(256)])
    used = np.nonzero(steps)[0]
    assert list(used) == list(range(len(used)))                      # (empty workgroups only at the end)
    assert (steps[used[:-1]] >= steps.max() - 9).all()
    assert steps.max() <= -(-int(steps.sum()) // 256) + 9


@pytest.mark.parametrize("kind,cols", [(1, 32), (2, 64)])
@pytest.mark.parametrize("dims", PLANE_SETS)
def test_conv3_sw_work_list_of_a_plane_batch(kind, cols, dims):
    segs, sbeg = _segments_planes(kind, dims)
    cover = [np.zeros(d, np.int32) for d in dims]
    for c0, y0, y1, _, _, pl in segs[:, :6]:
        h, w = dims[pl]
        assert c0 % cols == 0 and y0 % 4 == 0 and (y1 % 4 == 0 or y1 == h) and 0 <= y0 < y1 <= h
        cover[pl][y0:y1, c0:min(w, c0 + cols)] += 1
    assert all((c == 1).all() for c in cover)
    # blocks + one per segment start are what the workgroups share evenly (all but the last one used: within 1 of the budget)
    steps = np.array([sum(-(-int(s[2] - s[1]) // 4) + 1 for s in segs[sbeg[g]:sbeg[g + 1]]) for g in range(256)])
    used = np.nonzero(steps)[0]
    assert list(used) == list(range(len(used)))
    assert (steps[used[:-1]] >= steps.max() - 1).all() and steps.max() <= -(-int(steps.sum()) // 256) + 1


def test_valar_dense_blocks_are_recognised(uva):
    """find_rdbs: all 69 residual dense blocks of 4x_Valar_v1 run their first four convolutions as one launch (the first
    block's x is written into the chain's array by the 3-channel head convolution), and none of the 276 Concats copies."""
    L = _lib.load()
    net = uva.Net()
    assert net.load_param(VALAR) == 0, getattr(net, "last_error", "")
    info = (ctypes.c_int * 8)()
    assert L.uva_net_debug_generic_plan(net._h, info) == 0, L.uva_last_error()
    assert info[6] == 69 and info[1] == 276 and info[2] == 276 and info[3] == 0
