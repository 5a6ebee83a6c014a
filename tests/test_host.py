"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol that
include/uva.h declares, the .param/.bin loader accepts exactly the SRVGGNetCompact graphs, the
MFMA weight image is a faithful permutation of the OIHW weights, and -- with no GPU -- every
compute entry point fails loudly (there is no CPU path in the product)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_net, model_paths


def test_library_exports_every_declared_symbol(uva):
    from upscale_video_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "uva.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)   # prototypes only, not the comments
    instr = re.findall(r"#ifdef UVA_INSTRUMENT(.*?)#endif", hdr, flags=re.S)
    hdr = re.sub(r"#ifdef UVA_INSTRUMENT.*?#endif", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(uva_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    assert sorted(_lib.SYMBOLS) == declared
    # the instrumentation entry points are declared for -DUVA_INSTRUMENT builds only and are NOT in the product library
    instr_declared = sorted(set(re.findall(r"\b(uva_[a-z0-9_]+)\s*\(", "".join(instr))))
    assert instr_declared == sorted(_lib.INSTRUMENT_SYMBOLS)
    assert not any(hasattr(ctypes.CDLL(_lib.LIB_PATH), n) for n in instr_declared)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert _lib.load().uva_abi_version() == _lib.ABI_VERSION == 15


@pytest.mark.parametrize("key,facts", [("2x", (2, 64, 18)), ("4x", (4, 64, 18)), ("1x", (1, 24, 10))])
def test_loader_accepts_compact_models(uva, key, facts):
    net = load_net(uva, key)
    assert (net.scale, net.num_features, net.num_convs) == facts


def test_loader_rejects_bad_files(uva, tmp_path):
    net = uva.Net()
    assert net.load_param(str(tmp_path / "missing.param")) != 0
    p, b = model_paths("2x")
    assert net.load_model(b) != 0            # load_model before load_param
    assert net.load_param(p) == 0
    raw = open(b, "rb").read()
    (tmp_path / "short.bin").write_bytes(raw[:-7])
    assert net.load_model(str(tmp_path / "short.bin")) != 0
    (tmp_path / "long.bin").write_bytes(raw + b"\0\0\0\0")
    assert net.load_model(str(tmp_path / "long.bin")) != 0 and "unread bytes" in net.last_error
    # weights of another graph do not fit
    assert net.load_model(model_paths("1x")[1]) != 0
    assert net.load_model(b) == 0


def test_negative_device_rejected(uva):
    net = uva.Net()
    with pytest.raises(Exception, match="no CPU path"):
        net.set_vulkan_device(-1)


def _unpack_conv(pk, nf, cout):
    spp = nf // 8
    ko_n = 9 * spp
    ks_n, mf = (ko_n + 1) // 2, (cout + 31) // 32
    img = pk.view(np.float16).astype(np.float32).reshape(ks_n, mf, 64, 8)
    w = np.zeros((cout, nf, 9), np.float32)
    seen = np.zeros((cout, nf, 9), bool)
    for ks in range(ks_n):
        for m in range(mf):
            for lane in range(64):
                i, half = lane & 31, lane >> 5
                co, ko = 32 * m + i, 2 * ks + half
                if co >= cout or ko >= ko_n:
                    assert not img[ks, m, lane].any()
                    continue
                tap, octet = divmod(ko, spp)
                w[co, octet * 8:octet * 8 + 8, tap] = img[ks, m, lane]
                seen[co, octet * 8:octet * 8 + 8, tap] = True
    assert seen.all()
    return w.reshape(cout, nf, 3, 3)


def _unpack_trunk64(pk):
    """trunk_kernel<64> (v_mfma_f32_16x16x32_f16): [18 k-steps = 2*tap + ch][4 blocks][64 lanes][8];
    lane = (octet << 4) | i holds output channel 16*mb + i, input channels 32*ch + 8*octet .. +7."""
    img = pk.view(np.float16).astype(np.float32).reshape(18, 4, 64, 8)
    w = np.zeros((64, 64, 9), np.float32)
    seen = np.zeros((64, 64, 9), bool)
    for ks in range(18):
        for mb in range(4):
            for lane in range(64):
                co, ci0 = 16 * mb + (lane & 15), 32 * (ks & 1) + 8 * (lane >> 4)
                w[co, ci0:ci0 + 8, ks >> 1] = img[ks, mb, lane]
                seen[co, ci0:ci0 + 8, ks >> 1] = True
    assert seen.all()
    return w.reshape(64, 64, 3, 3)


@pytest.mark.parametrize("key", ["2x", "4x", "1x"])
def test_packed_weights_are_a_permutation_of_oihw(uva, oracle_models, key):
    """The kernel's B operand supplies, for k-step ks and lane half h, K octet ko = 2ks+h = channels
    8*(ko % (nf/8)).. of tap ko // (nf/8); the packed A image must put the matching weights there."""
    net = load_net(uva, key)
    om = oracle_models[key]
    nf = net.num_features
    for idx in (1, net.num_convs // 2, net.num_convs - 1):
        w, _, _ = om.conv(idx)
        trunk64 = nf == 64 and idx + 1 < net.num_convs
        got = _unpack_trunk64(net.debug_packed_weights(idx)) if trunk64 else _unpack_conv(net.debug_packed_weights(idx), nf, w.shape[0])
        with np.errstate(over="ignore"):
            want = w.astype(np.float16).astype(np.float32)
        assert np.array_equal(got, want), (key, idx)
    if nf == 64:
        # tail_kernel's image of the last convolution: [18][MB][64][8], rows beyond cout are zero
        wt, _, _ = om.conv(net.num_convs - 1)
        cout = wt.shape[0]
        mb = (cout + 15) // 16
        img = net.debug_packed_weights(-1).view(np.float16).astype(np.float32).reshape(18, mb, 64, 8)
        rec = np.zeros((16 * mb, 64, 9), np.float32)
        for ks in range(18):
            for m in range(mb):
                for lane in range(64):
                    ci0 = 32 * (ks & 1) + 8 * (lane >> 4)
                    rec[16 * m + (lane & 15), ci0:ci0 + 8, ks >> 1] = img[ks, m, lane]
        assert not rec[cout:].any()
        assert np.array_equal(rec[:cout].reshape(cout, 64, 3, 3), wt.astype(np.float16).astype(np.float32)), key
    # head: K = [tap][4] (3 channels + zero), octet o = 2ks+h holds taps 2o, 2o+1
    w0, _, _ = om.conv(0)
    pk = net.debug_packed_weights(0).view(np.float16).astype(np.float32)
    mf = (nf + 31) // 32
    img = pk.reshape(3, mf, 64, 8)
    rec = np.zeros((nf, 3, 9), np.float32)
    for ks in range(3):
        for m in range(mf):
            for lane in range(64):
                co, o = 32 * m + (lane & 31), 2 * ks + (lane >> 5)
                for e in range(8):
                    tap, ch = 2 * o + (e >> 2), e & 3
                    v = img[ks, m, lane, e]
                    if co < nf and tap < 9 and ch < 3:
                        rec[co, ch, tap] = v
                    else:
                        assert v == 0
    with np.errstate(over="ignore"):
        assert np.array_equal(rec.reshape(nf, 3, 3, 3), w0.astype(np.float16).astype(np.float32))


def test_mat_from_pixels_and_normalize_match_oracle(uva, oracle):
    img = oracle.synthetic_frame(9, 13, kind="random")
    m = uva.Mat.from_pixels(img, uva.Mat.PixelType.PIXEL_BGR, 13, 9)
    m.substract_mean_normalize([], [1 / 255.0] * 3)
    assert np.array_equal(np.array(m), oracle.from_pixels_normalize(img))
    with pytest.raises(ValueError):
        uva.Mat.from_pixels(img, uva.Mat.PixelType.PIXEL_BGR, 9, 13)


def test_extractor_takes_the_graph_s_blob_names(uva, tmp_path):
    """ex.input / ex.extract check their names against the loaded graph's input and output blobs (the reference passes
    model_input / model_output through, upscale_processing.py:72-73, :278-280), not against fixed strings."""
    from upscale_video_amd import ncnn
    p, _ = model_paths("1x")
    assert ncnn._param_blob_names(p) == ("input", "output")
    renamed = open(p).read().replace(" input", " data_in").replace("output", "result")
    (tmp_path / "renamed.param").write_text(renamed)
    assert ncnn._param_blob_names(str(tmp_path / "renamed.param")) == ("data_in", "result")
    net = load_net(uva, "1x")
    ex = net.create_extractor()
    assert net.blob_names == ("input", "output")
    assert ex.input("data", uva.Mat(np.zeros((3, 4, 4), np.float32))) == -1
    assert ex.extract("output") == (-1, None)            # nothing was fed
    net.blob_names = ("data_in", "result")               # as load_param of such a graph would have left it
    assert ex.input("input", uva.Mat(np.zeros((3, 4, 4), np.float32))) == -1
    assert ex.input("data_in", uva.Mat(np.zeros((3, 4, 4), np.float32))) == 0


def test_compute_fails_loudly_without_gpu(uva):
    if uva.get_gpu_count() > 0:
        pytest.skip("a GPU is present")
    assert uva.get_default_gpu_index() == -1
    net = load_net(uva, "1x")
    from upscale_video_amd._lib import UvaError
    with pytest.raises(UvaError, match="no HIP device|no CPU path"):
        net.process_u8(np.zeros((8, 8, 3), np.uint8))
    ex = net.create_extractor()
    ex.input("input", uva.Mat(np.zeros((3, 8, 8), np.float32)))
    with pytest.raises(UvaError):
        ex.extract("output")


def test_missing_library_is_an_error(monkeypatch, uva):
    from upscale_video_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libuva.so")
    with pytest.raises(_lib.UvaError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "upscale_video_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "uvoracle" not in text and "liboracle" not in text and "oracle/" not in text, f


def test_debug_switches_need_the_opt_in(uva):
    """A stray UVA_* variable in a user's shell must not change what the drop-in library does: the A/B and
    debug switches are read through csrc/uva_devutil.hip.h debug_env, which returns them only under UVA_DEBUG_SWITCHES=1 and
    otherwise names the ignored variable once on stderr.  Observable without a GPU: the trunkw step lists of a frame whose
    two 70-wide planes fold their last strips (host-only hook uva_debug_trunkw_schedule); UVA_TW_FOLD=0 un-folds them."""
    import subprocess
    import sys
    code = ("import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from upscale_video_amd import ncnn\n"
            "from test_trunkw_schedule import schedule\n"
            "steps, nsteps, planes, guard = schedule(ncnn, 40, 120, 60, 10)\n"
            "print(int(nsteps.sum()), hashlib.sha256(steps.tobytes()).hexdigest()[:16])\n") % (ROOT, os.path.join(ROOT, "tests"))

    def run(**env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("UVA_")}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        return r.stdout.split(), r.stderr
    base, err = run()
    assert "IGNORED" not in err
    ignored, err = run(UVA_TW_FOLD="0", UVA_TRUNK_WINO="0")
    assert ignored == base                                               # same step lists, byte for byte
    assert err.count("UVA_TW_FOLD is set but IGNORED") == 1              # named, once
    honoured, err = run(UVA_TW_FOLD="0", UVA_DEBUG_SWITCHES="1")
    assert honoured != base and int(honoured[0]) > int(base[0]) and "IGNORED" not in err
    again, _ = run(UVA_DEBUG_SWITCHES="1")
    assert again == base
