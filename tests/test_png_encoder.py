"""The GPU PNG encoder (csrc/uva_png.hip.h, SURVEY.md section 8 row f1: the imwrite side of the reference's per-frame hop,
upscale_processing.py:288 / :519).  CPU part: the code tables, block headers, Adler-32 combination, chunk CRC and
framing -- exercised through the host restatement of the kernel (uva_debug_png_deflate_host, a test hook) and checked
with independent PNG readers (Pillow, and zlib itself on the IDAT payload).  The `gpu` part checks that the kernel
produces the same bytes as that restatement and that frames survive the file round trip bit-exactly."""
import ctypes
import io
import struct
import zlib

import numpy as np
import pytest

from upscale_video_amd import _lib


def frames():
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:97, 0:131]
    smooth = np.stack([(xx * 2 + yy) % 256, (xx + yy * 3) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8)
    noisy = np.clip(smooth.astype(np.int32) + rng.integers(-6, 7, smooth.shape), 0, 255).astype(np.uint8)
    return {
        "smooth": smooth,
        "noisy": noisy,
        "random": rng.integers(0, 256, (64, 200, 3), dtype=np.uint8),
        "flat": np.full((33, 17, 3), 128, np.uint8),
        "1x1": np.array([[[1, 2, 3]]], np.uint8),
        "one_row": rng.integers(0, 256, (1, 500, 3), dtype=np.uint8),
        "many_blocks": rng.integers(0, 256, (40, 5000, 3), dtype=np.uint8),      # 3 rows per block, ragged last block
        # 3 pixels wide -- a raw row takes 12 bytes in the staging area for its 10 filtered ones, so the rows of a
        # block are bounded by the staging area, not by the filtered bytes
        "narrow_tall": rng.integers(0, 256, (5200, 3, 3), dtype=np.uint8),
    }


def host_encode(img):
    L = _lib.load()
    h, w, _ = img.shape
    n = L.uva_png_workspace_bytes(h, w)
    assert n > 0
    ws = np.zeros(n, np.uint8)
    assert L.uva_debug_png_deflate_host(img.ctypes.data, h, w, w * 3, ws.ctypes.data, n) == 0, L.uva_last_error()
    ln = ctypes.c_size_t(0)
    L.uva_png_assemble(ws.ctypes.data, h, w, None, 0, ln)            # size query
    out = np.zeros(ln.value, np.uint8)
    assert L.uva_png_assemble(ws.ctypes.data, h, w, out.ctypes.data, out.size, ln) == 0, L.uva_last_error()
    return out.tobytes(), ws


def decode(png):
    from PIL import Image
    with Image.open(io.BytesIO(png)) as im:
        assert im.mode == "RGB"
        return np.asarray(im)[:, :, ::-1]


@pytest.mark.parametrize("name", sorted(frames()))
def test_host_restatement_makes_a_png_every_reader_accepts(name):
    img = frames()[name]
    png, _ = host_encode(img)
    np.testing.assert_array_equal(decode(png), img)
    # chunk walk: lengths, CRCs, and the IDAT payload through zlib itself (checks the Adler-32 combination)
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, kinds, idat = 8, [], b""
    while pos < len(png):
        n, = struct.unpack(">I", png[pos:pos + 4])
        kind, body = png[pos + 4:pos + 8], png[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", png[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(kind + body) & 0xFFFFFFFF
        kinds.append(kind)
        if kind == b"IDAT":
            idat += body
        pos += 12 + n
    assert kinds == [b"IHDR", b"IDAT", b"IEND"]
    h, w, _ = img.shape
    raw = zlib.decompress(idat)
    assert len(raw) == h * (3 * w + 1)
    rows = np.frombuffer(raw, np.uint8).reshape(h, 3 * w + 1)
    assert (rows[:, 0] == 1).all()                               # every scanline: filter type Sub


def test_the_cheapest_table_is_chosen_per_block():
    # literals only: a flat frame costs a little over 1 bit per byte (plus ~200 bytes of headers, which is what the
    # ratio of this tiny one shows), noise about 8 bits per byte: the four tables cover that range
    f = frames()
    big_flat = np.full((256, 256, 3), 77, np.uint8)
    png, _ = host_encode(big_flat)
    assert len(png) / big_flat.size < 0.14
    for name, lo, hi in (("flat", 0.0, 0.35), ("smooth", 0.1, 0.75), ("random", 0.95, 1.15)):
        png, _ = host_encode(f[name])
        ratio = len(png) / f[name].size
        assert lo <= ratio <= hi, (name, ratio)


def test_workspace_rules():
    L = _lib.load()
    assert L.uva_png_workspace_bytes(10, 16384) == 0              # one filtered row must fit a block
    assert L.uva_png_workspace_bytes(0, 10) == 0
    assert L.uva_png_workspace_bytes(2160, 3840) > 0
    ws = np.zeros(L.uva_png_workspace_bytes(8, 8), np.uint8)
    ln = ctypes.c_size_t(0)
    assert L.uva_png_assemble(ws.ctypes.data, 8, 8, None, 0, ln) != 0      # never filled: refused, not framed
    assert b"kernel not run" in L.uva_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(frames()))
def test_kernel_equals_host_restatement(name):
    from upscale_video_amd import ncnn
    img = frames()[name]
    h, w, _ = img.shape
    ws = ncnn.PngWorkspace(h, w)
    L = _lib.load()
    _lib.check(L.uva_png_deflate_u8(0, img.ctypes.data, h, w, w * 3, ws.buf.ctypes.data, ws.buf.nbytes))
    png = bytes(ws.file_bytes())
    ref, _ = host_encode(img)
    assert png == ref
    np.testing.assert_array_equal(decode(png), img)


@pytest.mark.gpu
def test_net_result_through_the_png_route_equals_the_frame_route():
    import os
    from upscale_video_amd import ncnn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    net = ncnn.Net()
    net.set_vulkan_device(0)
    base = os.path.join(root, "models", "2x_Compact_Pretrain")
    assert net.load_param(base + ".param") == 0 and net.load_model(base + ".bin") == 0
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, (270, 480, 3), dtype=np.uint8) for _ in range(5)]
    want = [net.process_u8(im, tile_size=200, border=10).copy() for im in imgs]
    spaces = [ncnn.PngWorkspace(540, 960) for _ in range(3)]
    tickets = []
    got = []
    for i, im in enumerate(imgs):                       # three in flight, like the workers
        if len(tickets) == 3:
            got.append(bytes(net.collect_u8(tickets.pop(0)).file_bytes()))
        tickets.append(net.submit_u8_png(im, workspace=spaces[i % 3], tile_size=200, border=10))
    while tickets:
        got.append(bytes(net.collect_u8(tickets.pop(0)).file_bytes()))
    for png, ref in zip(got, want):
        np.testing.assert_array_equal(decode(png), ref)


@pytest.mark.gpu
def test_a_frame_that_compresses_worse_than_the_one_before_it():
    """The download that travels with a frame carries as many packed bytes as the previous frame had (+ 6 %); what a
    larger frame has beyond that is fetched by collect_u8.  Flat frames (tiny files) alternate with noise (the largest)."""
    import os
    from upscale_video_amd import ncnn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    net = ncnn.Net()
    net.set_vulkan_device(0)
    base = os.path.join(root, "models", "2x_Compact_Pretrain")
    assert net.load_param(base + ".param") == 0 and net.load_model(base + ".bin") == 0
    rng = np.random.default_rng(12)
    imgs = []
    for i in range(6):
        imgs.append(np.full((270, 480, 3), 40 * i, np.uint8) if i % 2 == 0 else rng.integers(0, 256, (270, 480, 3), dtype=np.uint8))
    want = [net.process_u8(im, tile_size=0).copy() for im in imgs]
    spaces = [ncnn.PngWorkspace(540, 960) for _ in range(3)]
    sizes = []
    for depth in (1, 3):                                 # one at a time (the guess is always the previous frame), and pipelined
        tickets, got = [], []
        for i, im in enumerate(imgs):
            if len(tickets) == depth:
                got.append(bytes(net.collect_u8(tickets.pop(0)).file_bytes()))
            tickets.append(net.submit_u8_png(im, workspace=spaces[i % 3], tile_size=0))
        while tickets:
            got.append(bytes(net.collect_u8(tickets.pop(0)).file_bytes()))
        for png, ref in zip(got, want):
            np.testing.assert_array_equal(decode(png), ref)
        sizes = [len(g) for g in got]
    assert min(sizes[1::2]) > 4 * max(sizes[0::2])        # the test does alternate small and large


@pytest.mark.gpu
def test_full_size_frame_round_trip():
    from upscale_video_amd import ncnn
    from upscale_video_amd.synth import synthetic_frame
    img = synthetic_frame(2160, 3840, seed=3)
    png = ncnn.png_encode_u8(img)
    np.testing.assert_array_equal(decode(png), img)
    assert len(png) < img.size                           # it does compress
