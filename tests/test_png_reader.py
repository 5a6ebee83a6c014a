"""The host-side PNG reader (csrc/uva_pngread.cpp: from-scratch inflate + un-filter, SURVEY.md section 8 row f1, the imread
side of upscale_processing.py:263 / :487).  Its oracle is zlib / Pillow: every stream zlib can produce must inflate to the
same bytes, every PNG Pillow writes must decode to the same pixels, and damaged files must be refused, not crash."""
import ctypes
import io
import struct
import zlib

import numpy as np
import pytest

from upscale_video_amd import _lib


def inflate(stream, n):
    L = _lib.load()
    out = np.zeros(max(1, n), np.uint8)
    rc = L.uva_debug_zlib_decompress(stream, len(stream), out.ctypes.data, n)
    return rc, out[:n].tobytes()


def corpus():
    rng = np.random.default_rng(17)
    yy, xx = np.mgrid[0:300, 0:400]
    smooth = ((xx * 3 + yy * 2) % 256).astype(np.uint8).tobytes()
    return {
        "empty": b"",
        "one": b"x",
        "zeros": bytes(100000),
        "text": (b"the quick brown fox jumps over the lazy dog. " * 3000),
        "random": rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(),
        "smooth": smooth,
        "sparse": bytes(rng.choice([0, 0, 0, 0, 1, 255], 150000).astype(np.uint8)),
        "long_matches": (bytes(range(256)) * 40 + b"abc" * 9000 + b"z" * 70000),
        "dist_lt_8": b"".join(bytes([i % 251] * (i % 7 + 1)) * 40 for i in range(200)),
    }


@pytest.mark.parametrize("name", sorted(corpus()))
@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_inflate_equals_zlib(name, level):
    data = corpus()[name]
    for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED, zlib.Z_FILTERED):
        c = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
        stream = c.compress(data) + c.flush()
        rc, out = inflate(stream, len(data))
        assert rc == 0, (_lib.load().uva_last_error(), name, level, strategy)
        assert out == data


def test_inflate_multi_block_with_flushes():
    # sync / full flushes put empty stored blocks between dynamic blocks (what the GPU encoder's streams look like too)
    rng = np.random.default_rng(3)
    c = zlib.compressobj(6)
    parts, stream = [], b""
    for i in range(40):
        d = rng.integers(0, 7, int(rng.integers(1, 5000)), dtype=np.uint8).tobytes()
        parts.append(d)
        stream += c.compress(d) + c.flush(zlib.Z_SYNC_FLUSH if i % 3 else zlib.Z_FULL_FLUSH)
    stream += c.flush()
    data = b"".join(parts)
    rc, out = inflate(stream, len(data))
    assert rc == 0 and out == data


def test_inflate_rejects_damage_without_crashing():
    L = _lib.load()
    data = corpus()["text"]
    stream = zlib.compress(data, 6)
    rng = np.random.default_rng(8)
    refused = 0
    for trial in range(300):
        bad = bytearray(stream)
        k = int(rng.integers(0, len(bad)))
        bad[k] ^= 1 << int(rng.integers(0, 8))
        rc, out = inflate(bytes(bad), len(data))
        if rc == 0:
            assert out == data          # a flip that does not matter (e.g. FLEVEL bits) must still give the right bytes
        else:
            refused += 1
    assert refused >= 290
    for cut in (0, 1, 2, 5, len(stream) // 2, len(stream) - 5, len(stream) - 1):
        rc, _ = inflate(stream[:cut], len(data))
        assert rc != 0
    assert inflate(stream, len(data) - 1)[0] != 0 and inflate(stream, len(data) + 1)[0] != 0
    assert L.uva_last_error()


def test_inflate_fixed_block_then_truncated_stored_block():
    """the fast loop of a fixed-Huffman block used to leave the read pointer past the end of its input (two
    8-byte refills per iteration behind one bounds check), after which a stored block's length check compared a negative
    distance as unsigned and copied from beyond the buffer.  Streams cut at every position of the tail -- each in a heap
    buffer of exactly its own size -- must be refused, never read past, and the whole stream must still inflate."""
    rng = np.random.default_rng(23)
    lits = rng.integers(0, 144, 4000, dtype=np.uint8).tobytes()           # 8-bit codes in the fixed table: literals only
    c = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
    fixed = c.compress(lits) + c.flush(zlib.Z_SYNC_FLUSH)                   # fixed block(s) + an empty stored block
    tail = rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()
    raw = fixed + b"\x01" + struct.pack("<HH", len(tail), len(tail) ^ 0xffff) + tail     # final stored block
    data = lits + tail
    whole = b"\x78\x01" + raw + struct.pack(">I", zlib.adler32(data))
    assert zlib.decompress(whole) == data
    rc, out = inflate(whole, len(data))
    assert rc == 0 and out == data
    for cut in list(range(len(fixed) - 12, len(fixed) + 24)) + [len(raw) - 1, len(raw) - 700]:
        stream = bytes(bytearray(b"\x78\x01" + raw[:cut]))                  # its own allocation: reads past it are ASan / valgrind errors
        rc, _ = inflate(stream, len(data))
        assert rc != 0, cut


def png_of(arr, mode, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr, mode).save(buf, format="PNG", **kw)
    return buf.getvalue()


def decode(png):
    L = _lib.load()
    h, w = ctypes.c_int(0), ctypes.c_int(0)
    rc = L.uva_png_decode_bgr(png, len(png), None, 0, h, w)
    if rc:
        return rc, None
    out = np.zeros((h.value, w.value, 3), np.uint8)
    rc = L.uva_png_decode_bgr(png, len(png), out.ctypes.data, out.size, h, w)
    return rc, out


def pil_bgr(png):
    from PIL import Image
    with Image.open(io.BytesIO(png)) as im:
        return np.asarray(im.convert("RGB"))[:, :, ::-1]


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (9, 1), (37, 53), (270, 480)])
def test_png_reader_equals_pillow(shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 5 + yy) % 256, (xx + yy * 7) % 256, (xx * yy) % 256], -1).astype(np.uint8)
    noisy = np.clip(base.astype(int) + rng.integers(-9, 10, base.shape), 0, 255).astype(np.uint8)
    for arr in (base, noisy, rng.integers(0, 256, (h, w, 3), dtype=np.uint8)):
        for kw in ({}, {"compress_level": 1}, {"compress_level": 9, "optimize": True}):
            png = png_of(arr, "RGB", **kw)                     # Pillow picks filters adaptively: all five types occur
            rc, got = decode(png)
            assert rc == 0, _lib.load().uva_last_error()
            np.testing.assert_array_equal(got, arr[:, :, ::-1])
    rgba = np.dstack([noisy, rng.integers(0, 256, (h, w), dtype=np.uint8)])
    rc, got = decode(png_of(rgba, "RGBA"))
    assert rc == 0
    np.testing.assert_array_equal(got, noisy[:, :, ::-1])         # cv2.imread's default flag drops alpha
    grey = noisy[:, :, 0]
    rc, got = decode(png_of(grey, "L"))
    assert rc == 0
    np.testing.assert_array_equal(got, np.repeat(grey[:, :, None], 3, 2))


def test_png_reader_reads_what_the_gpu_encoder_writes():
    from test_png_encoder import frames, host_encode
    for name, img in frames().items():
        png, _ = host_encode(img)
        rc, got = decode(png)
        assert rc == 0, (name, _lib.load().uva_last_error())
        np.testing.assert_array_equal(got, img)


def test_png_reader_refuses_other_kinds_and_damage():
    rng = np.random.default_rng(1)
    arr = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr, "RGB").quantize(16).save(buf, format="PNG")
    assert decode(buf.getvalue())[0] == 2                          # palette
    a16 = (rng.integers(0, 65536, (20, 30), dtype=np.uint16))
    assert decode(png_of(a16, "I;16"))[0] == 2                     # 16-bit
    buf = io.BytesIO()
    Image.fromarray(arr, "RGB").save(buf, format="PNG", interlace=True) if False else None
    png = png_of(arr, "RGB")
    assert decode(png[:-1])[0] == 1 and decode(b"not a png at all, just some bytes" * 4)[0] == 1
    bad = bytearray(png)
    bad[len(png) // 2] ^= 0x10
    assert decode(bytes(bad))[0] == 1                              # chunk CRC
    # a wrong Adler-32 behind a correct chunk CRC
    pos, out = 8, bytearray(png[:8])
    while pos < len(png):
        n, = struct.unpack(">I", png[pos:pos + 4])
        kind, body = png[pos + 4:pos + 8], bytearray(png[pos + 8:pos + 8 + n])
        if kind == b"IDAT":
            body[-1] ^= 1
        out += struct.pack(">I", n) + kind + body + struct.pack(">I", zlib.crc32(kind + bytes(body)) & 0xFFFFFFFF)
        pos += 12 + n
    assert decode(bytes(out))[0] == 1
    assert b"Adler" in _lib.load().uva_last_error()
    # a header that claims a 16M x 16M image over a few hundred bytes of data: refused before anything is allocated
    pos, out = 8, bytearray(png[:8])
    while pos < len(png):
        n, = struct.unpack(">I", png[pos:pos + 4])
        kind, body = png[pos + 4:pos + 8], bytearray(png[pos + 8:pos + 8 + n])
        if kind == b"IHDR":
            body[0:8] = struct.pack(">II", 1 << 24, 1 << 24)
        out += struct.pack(">I", n) + kind + body + struct.pack(">I", zlib.crc32(kind + bytes(body)) & 0xFFFFFFFF)
        pos += 12 + n
    L = _lib.load()
    h, w = ctypes.c_int(0), ctypes.c_int(0)
    buf = np.zeros(16, np.uint8)
    assert L.uva_png_decode_bgr(bytes(out), len(out), buf.ctypes.data, 1 << 60, h, w) == 1
    assert b"larger than its data" in L.uva_last_error()
