"""PNG I/O with cv2.imread / cv2.imwrite semantics (u8 HWC, BGR channel order).

The reference uses OpenCV (upscale/upscale_processing.py:263, :288, :487, :519).  cv2 is used
when it is importable; otherwise Pillow, with the RGB<->BGR swap cv2 would have implied.
"""
import numpy as np

try:  # pragma: no cover - depends on the host image
    import cv2 as _cv2
except Exception:  # noqa: BLE001
    _cv2 = None


def _fast_png(path):
    """The library's own PNG reader (include/uva.h uva_png_decode_bgr: from-scratch inflate, 2-3x a zlib-based one;
    the GIL is released while it runs).  -> array, None (unreadable), or NotImplemented (not a PNG of the common kind,
    or the library is not built: the caller falls back to Pillow)."""
    import ctypes
    try:
        from . import _lib
        L = _lib.load()
        if not hasattr(L, "uva_png_decode_bgr"):
            return NotImplemented
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return None
    except Exception:  # noqa: BLE001 - no library: Pillow
        return NotImplemented
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        return NotImplemented
    h, w = ctypes.c_int(0), ctypes.c_int(0)
    rc = L.uva_png_decode_bgr(data, len(data), None, 0, h, w)
    if rc == 2:
        return NotImplemented
    if rc:
        return None
    out = np.empty((h.value, w.value, 3), np.uint8)
    rc = L.uva_png_decode_bgr(data, len(data), out.ctypes.data, out.size, h, w)
    return out if rc == 0 else None


def imread(path):
    """-> u8 [h][w][3] BGR, or None if unreadable (cv2.imread's convention)."""
    if _cv2 is not None:
        return _cv2.imread(path)
    img = _fast_png(path)
    if img is not NotImplemented:
        return img
    from PIL import Image
    try:
        with Image.open(path) as im:
            im = im.convert("RGB")
            w, h = im.size
            # Pillow's raw encoder writes the channels in BGR order itself (in C, without the GIL)
            return np.frombuffer(im.tobytes("raw", "BGR"), np.uint8).reshape(h, w, 3).copy()
    except Exception:  # noqa: BLE001
        return None


def png_bytes(bgr, level=1):
    """8-bit RGB PNG of a u8 BGR frame: every scanline with the Sub filter, one zlib stream at `level` with
    the Z_RLE strategy -- cv2.imwrite's own defaults (IMWRITE_PNG_COMPRESSION 1, IMWRITE_PNG_STRATEGY_RLE; the
    reference calls it at :288/:519, and OpenCV is not available here).  Pillow's encoder spends more time
    choosing a filter per row than compressing: a 3840x2160 frame takes 1.3 s there, 0.65 s with this filter
    and the default strategy, 0.27 s with Z_RLE (and is 10 % smaller).  zlib releases the GIL, so a worker's
    encode threads run in parallel."""
    import struct
    import zlib
    h, w, _ = bgr.shape
    flat = bgr[:, :, ::-1].reshape(h, 3 * w)
    rows = np.empty((h, 1 + 3 * w), np.uint8)
    rows[:, 0] = 1                                   # filter type 1 (Sub): byte minus the byte one pixel to the left
    rows[:, 1:4] = flat[:, :3]
    np.subtract(flat[:, 3:], flat[:, :-3], out=rows[:, 4:])

    deflate = zlib.compressobj(level, zlib.DEFLATED, 15, 9, zlib.Z_RLE)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", deflate.compress(rows.tobytes()) + deflate.flush()) + chunk(b"IEND", b""))


def to_u8(arr):
    """cv2's Mat::convertTo(CV_8U): round half to even, saturate to [0, 255]."""
    a = np.asarray(arr)
    if a.dtype == np.uint8:
        return a
    return np.clip(np.rint(a), 0, 255).astype(np.uint8)


def imwrite(path, arr):
    """arr: [h][w][3] BGR, u8 or float (floats are converted like cv2.imwrite does)."""
    if _cv2 is not None:
        return bool(_cv2.imwrite(path, arr))
    bgr = to_u8(arr)
    if bgr.ndim != 3 or bgr.shape[2] != 3:
        raise ValueError("frame must be [h][w][3]")
    with open(path, "wb") as f:
        f.write(png_bytes(bgr))
    return True
