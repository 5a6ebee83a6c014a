"""Where a PNG input's 16 ms go: uva_png_decode_bgr (csrc/uva_pngread.cpp: the reference's cv2.imread,
upscale/upscale_processing.py:263,487) = inflate + un-filter + RGB->BGR layout.  Timed on one host core, per 1080p frame: the
inflate alone (uva_debug_zlib_decompress on the file's concatenated IDAT stream) against the whole call, for the three kinds of
file a worker meets -- ffmpeg's extract (here: PIL's encoder at its default level 6, adaptive filters, like ffmpeg's png
encoder), this package's own GPU-deflated files (fixed Huffman, filter Sub), and a stored (level 0) file as the floor.
If the difference (un-filter + layout) is small, moving it to the device buys nothing; the inflate's share is the answer.

    python tools/png_decode_split.py [repeats]
"""
import io
import os
import struct
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upscale_video_amd import _lib                     # noqa: E402
from upscale_video_amd.synth import synthetic_frame    # noqa: E402


def idat_stream(png):
    pos, out = 8, []
    while pos < len(png):
        n, kind = struct.unpack(">I4s", png[pos:pos + 8])
        if kind == b"IDAT":
            out.append(png[pos + 8:pos + 8 + n])
        pos += 12 + n
    return b"".join(out)


def png_of(frame_bgr, level, filt=None):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(frame_bgr[:, :, ::-1]).save(buf, "PNG", compress_level=level)
    return buf.getvalue()


def best(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    L = _lib.load()
    h, w = 1080, 1920
    frame = synthetic_frame(h, w, seed=1)
    files = {"PIL level 6 (adaptive filters: an ffmpeg-like extract)": png_of(frame, 6),
             "PIL level 1": png_of(frame, 1),
             "stored (level 0: the floor of the un-filter + layout part)": png_of(frame, 0)}
    try:
        from upscale_video_amd._imageio import imwrite_bytes      # this package's own encoder, if it has a host entry
        files["this package's encoder"] = imwrite_bytes(frame)
    except Exception:  # noqa: BLE001
        pass
    raw_len = h * (1 + 3 * w)
    out = np.empty((h, w, 3), np.uint8)
    raw = np.empty(raw_len, np.uint8)
    import ctypes
    hh, ww = ctypes.c_int(0), ctypes.c_int(0)
    print("one host core, %d x %d, best / median of %d" % (w, h, reps))
    for name, png in files.items():
        z = idat_stream(png)
        zb = np.frombuffer(z, np.uint8)
        pb = np.frombuffer(png, np.uint8)
        assert L.uva_debug_zlib_decompress(zb.ctypes.data, zb.size, raw.ctypes.data, raw_len) == 0, L.uva_last_error()
        assert bytes(raw) == zlib.decompress(z)
        t_inf = best(lambda: L.uva_debug_zlib_decompress(zb.ctypes.data, zb.size, raw.ctypes.data, raw_len), reps)
        assert L.uva_png_decode_bgr(pb.ctypes.data, pb.size, out.ctypes.data, out.size, ctypes.byref(hh), ctypes.byref(ww)) == 0
        assert np.array_equal(out, frame)
        t_all = best(lambda: L.uva_png_decode_bgr(pb.ctypes.data, pb.size, out.ctypes.data, out.size, ctypes.byref(hh), ctypes.byref(ww)), reps)
        t_zlib = best(lambda: zlib.decompress(z), reps)
        filt = np.bincount(np.frombuffer(zlib.decompress(z), np.uint8).reshape(h, 1 + 3 * w)[:, 0], minlength=5)
        print("%-62s %5.2f MB  inflate %6.2f / %6.2f ms   whole decode %6.2f / %6.2f ms   un-filter + layout + CRC = %5.2f ms (%2.0f %%)   "
              "[zlib's inflate %6.2f ms]  filters N/S/U/A/P = %s"
              % (name, len(png) / 1e6, t_inf[0] * 1e3, t_inf[1] * 1e3, t_all[0] * 1e3, t_all[1] * 1e3, (t_all[0] - t_inf[0]) * 1e3,
                 100 * (t_all[0] - t_inf[0]) / t_all[0], t_zlib[0] * 1e3, filt.tolist()))


if __name__ == "__main__":
    main()
