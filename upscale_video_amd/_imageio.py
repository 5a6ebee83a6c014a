"""PNG I/O with cv2.imread / cv2.imwrite semantics (u8 HWC, BGR channel order).

The reference uses OpenCV (upscale/upscale_processing.py:263, :288, :487, :519).  cv2 is used
when it is importable; otherwise Pillow, with the RGB<->BGR swap cv2 would have implied.
"""
import numpy as np

try:  # pragma: no cover - depends on the host image
    import cv2 as _cv2
except Exception:  # noqa: BLE001
    _cv2 = None


def imread(path):
    """-> u8 [h][w][3] BGR, or None if unreadable (cv2.imread's convention)."""
    if _cv2 is not None:
        return _cv2.imread(path)
    from PIL import Image
    try:
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    except Exception:  # noqa: BLE001
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


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
    from PIL import Image
    bgr = to_u8(arr)
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1]), "RGB").save(path, format="PNG", compress_level=1)
    return True
