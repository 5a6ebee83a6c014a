"""upscale_video_amd -- MI355X (gfx950) per-frame super-resolution path for davlee1972/upscale_video.

Only the hot path lives here: the ncnn-shaped shim (ncnn.py) over the HIP engine (libuva.so,
C ABI in include/uva.h) and the mirror of the reference's worker functions
(upscale_processing.py).  ffmpeg orchestration stays in the reference.
"""
__all__ = ["ncnn", "upscale_processing"]
