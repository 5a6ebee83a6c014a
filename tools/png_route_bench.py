"""File-to-file frames/s of the PNG route: N synthetic 1080p '<n>.extract.png' frames in a
RAM disk -> upscale_frames -> '<n>.png' (3840x2160), inputs deleted after their outputs exist, for
  * the reference-shaped route (a fresh spawn Pool per call, every worker imread -> net -> imwrite in turn), and
  * the persistent FramePool workers (decode threads -> pipelined GPU -> encode threads),
each with `-g 0` and `-g 0,0,0,0`.  usage: python tools/png_route_bench.py [frames=192]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from upscale_video_amd import _imageio, upscale_processing as up  # noqa: E402
from upscale_video_amd.synth import synthetic_frame  # noqa: E402

MODELS = os.path.join(ROOT, "models")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    base = tempfile.mkdtemp(prefix="png_route_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    src_dir = os.path.join(base, "src")
    os.makedirs(src_dir)
    frames = [synthetic_frame(1080, 1920, seed=i) for i in range(1, 9)]     # 8 distinct frames, copied N times
    t0 = time.perf_counter()
    for i, f in enumerate(frames, 1):
        _imageio.imwrite(os.path.join(src_dir, "%d.png" % i), f)
    print("PNG encode of a 1920x1080 frame on this host: %.1f ms" % ((time.perf_counter() - t0) / 8 * 1e3))
    big = np.repeat(np.repeat(frames[0], 2, 0), 2, 1)
    t0 = time.perf_counter()
    _imageio.imwrite(os.path.join(src_dir, "big.png"), big)
    print("PNG encode of a 3840x2160 frame on this host: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    t0 = time.perf_counter()
    for i in range(1, 9):
        _imageio.imread(os.path.join(src_dir, "%d.png" % i))
    print("PNG decode of a 1920x1080 frame on this host: %.1f ms" % ((time.perf_counter() - t0) / 8 * 1e3))
    try:
        for persistent in (False, True):
            for gpus in ([0], [0, 0, 0, 0]):
                if not persistent:
                    n_run = min(n, 16 * len(gpus))          # 4-15 frames/s: keep the run short
                else:
                    n_run = n
                work = os.path.join(base, "work")
                shutil.rmtree(work, ignore_errors=True)
                os.makedirs(work)
                os.chdir(work)
                for f in range(1, n_run + 1):
                    shutil.copy(os.path.join(src_dir, "%d.png" % ((f - 1) % 8 + 1)), "%d.extract.png" % f)
                up.PERSISTENT_WORKERS = persistent
                if persistent:                               # workers and nets exist before the batch, as in a long job
                    shutil.copy(os.path.join(src_dir, "1.png"), "0.extract.png")
                    up.upscale_frames(0, 0, 0, "extract", 2, gpus, 0, MODELS, "x_Compact_Pretrain", "input", "output")
                used = 0 if persistent else up.workers_spawned_so_far()
                t0 = time.perf_counter()
                up.upscale_frames(1, 1, n_run, "extract", 2, gpus, used, MODELS, "x_Compact_Pretrain", "input", "output")
                dt = time.perf_counter() - t0
                done = sum(os.path.exists("%d.png" % f) for f in range(1, n_run + 1))
                left = sum(os.path.exists("%d.extract.png" % f) for f in range(1, n_run + 1))
                assert done == n_run and left == 0, (done, left)
                print("%-34s -g %-8s %6.1f frames/s file-to-file (%d frames, %.2f s)" % (
                    "persistent FramePool workers" if persistent else "reference-shaped Pool per call",
                    ",".join(map(str, gpus)), n_run / dt, n_run, dt))
                up.shutdown_workers()
                os.chdir(base)
    finally:
        os.chdir(ROOT)
        shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
