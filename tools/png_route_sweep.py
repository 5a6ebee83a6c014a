"""PNG -> PNG frames/s of the persistent FramePool route against the number of workers on one GPU and the encode
threads per worker.  usage: python tools/png_route_sweep.py [frames=240]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(gpus, n):
    from upscale_video_amd import _imageio, upscale_processing as up
    from upscale_video_amd.synth import synthetic_frame
    base = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    os.chdir(base)
    _imageio.imwrite("src.png", synthetic_frame(1080, 1920, seed=1))
    for i in range(0, n + 1):
        shutil.copy("src.png", "%d.extract.png" % i)
    models = os.path.join(ROOT, "models")
    up.upscale_frames(0, 0, 0, "extract", 2, gpus, 0, models, "x_Compact_Pretrain", "input", "output")
    t = time.perf_counter()
    up.upscale_frames(1, 1, n, "extract", 2, gpus, 0, models, "x_Compact_Pretrain", "input", "output")
    dt = time.perf_counter() - t
    up.shutdown_workers()
    os.chdir("/")
    shutil.rmtree(base)
    print("-g %-8s encode threads/worker %-3s decode %-2s : %6.1f frames/s" % (
        ",".join(map(str, gpus)), os.environ.get("UVA_ENCODE_THREADS"), os.environ.get("UVA_DECODE_THREADS", "4"), n / dt), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one([int(x) for x in sys.argv[2].split(",")], int(sys.argv[3]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
        for et, dt in ((8, 4), (16, 4), (32, 4), (32, 8)):
            for g in ("0", "0,0", "0,0,0,0"):
                env = dict(os.environ, UVA_ENCODE_THREADS=str(et), UVA_DECODE_THREADS=str(dt))
                subprocess.run([sys.executable, os.path.abspath(__file__), "--one", g, str(n)], env=env)
