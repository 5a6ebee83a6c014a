"""ONE output file on tmpfs with 1 / 4 / 8 workers: the one-writer route the streamer now takes there against the segment route
(UVA_RAW_TMPFS_SEGMENTS=1).  The output file is removed before every run (a file that exists keeps its pages: rewriting it is not
what a job does)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from upscale_video_amd.synth import synthetic_frame
N = 400
src, dst = "/dev/shm/uva_in.bgr24", "/dev/shm/uva_out.bgr24"
fr = [synthetic_frame(1080, 1920, seed=i) for i in range(4)]
with open(src, "wb") as o:
    for i in range(N):
        o.write(fr[i % 4].tobytes())
base = f"{sys.executable} -m upscale_video_amd.rawvideo -W 1920 -H 1080"
def wall(cmd, env=None):
    if os.path.exists(dst):
        os.remove(dst)
    t0 = time.perf_counter()
    subprocess.run(cmd, shell=True, check=True, env=dict(os.environ, **(env or {})), stdin=subprocess.DEVNULL, timeout=300)
    return time.perf_counter() - t0
for rep in range(2):
    for g in ("0", "0,0,0,0", "0,0,0,0,0,0,0,0"):
        k = len(g.split(","))
        for label, env in (("one writer (default on tmpfs)", {}), ("segments (UVA_RAW_TMPFS_SEGMENTS=1)", {"UVA_RAW_TMPFS_SEGMENTS": "1"})):
            t1 = wall(f"{base} -s 2 -g {g} -i {src} -o {dst} --frames {k} 2>/dev/null", env)
            tn = wall(f"{base} -s 2 -g {g} -i {src} -o {dst} 2>/dev/null", env)
            print(f"-s 2 -g {g:16s} file -> ONE file on tmpfs, {label:36s}: {(N - k) / (tn - t1):7.1f} frames/s", flush=True)
os.remove(src); os.remove(dst)
