"""File-to-file rate of the rawvideo streamer (python -m upscale_video_amd.rawvideo) on 1080p bgr24
frames held in /dev/shm: whole process wall time minus the wall time of a 1-frame run (interpreter
start, model load, first-use allocations)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from upscale_video_amd.synth import synthetic_frame

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
src = "/dev/shm/uva_in.bgr24"
fr = [synthetic_frame(1080, 1920, seed=i) for i in range(4)]
with open(src, "wb") as o:
    for i in range(N):
        o.write(fr[i % 4].tobytes())
base = [sys.executable, "-m", "upscale_video_amd.rawvideo", "-W", "1920", "-H", "1080"]


def wall(cmd, shell=False):
    t0 = time.perf_counter()
    subprocess.run(cmd, shell=shell, check=True, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL if not shell else None)
    return time.perf_counter() - t0


for args in (["-s", "2"], ["-s", "2", "-g", "0,0"], ["-s", "2", "-m", "a"], ["-s", "2", "-m", "a", "-g", "0,0"], ["-s", "4"], ["-s", "1", "-m", "a"]):
    t1 = wall(base + args + ["-i", src, "-o", "/dev/null", "--frames", "1"])
    tn = wall(base + args + ["-i", src, "-o", "/dev/null"])
    print(f"{' '.join(args):20s} file -> /dev/null : {N} frames in {tn:6.2f} s (start-up {t1:5.2f} s) = {(N - 1) / (tn - t1):7.1f} frames/s")
# file -> FILE: one worker with one and with four positional writers; several workers through ONE reader and ONE writer
# (--round-robin); a segment, reader and writers each into ONE shared output file -- by positional writes (the default) and
# through shared mappings (UVA_RAW_MMAP=1) --; one output file per worker.  On /dev/shm (tmpfs: page
# cache and nothing else) and on a directory of a REAL file system (UVA_BENCH_DIR, default: the repository's gpurun_out/).
from upscale_video_amd.rawvideo import filesystem_type
real_dir = os.environ.get("UVA_BENCH_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(real_dir, exist_ok=True)
for where in ("/dev/shm", real_dir):
    dst = os.path.join(where, "uva_out.bgr24")
    st = os.statvfs(where)
    free = st.f_bavail * st.f_frsize
    M = max(8, min(N, int(free * 0.5) // (2160 * 3840 * 3), 400 if where == "/dev/shm" else 160))
    print(f"--- output files in {where} ({filesystem_type(where)}, {free / 1e9:.0f} GB free), {M} frames")
    for args, per_lane, env in ((["-s", "2", "-g", "0", "--write-threads", "1"], False, {}), (["-s", "2", "-g", "0"], False, {}),
                                (["-s", "2", "-g", "0,0", "--round-robin"], False, {}),
                                (["-s", "2", "-g", "0,0"], False, {}), (["-s", "2", "-g", "0,0"], False, {"UVA_RAW_MMAP": "1"}),
                                (["-s", "2", "-g", "0,0"], True, {}),
                                (["-s", "2", "-g", "0,0,0,0"], False, {}), (["-s", "2", "-g", "0,0,0,0"], False, {"UVA_RAW_MMAP": "1"}),
                                (["-s", "2", "-g", "0,0,0,0"], True, {}),
                                (["-s", "2", "-g", "0,0,0,0,0,0,0,0"], False, {}), (["-s", "2", "-g", "0,0,0,0,0,0,0,0"], True, {})):
        k = len(args[3].split(","))
        outs = [dst + ".%d" % i for i in range(k)] if per_lane else [dst]
        os.environ.update(env)
        try:
            t1 = wall(base + args + ["-i", src, "-o", ",".join(outs), "--frames", str(k)])
            tn = wall(base + args + ["-i", src, "-o", ",".join(outs), "--frames", str(M)])
        finally:
            for key in env:
                os.environ.pop(key, None)
        label = " ".join(args) + (" -o one file per worker" if per_lane else "") + (" through shared mappings (UVA_RAW_MMAP=1)" if env else "")
        print(f"{label:64s} file -> file : {M} frames in {tn:6.2f} s (start-up {t1:5.2f} s) = {(M - k) / (tn - t1):7.1f} frames/s", flush=True)
        for o in outs:
            os.remove(o)
t1 = wall(base + ["-s", "2", "-i", src, "-o", "/dev/null", "--frames", "1"])
tn = wall(f"cat {src} | {' '.join(base)} -s 2 2>/dev/null | cat > /dev/null", shell=True)
print(f"-s 2         pipe -> pipe      : {N} frames in {tn:6.2f} s = {(N - 1) / (tn - t1):7.1f} frames/s")
os.remove(src)
