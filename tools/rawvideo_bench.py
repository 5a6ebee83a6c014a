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
# file -> FILE (a real output file in /dev/shm: page cache): one worker; several workers through ONE reader and ONE writer
# (--round-robin), with a segment, reader and writer each into ONE shared output file, and into one output file each
dst = "/dev/shm/uva_out.bgr24"
free = os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize
M = max(8, min(N, int(free * 0.6) // (2160 * 3840 * 3)))
for args, per_lane in ((["-s", "2", "-g", "0"], False), (["-s", "2", "-g", "0,0", "--round-robin"], False), (["-s", "2", "-g", "0,0"], False),
                       (["-s", "2", "-g", "0,0"], True), (["-s", "2", "-g", "0,0,0,0", "--round-robin"], False), (["-s", "2", "-g", "0,0,0,0"], False),
                       (["-s", "2", "-g", "0,0,0,0"], True), (["-s", "2", "-g", "0,0,0,0,0,0,0,0"], True)):
    k = len(args[3].split(","))
    outs = [dst + ".%d" % i for i in range(k)] if per_lane else [dst]
    t1 = wall(base + args + ["-i", src, "-o", ",".join(outs), "--frames", str(k)])
    tn = wall(base + args + ["-i", src, "-o", ",".join(outs), "--frames", str(M)])
    label = " ".join(args) + (" -o one file per worker" if per_lane else "")
    print(f"{label:52s} file -> file : {M} frames in {tn:6.2f} s (start-up {t1:5.2f} s) = {(M - k) / (tn - t1):7.1f} frames/s")
    for o in outs:
        os.remove(o)
t1 = wall(base + ["-s", "2", "-i", src, "-o", "/dev/null", "--frames", "1"])
tn = wall(f"cat {src} | {' '.join(base)} -s 2 2>/dev/null | cat > /dev/null", shell=True)
print(f"-s 2         pipe -> pipe      : {N} frames in {tn:6.2f} s = {(N - 1) / (tn - t1):7.1f} frames/s")
os.remove(src)
