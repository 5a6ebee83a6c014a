#!/bin/bash
# Run ON THE MI355X BOX: the round's LAST build once more -- every GPU test, smoke, the driver's form of the bench line, the 1x net's line,
# the pipe route -- so that HEAD itself (not only the build of the evidence set) is known green.
exec < /dev/null
O=gpurun_out/r06_last; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -q) > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
(python -c "import __graft_entry__ as g; g.smoke(); print('smoke: ok')" 2>&1 | tail -2) > $O/smoke.txt; cat $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err; cut -c1-330 $O/bench_driver_form.json
python bench.py --workload 1x_hurrdeblur_1080p --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_1x.json 2>> $O/bench.err
python bench.py --workload 4x_valar_1080p --steps 12 --warmup 2 --no-cpu-baseline > $O/bench_valar.json 2>> $O/bench.err
python -c "
import json
for f in ('bench_driver_form','bench_1x','bench_valar'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['frac'])"
python - > $O/rawvideo_pipe.txt 2>&1 <<'PY'
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
from upscale_video_amd.synth import synthetic_frame
N = 400
src = "/dev/shm/uva_in.bgr24"
fr = [synthetic_frame(1080, 1920, seed=i) for i in range(4)]
with open(src, "wb") as o:
    for i in range(N):
        o.write(fr[i % 4].tobytes())
base = f"{sys.executable} -m upscale_video_amd.rawvideo -W 1920 -H 1080"
def wall(cmd, env=None):
    t0 = time.perf_counter()
    subprocess.run(cmd, shell=True, check=True, env=dict(os.environ, **(env or {})), stdin=subprocess.DEVNULL, timeout=300)
    return time.perf_counter() - t0
t1 = wall(f"{base} -s 2 -i {src} -o /dev/null --frames 1 2>/dev/null")
for rep in range(3):
    tn = wall(f"cat {src} | {base} -s 2 2>/dev/null | cat > /dev/null")
    print(f"-s 2 pipe -> pipe (vmsplice): {(N - 1) / (tn - t1):7.1f} frames/s", flush=True)
# the bytes that arrive through the pipe are the bytes of the file route
wall(f"{base} -s 2 -i {src} -o /dev/shm/uva_file.bgr24 --frames 24 2>/dev/null")
wall(f"head -c {24 * 1080 * 1920 * 3} {src} | {base} -s 2 2>/dev/null > /dev/shm/uva_pipe.bgr24")
a, b = open("/dev/shm/uva_file.bgr24", "rb").read(), open("/dev/shm/uva_pipe.bgr24", "rb").read()
print("pipe route bytes == file route bytes:", a == b, len(a), len(b))
wall(f"head -c {24 * 1080 * 1920 * 3} {src} | {base} -s 2 2>/dev/null | cat > /dev/shm/uva_pipe2.bgr24")
c = open("/dev/shm/uva_pipe2.bgr24", "rb").read()
print("pipe -> pipe (vmsplice into cat) bytes == file route bytes:", a == c, len(c))
for f in (src, "/dev/shm/uva_file.bgr24", "/dev/shm/uva_pipe.bgr24", "/dev/shm/uva_pipe2.bgr24"):
    os.remove(f)
PY
cat $O/rawvideo_pipe.txt
