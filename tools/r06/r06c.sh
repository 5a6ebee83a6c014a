#!/bin/bash
# Run ON THE MI355X BOX (gpurun): the raw-video pipe route with and without vmsplice, the PNG decode split, the driver's N > 1
# form with one rank (TorchComm + the per-rank records + --dynamic), and what this box lets us do to the shader clock.
O=gpurun_out/r06c; mkdir -p $O
python -c "import torch" 2>/dev/null
python tools/png_decode_split.py 9 > $O/png_decode_split.txt 2>&1; cat $O/png_decode_split.txt
python tools/pipe_bench.py 200 > $O/pipe_bench.txt 2>&1; cat $O/pipe_bench.txt
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
    subprocess.run(cmd, shell=True, check=True, env=dict(os.environ, **(env or {})))
    return time.perf_counter() - t0
t1 = wall(f"{base} -s 2 -i {src} -o /dev/null --frames 1 2>/dev/null")
for label, env in (("vmsplice (default)", {}), ("write() (UVA_RAW_VMSPLICE=0)", {"UVA_RAW_VMSPLICE": "0"})):
    for rep in range(3):
        for g in ("0", "0,0"):
            tn = wall(f"cat {src} | {base} -s 2 -g {g} 2>/dev/null | cat > /dev/null", env)
            print(f"-s 2 -g {g:4s} pipe -> pipe, {label:30s}: {N} frames in {tn:6.2f} s = {(N - 1) / (tn - t1):7.1f} frames/s", flush=True)
    tn = wall(f"cat {src} | {base} -s 2 2>/dev/null | dd bs=4M of=/dev/null status=none", env)
    print(f"-s 2         pipe -> dd bs=4M, {label:30s}: {(N - 1) / (tn - t1):7.1f} frames/s", flush=True)
    tn = wall(f"{base} -s 2 -i {src} 2>/dev/null | cat > /dev/null", env)
    print(f"-s 2         file -> pipe,   {label:30s}: {(N - 1) / (tn - t1):7.1f} frames/s", flush=True)
# one output file on tmpfs with 1 / 4 / 8 workers: the workers now feed one writer there
for g in ("0", "0,0,0,0", "0,0,0,0,0,0,0,0"):
    k = len(g.split(","))
    t1g = wall(f"{base} -s 2 -g {g} -i {src} -o /dev/shm/uva_out.bgr24 --frames {k} 2>/dev/null")
    tn = wall(f"{base} -s 2 -g {g} -i {src} -o /dev/shm/uva_out.bgr24 2>/dev/null")
    print(f"-s 2 -g {g:16s} file -> ONE file on tmpfs: {(N - k) / (tn - t1g):7.1f} frames/s", flush=True)
    t1g = wall(f"{base} -s 2 -g {g} -i {src} -o /dev/shm/uva_out.bgr24 --frames {k} 2>/dev/null", {"UVA_RAW_TMPFS_SEGMENTS": "1"})
    tn = wall(f"{base} -s 2 -g {g} -i {src} -o /dev/shm/uva_out.bgr24 2>/dev/null", {"UVA_RAW_TMPFS_SEGMENTS": "1"})
    print(f"-s 2 -g {g:16s} ... segments as before (UVA_RAW_TMPFS_SEGMENTS=1): {(N - k) / (tn - t1g):7.1f} frames/s", flush=True)
os.remove(src); os.remove("/dev/shm/uva_out.bgr24")
PY
cat $O/rawvideo_pipe.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --dynamic --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_torchrun_one_rank_dynamic.json 2> $O/torchrun.err; tail -c 400 $O/torchrun.err; cut -c1-300 $O/bench_torchrun_one_rank_dynamic.json
echo "== clock controls"
(rocm-smi --setperfdeterminism 1700; rocm-smi --showperflevel; rocm-smi --showsclkrange; rocm-smi --setextremum max sclk 1700; amd-smi set --help | head -40) > $O/clock_controls.txt 2>&1; tail -30 $O/clock_controls.txt
python bench.py --workload 1x_hurrdeblur_1080p --batch 1 --steps 20000 --warmup 100 --repeats 1 --no-cpu-baseline --no-parity > $O/run_1x_after_cap.json 2>/dev/null &
sleep 14; rocm-smi --showclocks --showpower | grep -E "sclk|Power"; wait
python -c "import json; d=json.load(open('$O/run_1x_after_cap.json')); print('1x b1 after cap attempts:', d['value'])"
rocm-smi --resetperfdeterminism > /dev/null 2>&1; rocm-smi --resetclocks > /dev/null 2>&1
