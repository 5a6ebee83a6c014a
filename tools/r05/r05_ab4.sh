#!/bin/bash
# Round 5, GPU call 4 (ON THE BOX): the -m gpu suite on the build with folded last strips; fold on / off and the deeper producers'
# read-ahead; sub5_kernel with its light layer split between the front and back waves; the dense-block ceiling (Valar); raw video.
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab4; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
cp gpurun_out/parity_report.json $O/ 2>/dev/null
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["config"]["kernel_ms_per_frame"], d["roofline"]["frac"], d.get("parity", {}).get("psnr_db"))'
for r in 1 2 3; do
  for f in 0 1; do
    echo -n "2x_compact_1080p UVA_TW_FOLD=$f: "; UVA_TW_FOLD=$f python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$O/bench.err | python -c "$P"
  done
  echo -n "2x_compact_1080p fold + PFF_A=12: "; UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_pffA12.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$O/bench.err | python -c "$P"
done > $O/ab_fold.txt 2>&1
for wl in 4x_compact_1080p; do for f in 0 1; do echo -n "$wl UVA_TW_FOLD=$f: "; UVA_TW_FOLD=$f python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline 2>>$O/bench.err | python -c "$P"; done; done >> $O/ab_fold.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_pffA12.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not huge and not too_large" 2>&1 | tail -n 2 > $O/parity_pffA12.txt
Q='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["kernel"][:12], d.get("parity", {}).get("psnr_db"))'
timeout 900 python -m pytest tests/test_gpu_sub5.py -m gpu -x -q 2>&1 | tail -n 3 > $O/sub5_tests.txt
for r in 1 2 3; do
  for s5 in 0 1; do
    echo -n "1x_hurrdeblur_1080p UVA_SUB5=$s5: "; UVA_SUB5=$s5 python bench.py --workload 1x_hurrdeblur_1080p --steps 300 --warmup 30 --no-cpu-baseline 2>>$O/bench.err | python -c "$Q"
  done
done > $O/ab_sub5.txt 2>&1
UVA_SUB5=1 UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr.so python tools/sub5_anatomy.py > $O/sub5_anatomy.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr.so python tools/sub10_anatomy.py > $O/sub10_anatomy.txt 2>&1
for r in 1 2; do
  echo -n "main: "; python tools/valar_bench.py 3 2>&1 | grep frames
  echo -n "ceiling (x1..x4 neither stored nor fetched from HBM, WRONG results): "; UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_ceil_rdb.so python tools/valar_bench.py 3 2>&1 | grep frames
done > $O/valar_ceiling.txt 2>&1
python tools/rawvideo_bench.py 400 > $O/rawvideo_bench.txt 2>&1
tail -n 4 $O/gpu_tests.txt; cat $O/ab_fold.txt $O/parity_pffA12.txt $O/sub5_tests.txt $O/ab_sub5.txt $O/sub5_anatomy.txt $O/valar_ceiling.txt $O/rawvideo_bench.txt
