#!/bin/bash
# Round 5, GPU call 5 (ON THE BOX): the fold with its 12-column limit (byte-for-byte fold on / off test), wave priorities with the
# epilogue ABOVE the k-loop, sub5_kernel with part 1's light layer split 1 / 3, Valar on frames that fit the Infinity Cache, raw video
# with positional writers.
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sub5.py -m gpu -x -q 2>&1 | tail -n 6 > $O/parity_tests.txt
bash tools/ab_libs.sh "main pffA6 prio02 prio13 prio00" 3 > $O/ab_trunkw.txt 2>&1
Q='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["kernel"][:12], d.get("parity", {}).get("psnr_db"))'
for r in 1 2 3; do
  for s5 in 0 1; do
    echo -n "1x_hurrdeblur_1080p UVA_SUB5=$s5: "; UVA_SUB5=$s5 python bench.py --workload 1x_hurrdeblur_1080p --steps 300 --warmup 30 --no-cpu-baseline 2>>$O/bench.err | python -c "$Q"
  done
done > $O/ab_sub5.txt 2>&1
for s5 in 0 1; do echo -n "chain UVA_SUB5=$s5: "; UVA_SUB5=$s5 python bench.py --workload chain_1x_2x_1080p --steps 200 --warmup 20 --no-cpu-baseline 2>>$O/bench.err | python -c "$Q"; done >> $O/ab_sub5.txt 2>&1
UVA_SUB5=1 UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr.so python tools/sub5_anatomy.py > $O/sub5_anatomy.txt 2>&1
for hw in "360 640" "540 960" "720 1280" "1080 1920"; do python tools/valar_bench.py 3 $hw 2>&1 | grep frames; done > $O/valar_sizes.txt 2>&1
python tools/rawvideo_bench.py 400 > $O/rawvideo_bench.txt 2>&1
cat $O/parity_tests.txt $O/ab_trunkw.txt $O/ab_sub5.txt $O/sub5_anatomy.txt $O/valar_sizes.txt $O/rawvideo_bench.txt
