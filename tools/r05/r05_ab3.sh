#!/bin/bash
# Round 5, GPU call 3 (ON THE BOX): sub5_kernel's first run (parity against sub10_kernel and the oracle, then the bench A/B), and
# trunkw_kernel with deeper read-ahead / the per-group pre-barrier prefetch.
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sub5.py -m gpu -x -q 2>&1 | tail -25 > $O/sub5_tests.txt
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["kernel"][:12], d.get("parity", {}).get("psnr_db"))'
for r in 1 2 3; do
  for s5 in 0 1; do
    echo -n "1x_hurrdeblur_1080p UVA_SUB5=$s5: "; UVA_SUB5=$s5 python bench.py --workload 1x_hurrdeblur_1080p --steps 300 --warmup 30 --no-cpu-baseline 2>>$O/bench.err | python -c "$P"
  done
done > $O/ab_sub5.txt 2>&1
for s5 in 0 1; do echo -n "chain UVA_SUB5=$s5: "; UVA_SUB5=$s5 python bench.py --workload chain_1x_2x_1080p --steps 200 --warmup 20 --no-cpu-baseline 2>>$O/bench.err | python -c "$P"; done >> $O/ab_sub5.txt 2>&1
bash tools/ab_libs.sh "main r4 pffA8 pffA10 pffA12 pffA10_B7 pffB4 pre1 pre2 pre3" 3 > $O/ab_trunkw.txt 2>&1
for v in pffA10 pre2; do
  UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not huge and not too_large" 2>&1 | tail -2 > $O/parity_$v.txt
done
cat $O/sub5_tests.txt $O/ab_sub5.txt $O/parity_*.txt $O/ab_trunkw.txt
