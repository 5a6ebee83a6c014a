#!/bin/bash
# Run ON THE MI355X BOX: the final set's box read 3 075 frames/s for the 1x net where blocks 19/20 read 3 448 -- same-box A/B of the
# old behaviour (-DS10_BAL=0 -DS10_ROWSKIP=0) against the default, with the package power and clock while each runs
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab25; mkdir -p $O
U=upscale_video_amd
bash tools/ab_libs.sh "old10 main" 4 "1x_hurrdeblur_1080p" > $O/ab_1x.txt 2>&1
for v in old10 main; do
  L=$PWD/$U/libuva_$v.so; [ $v = main ] && L=$PWD/$U/libuva.so
  UVA_LIB_PATH=$L python bench.py --workload 1x_hurrdeblur_1080p --steps 40000 --warmup 100 --no-cpu-baseline --no-parity > $O/long_$v.json 2>/dev/null &
  sleep 7
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; sleep 1; done > $O/power_$v.txt
  wait
  python -c "import json; d=json.loads(open('$O/long_$v.json').read().strip().splitlines()[-1]); print('$v 40000 steps:', d['value'], d['roofline']['frac'])" >> $O/ab_1x.txt
done
bash tools/ab_libs.sh "old10 main" 2 "1x_hurrdeblur_1080p chain_1x_2x_1080p 2x_compact_1080p" >> $O/ab_1x.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_sub5.py -m gpu -x -q > $O/tests_sub5.txt 2>&1; tail -n 2 $O/tests_sub5.txt
cat $O/ab_1x.txt; grep -h -E "Power|sclk" $O/power_old10.txt | head -4; echo; grep -h -E "Power|sclk" $O/power_main.txt | head -4
