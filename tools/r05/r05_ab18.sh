#!/bin/bash
# Run ON THE MI355X BOX: the 1x net's last layer with channel j in MFMA row 4j (one byte store per fragment, three lane groups)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab18; mkdir -p $O
U=upscale_video_amd
timeout 900 python -m pytest tests/test_gpu_sub5.py tests/test_gpu_parity.py -m gpu -x -q -k "sub5 or 1x or chain or golden or sub10 or hurr or random_geometries or whole_frame" > $O/tests_1x.txt 2>&1; tail -n 3 $O/tests_1x.txt
bash tools/ab_libs.sh "prev main" 3 "1x_hurrdeblur_1080p chain_1x_2x_1080p" > $O/ab_1x.txt 2>&1
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["config"]["kernel_ms_per_frame"], d["roofline"]["frac"], d.get("parity", {}).get("psnr_db"))'
for r in 1 2; do for v in prev main; do
  L=$PWD/$U/libuva_$v.so; [ $v = main ] && L=$PWD/$U/libuva.so
  echo -n "1x sub5 $v: "; UVA_SUB5=1 UVA_LIB_PATH=$L python bench.py --workload 1x_hurrdeblur_1080p --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "$P"
done; done >> $O/ab_1x.txt 2>&1
UVA_LIB_PATH=$PWD/$U/libuva_instr.so python tools/sub10_anatomy.py > $O/sub10_anatomy.txt 2>&1
cat $O/ab_1x.txt $O/sub10_anatomy.txt
