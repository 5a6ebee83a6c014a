#!/bin/bash
# Run ON THE MI355X BOX: no v_pk_*_f32 (a) in every kernel of the library, other workloads; (b) trunkw variants on top of it
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab12; mkdir -p $O
bash tools/ab_libs.sh "main nopk nopk_ina2 nopk_inb1 nopk_pre1 nopk_pre2" 2 > $O/ab_trunkw.txt 2>&1
bash tools/ab_libs.sh "main nopkall" 2 "1x_hurrdeblur_1080p 4x_compact_1080p chain_1x_2x_1080p" > $O/ab_all.txt 2>&1
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["config"]["kernel_ms_per_frame"], d["roofline"]["frac"], d.get("parity", {}).get("psnr_db"))'
for v in main nopkall; do
  L=$PWD/upscale_video_amd/libuva_$v.so; [ $v = main ] && L=$PWD/upscale_video_amd/libuva.so
  echo -n "4x_valar_1080p $v: "; UVA_LIB_PATH=$L python bench.py --workload 4x_valar_1080p --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P"
done >> $O/ab_all.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_nopkall.so timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests_nopkall.txt 2>&1; tail -n 3 $O/gpu_tests_nopkall.txt
cat $O/ab_trunkw.txt $O/ab_all.txt
