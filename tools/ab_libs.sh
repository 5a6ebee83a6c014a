#!/bin/bash
# Run ON THE MI355X BOX: same-box comparison of several builds of the library shipped side by side
# (upscale_video_amd/libuva_<name>.so; "main" = libuva.so).   tools/ab_libs.sh "prev pin8 main" [rounds] [workloads]
cd "$(dirname "$0")/.."
LIBS=${1:-"prev main"}; ROUNDS=${2:-2}; WLS=${3:-"2x_compact_1080p"}
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["config"]["kernel_ms_per_frame"], d["roofline"]["frac"], d.get("parity", {}).get("psnr_db"))'
for wl in $WLS; do
  for r in $(seq 1 $ROUNDS); do
    for v in $LIBS; do
      L=$PWD/upscale_video_amd/libuva_$v.so; [ $v = main ] && L=$PWD/upscale_video_amd/libuva.so
      echo -n "$wl $v: "; UVA_LIB_PATH=$L python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "$P"
    done
  done
done
