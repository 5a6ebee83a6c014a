#!/bin/bash
# Same-box A/B: A = HEAD, B = the working tree.  usage: tools/ab_head.sh [label]
cd "$(dirname "$0")/.."
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-variable"
$HIPCC upscale_video_amd/csrc/uva_api.hip upscale_video_amd/csrc/uva_model.cpp upscale_video_amd/csrc/uva_generic.cpp -o upscale_video_amd/libuva_B.so 2>&1 | grep error
git stash -q || exit 1
$HIPCC upscale_video_amd/csrc/uva_api.hip upscale_video_amd/csrc/uva_model.cpp upscale_video_amd/csrc/uva_generic.cpp -o upscale_video_amd/libuva_A.so 2>&1 | grep error
git stash pop -q
echo "== A (HEAD) vs B (working tree): $1"
/usr/local/graft/bin/gpurun --timeout 400 -- 'P="import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"config\"][\"kernel_ms_per_frame\"])"; for i in 1 2 3; do for v in A B; do echo -n "$v: "; UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_$v.so python bench.py --workload '"${WORKLOAD:-2x_compact_1080p}"' --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P"; done; done' 2>&1 | grep -E "^A:|^B:"
rm -f upscale_video_amd/libuva_A.so upscale_video_amd/libuva_B.so
