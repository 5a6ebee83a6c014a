#!/bin/bash
# Same-box A/B (A = HEAD, B = working tree) preceded by the GPU test-suite on B, all in one gpurun call.
# usage: tools/ab_test.sh [pytest-args]   (WORKLOAD, REPS env)
cd "$(dirname "$0")/.."
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-variable"
python -c "from upscale_video_amd import build; build.build_lib(force=True)" || exit 1
cp upscale_video_amd/libuva.so upscale_video_amd/libuva_B.so
# A: HEAD's sources, extracted beside the working tree (no stash: the tree may be edited meanwhile)
rm -rf /tmp/uva_A && mkdir -p /tmp/uva_A && git archive ${BASE:-HEAD} upscale_video_amd/csrc include | tar -x -C /tmp/uva_A
$HIPCC /tmp/uva_A/upscale_video_amd/csrc/uva_api.hip /tmp/uva_A/upscale_video_amd/csrc/uva_model.cpp $(ls /tmp/uva_A/upscale_video_amd/csrc/uva_generic.cpp 2>/dev/null) -o upscale_video_amd/libuva_A.so 2>&1 | grep error
touch upscale_video_amd/libuva.so
PYT="${1:--m gpu -x -q}"
/usr/local/graft/bin/gpurun --timeout ${TIMEOUT:-1500} -- 'mkdir -p gpurun_out; timeout 1200 python -m pytest tests '"$PYT"' 2>&1 | tail -15; P="import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"config\"][\"kernel_ms_per_frame\"], d[\"roofline\"][\"frac\"])"; for i in $(seq 1 '"${REPS:-3}"'); do for v in A B; do echo -n "$v: "; UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_$v.so python bench.py --workload '"${WORKLOAD:-2x_compact_1080p}"' --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "$P"; done; done' 2>&1 | tail -40
rm -f upscale_video_amd/libuva_A.so upscale_video_amd/libuva_B.so
