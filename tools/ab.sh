#!/bin/bash
# Same-box A/B of a kernel variant (box-to-box variance is ~3 %, so variants are only comparable
# within one gpurun call).  usage: tools/ab.sh "<python statements editing the source text s>" label
cd /root/repo
python upscale_video_amd/build.py > /dev/null
cp upscale_video_amd/csrc/uva_kernels.hip.h /tmp/uva_kernels.backup
python - <<PY
p='upscale_video_amd/csrc/uva_kernels.hip.h'
s=open(p).read()
$1
open(p,'w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function upscale_video_amd/csrc/uva_api.hip upscale_video_amd/csrc/uva_model.cpp -o upscale_video_amd/libuva_B.so 2>&1 | grep error
cp /tmp/uva_kernels.backup upscale_video_amd/csrc/uva_kernels.hip.h
touch upscale_video_amd/libuva.so
echo "== A/B: $2"
/usr/local/graft/bin/gpurun --timeout 400 -- 'P="import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"config\"][\"kernel_ms_per_frame\"][\"trunk\"])"; for i in 1 2 3; do echo -n "A: "; python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P"; echo -n "B: "; UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_B.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P"; done' 2>&1 | grep -E "^A:|^B:"
rm -f upscale_video_amd/libuva_B.so
