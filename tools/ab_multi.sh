#!/bin/bash
# Same-box comparison of several one-line variants of uva_kernels.hip.h (V0 = the working tree as is,
# V1.. = one sed expression each).  usage: tools/ab_multi.sh "<sed expr 1>" "<sed expr 2>" ...
cd /root/repo
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-variable"
F=upscale_video_amd/csrc/uva_kernels.hip.h
cp $F /tmp/ab_multi.backup
$HIPCC upscale_video_amd/csrc/uva_api.hip upscale_video_amd/csrc/uva_model.cpp upscale_video_amd/csrc/uva_generic.cpp -o upscale_video_amd/libuva_V0.so 2>&1 | grep error
i=1
for e in "$@"; do
  sed -i "$e" $F
  $HIPCC upscale_video_amd/csrc/uva_api.hip upscale_video_amd/csrc/uva_model.cpp upscale_video_amd/csrc/uva_generic.cpp -o upscale_video_amd/libuva_V$i.so 2>&1 | grep error
  cp /tmp/ab_multi.backup $F
  i=$((i+1))
done
touch upscale_video_amd/libuva.so
N=$((i-1))
/usr/local/graft/bin/gpurun --timeout 600 -- "P='import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"config\"][\"kernel_ms_per_frame\"])'; for r in 1 2; do for v in \$(seq 0 $N); do echo -n \"V\$v: \"; UVA_LIB_PATH=\$PWD/upscale_video_amd/libuva_V\$v.so python bench.py --workload ${WORKLOAD:-2x_compact_1080p} --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c \"\$P\"; done; done" 2>&1 | grep -E "^V[0-9]"
rm -f upscale_video_amd/libuva_V*.so
