#!/bin/bash
# trunkw_kernel alone -> /tmp/wk/wk.s in seconds (the library takes minutes):  tools/wino_isa.sh [-DFOO] [-DWK_ACT=0|1|2]
mkdir -p /tmp/wk
cat > /tmp/wk/wk.hip <<'EOT'
#include "/root/repo/upscale_video_amd/csrc/uva_wino.hip.h"
#ifndef WK_ACT
#define WK_ACT 1
#endif
template __global__ void uva::trunkw_kernel<64, WK_ACT>(uva::TrunkwArgs);
EOT
cd /tmp/wk && time /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function "$@" --cuda-device-only -S wk.hip -o wk.s 2>&1 | grep -v "hip-link" ; python /root/repo/tools/isa_stats.py /tmp/wk/wk.s trunkw_kernel
