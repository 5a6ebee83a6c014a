#!/bin/bash
# trunkw_kernel alone -> /tmp/wk/wk.s in seconds (the library takes minutes):  tools/wino_isa.sh [-DFOO]
cd /tmp/wk && time /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function "$@" --cuda-device-only -S wk.hip -o wk.s 2>&1 | grep -v "hip-link" ; python /root/repo/tools/isa_stats.py /tmp/wk/wk.s trunkw_kernel
