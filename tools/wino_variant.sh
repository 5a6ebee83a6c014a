#!/bin/bash
# libuva_<name>.so = the current objects of libuva.so (or, with INSTR=1, of libuva_instr.so) with uva_wino.hip recompiled under extra defines:
#   tools/wino_variant.sh name -DTW_PRIO_K=0 ...        (seconds; select the result with UVA_LIB_PATH)
set -e
NAME=$1; shift
R="$(cd "$(dirname "$0")/.." && pwd)"
C=$R/upscale_video_amd/csrc
O=$C/_obj${INSTR:+_instr}
mkdir -p /tmp/uva_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${INSTR:+-DUVA_INSTRUMENT} "$@" -c $C/uva_wino.hip -o /tmp/uva_var/uva_wino_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/uva_api.o /tmp/uva_var/uva_wino_$NAME.o $O/uva_sub5.o $O/uva_model.o $O/uva_generic.o $O/uva_pngread.o -o $R/upscale_video_amd/libuva_$NAME.so
echo $R/upscale_video_amd/libuva_$NAME.so
