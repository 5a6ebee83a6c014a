#!/bin/bash
# libuva_<name>.so = the current objects of libuva.so (or, with INSTR=1, of libuva_instr.so) with uva_wino.hip recompiled under extra defines:
#   tools/wino_variant.sh name -DTW_PRIO_K=0 ...        (seconds; select the result with UVA_LIB_PATH)
# (any other translation unit: tools/sww_variant.sh name uva_sub10.hip -D...)
set -e
NAME=$1; shift
R="$(cd "$(dirname "$0")/.." && pwd)"
C=$R/upscale_video_amd/csrc
O=$C/_obj${INSTR:+_instr}
mkdir -p /tmp/uva_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${INSTR:+-DUVA_INSTRUMENT} "$@" -c $C/uva_wino.hip -o /tmp/uva_var/uva_wino_$NAME.o
OBJS=""
for o in $O/*.o; do [ "$(basename $o)" = "uva_wino.o" ] && OBJS="$OBJS /tmp/uva_var/uva_wino_$NAME.o" || OBJS="$OBJS $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/upscale_video_amd/libuva_$NAME.so
echo $R/upscale_video_amd/libuva_$NAME.so
