#!/bin/bash
# libuva_<name>.so = the current objects of libuva.so with ONE translation unit recompiled under extra defines (seconds):
#   tools/sww_variant.sh name uva_sww.hip -DSWW_DBG=2 ...      (select the result with UVA_LIB_PATH)
set -e
NAME=$1; SRC=$2; shift 2
R="$(cd "$(dirname "$0")/.." && pwd)"
C=$R/upscale_video_amd/csrc
O=$C/_obj
mkdir -p /tmp/uva_var
BASE=$(basename $SRC); BASE=${BASE%.*}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $C/$SRC -o /tmp/uva_var/${BASE}_$NAME.o
OBJS=""
for o in $O/*.o; do [ "$(basename $o)" = "$BASE.o" ] && OBJS="$OBJS /tmp/uva_var/${BASE}_$NAME.o" || OBJS="$OBJS $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/upscale_video_amd/libuva_$NAME.so
echo $R/upscale_video_amd/libuva_$NAME.so
