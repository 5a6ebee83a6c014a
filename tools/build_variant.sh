#!/bin/bash
# Builds a variant of libuva.so with extra defines and keeps its ISA:  tools/build_variant.sh NAME [-DFOO=1 ...]
# (trunkw_kernel's object is taken as built: csrc/_obj/uva_wino.o; its own variants: tools/wino_variant.sh)
# -> upscale_video_amd/libuva_NAME.so, /tmp/uva_build/NAME/uva_api-hip-amdgcn-amd-amdhsa-gfx950.s
set -e
NAME=$1; shift
R="$(cd "$(dirname "$0")/.." && pwd)"
D=/tmp/uva_build/$NAME
mkdir -p $D
C=$R/upscale_video_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" \
  $C/uva_api.hip $C/uva_model.cpp $C/uva_generic.cpp $C/uva_pngread.cpp $C/_obj/uva_wino.o $C/_obj/uva_sub5.o -o $D/libuva.so -save-temps=obj 2>&1 | grep -E "error|warning: v|spill" || true
cp $D/libuva.so $R/upscale_video_amd/libuva_$NAME.so
echo "$R/upscale_video_amd/libuva_$NAME.so"
