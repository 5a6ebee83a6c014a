#!/bin/bash
# Run ON THE MI355X BOX: the consumers' step entry requested at the top of the iteration, not between their last store and barrier 1
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab22; mkdir -p $O
U=upscale_video_amd
timeout 900 python tools/lib_identity.py $U/libuva.so $U/libuva_ek.so > $O/identity.txt 2>&1; cat $O/identity.txt
bash tools/ab_libs.sh "main ek" 4 > $O/ab_trunkw.txt 2>&1
UVA_LIB_PATH=$PWD/$U/libuva_ek_instr.so python tools/trunkw_anatomy.py > $O/anatomy_ek.txt 2>&1
cat $O/ab_trunkw.txt $O/anatomy_ek.txt
