#!/bin/bash
# Run ON THE MI355X BOX: how many of a wave's five DMA pieces the consumers issue (TW_DMA_B = 0 / 2 / 3 / 4 / 5)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab16; mkdir -p $O
U=upscale_video_amd
timeout 900 python tools/lib_identity.py $U/libuva_prev.so $U/libuva.so $U/libuva_dmab0.so $U/libuva_dmab2.so $U/libuva_dmab3.so $U/libuva_dmab4.so > $O/identity.txt 2>&1; cat $O/identity.txt
bash tools/ab_libs.sh "prev dmab0 dmab2 dmab3 dmab4 main" 3 > $O/ab_trunkw.txt 2>&1
cat $O/ab_trunkw.txt
