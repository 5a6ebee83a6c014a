#!/bin/bash
# Run ON THE MI355X BOX: the new default (no packed fp32, consumers issue the DMA, consumers' first fragments in front of barrier 1)
# against the round's final set's library (prev), a parameter sweep around it, and the whole GPU suite
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab15; mkdir -p $O
U=upscale_video_amd
timeout 600 python tools/lib_identity.py $U/libuva_prev.so $U/libuva.so > $O/identity.txt 2>&1; cat $O/identity.txt
bash tools/ab_libs.sh "prev main pffa8 pffa16 pffb4 pffb7 raw0 raw8 noprio pre3" 2 > $O/ab_trunkw.txt 2>&1
bash tools/ab_libs.sh "prev main" 2 "1x_hurrdeblur_1080p 4x_compact_1080p chain_1x_2x_1080p 2x_compact_2160p" > $O/ab_all.txt 2>&1
UVA_LIB_PATH=$PWD/$U/libuva_instr.so python tools/trunkw_anatomy.py > $O/anatomy_main.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -n 3 $O/gpu_tests.txt
cat $O/ab_trunkw.txt $O/ab_all.txt $O/anatomy_main.txt
