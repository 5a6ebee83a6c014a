#!/bin/bash
# Run ON THE MI355X BOX: what bounds phase X of trunkw_kernel now (ceiling experiments: nobody waits for the raw rows' DMA / no DMA
# at all), the instruction kinds left out of the first co-issue table, and whether vmcnt counts a wave's loads and stores in order
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab13; mkdir -p $O
timeout 120 tools/scratch/vmorder > $O/vmorder.txt 2>&1
timeout 300 tools/scratch/coissue 256 > $O/coissue.txt 2>&1
bash tools/ab_libs.sh "main nopk nopk_pre2 abl_novm abl_nodma" 2 > $O/ab_trunkw.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_nopk_pre2_instr.so python tools/trunkw_anatomy.py > $O/anatomy_nopk_pre2.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_abl_novm_instr.so python tools/trunkw_anatomy.py > $O/anatomy_abl_novm.txt 2>&1
cat $O/vmorder.txt $O/ab_trunkw.txt $O/anatomy_nopk_pre2.txt $O/anatomy_abl_novm.txt; tail -n 14 $O/coissue.txt
