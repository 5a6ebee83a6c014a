#!/bin/bash
# Run ON THE MI355X BOX: the consumers issue the raw rows' LDS-DMA (TW_DMA_B)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab14; mkdir -p $O
U=upscale_video_amd
timeout 600 python tools/lib_identity.py $U/libuva.so $U/libuva_nopk.so $U/libuva_dmab.so $U/libuva_dmab_pre2.so > $O/identity.txt 2>&1; cat $O/identity.txt
bash tools/ab_libs.sh "main nopk nopk_pre2 dmab dmab_pre2" 3 > $O/ab_trunkw.txt 2>&1
UVA_LIB_PATH=$PWD/$U/libuva_dmab_instr.so python tools/trunkw_anatomy.py > $O/anatomy_dmab.txt 2>&1
UVA_LIB_PATH=$PWD/$U/libuva_dmab.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_order.py -m gpu -x -q > $O/parity_dmab.txt 2>&1; tail -n 3 $O/parity_dmab.txt
UVA_LIB_PATH=$PWD/$U/libuva_dmab.so timeout 600 python tools/soak.py 1500 10 > $O/soak_dmab.txt 2>&1; tail -n 6 $O/soak_dmab.txt
cat $O/ab_trunkw.txt $O/anatomy_dmab.txt
