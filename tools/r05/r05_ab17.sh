#!/bin/bash
# Run ON THE MI355X BOX: the producers finish rows 2 and 3 of a block in front of their NEXT k-loop (TW_DEFER_A=2)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab17; mkdir -p $O
U=upscale_video_amd
timeout 900 python tools/lib_identity.py $U/libuva.so $U/libuva_defer1.so > $O/identity.txt 2>&1; cat $O/identity.txt
bash tools/ab_libs.sh "main defer1" 3 > $O/ab_trunkw.txt 2>&1
UVA_LIB_PATH=$PWD/$U/libuva_defer1_instr.so python tools/trunkw_anatomy.py > $O/anatomy_defer1.txt 2>&1
UVA_LIB_PATH=$PWD/$U/libuva_defer1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_order.py -m gpu -x -q > $O/parity_defer1.txt 2>&1; tail -n 3 $O/parity_defer1.txt
cat $O/ab_trunkw.txt $O/anatomy_defer1.txt
