#!/bin/bash
# Round 5, GPU call 9 (ON THE BOX): trunkw_kernel with one-directional step counters in place of its two workgroup barriers per iteration
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab8; mkdir -p $O
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_flags.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not huge and not too_large" 2>&1 | tail -n 3 > $O/parity_flags.txt
cat $O/parity_flags.txt
bash tools/ab_libs.sh "main flags" 3 > $O/ab_trunkw.txt 2>&1
cat $O/ab_trunkw.txt
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_flags.so timeout 600 python tools/soak.py 1500 > $O/soak_flags.txt 2>&1; tail -n 4 $O/soak_flags.txt
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr_flags.so timeout 300 python tools/trunkw_anatomy.py > $O/anatomy_flags.txt 2>&1; cat $O/anatomy_flags.txt
