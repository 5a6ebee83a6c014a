#!/bin/bash
# Run ON THE MI355X BOX: trunkw_kernel without v_pk_*_f32 in the epilogues (tools/coissue_bench.hip: a packed-fp32 instruction
# gets no issue slot beside the other wave's MFMAs)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab11; mkdir -p $O
bash tools/ab_libs.sh "main nopk" 3 > $O/ab_trunkw.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr.so python tools/trunkw_anatomy.py > $O/anatomy_main.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_nopk_instr.so python tools/trunkw_anatomy.py > $O/anatomy_nopk.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_nopk.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/parity_nopk.txt 2>&1; tail -n 3 $O/parity_nopk.txt
timeout 600 python -m pytest tests/test_gpu_stream_order.py -m gpu -x -q > $O/stream_order.txt 2>&1; tail -n 3 $O/stream_order.txt
cat $O/ab_trunkw.txt $O/anatomy_main.txt $O/anatomy_nopk.txt
