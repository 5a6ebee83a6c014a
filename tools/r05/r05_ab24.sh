#!/bin/bash
# Run ON THE MI355X BOX: sub10_kernel's waves skip the rows nobody reads (S10_ROWSKIP) on top of block 19's build (libuva_bal.so)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab24; mkdir -p $O
U=upscale_video_amd
UVA_IDENTITY_1X=1 UVA_IDENTITY_RANDOM=60 timeout 900 python tools/lib_identity.py $U/libuva_prev.so $U/libuva.so > $O/identity_1x.txt 2>&1; cat $O/identity_1x.txt
timeout 900 python -m pytest tests/test_gpu_sub5.py tests/test_gpu_parity.py -m gpu -x -q -k "sub5 or 1x or chain or golden or sub10 or hurr or random_geometries or whole_frame" > $O/tests_1x.txt 2>&1; tail -n 3 $O/tests_1x.txt
bash tools/ab_libs.sh "prev bal main" 3 "1x_hurrdeblur_1080p" > $O/ab_1x.txt 2>&1
bash tools/ab_libs.sh "bal main" 2 "chain_1x_2x_1080p" >> $O/ab_1x.txt 2>&1
UVA_LIB_PATH=$PWD/$U/libuva_instr.so python tools/sub10_anatomy.py > $O/sub10_anatomy.txt 2>&1
cat $O/ab_1x.txt $O/sub10_anatomy.txt
