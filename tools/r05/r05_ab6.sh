#!/bin/bash
# Round 5, GPU call 7 (ON THE BOX): where the producers issue their LDS-DMA (in front of / behind / inside their k-loop), raw reads at fragment 4.
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab6; mkdir -p $O
bash tools/ab_libs.sh "main dmalate1 dmalate2 dmalate2_pff10 rawf4" 3 > $O/ab_trunkw.txt 2>&1
for v in dmalate1 dmalate2_pff10; do
  UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not huge and not too_large" 2>&1 | tail -n 2 > $O/parity_$v.txt
done
cat $O/ab_trunkw.txt $O/parity_*.txt
