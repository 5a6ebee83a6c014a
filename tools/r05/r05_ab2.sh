#!/bin/bash
# Round 5, GPU call 2 (ON THE BOX): trunkw_kernel variants, second batch: raw rows inside the consumers' k-loop (GAP 14 / STRIDE 3),
# + the k-loops' first fragments read in front of the barrier (PRE_BAR), + one / two rows of the consumers' epilogue in-stream,
# and the 2-D Winograd ceiling with dummy work that leaves the data alone.
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab2; mkdir -p $O
for v in g14_pre g14_inb1_pre g14_inb1; do
  UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not huge and not too_large" 2>&1 | tail -4 > $O/parity_$v.txt
done
bash tools/ab_libs.sh "main g14 g14_pre g14_inb1 g14_inb1_pre g14_inb2_pre g17 pre exp2d1 exp2d2 exp2d3" 3 > $O/ab_trunkw.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr_g14_inb1_pre.so python tools/trunkw_anatomy.py > $O/anatomy_g14_inb1_pre.txt 2>&1
cat $O/parity_*.txt; cat $O/ab_trunkw.txt
