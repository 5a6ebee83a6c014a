#!/bin/bash
# Run ON THE MI355X BOX: round 4's rejected epilogue variants again under the new balance (16-byte ring writes after a lane exchange,
# pinned slice order)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab19; mkdir -p $O
U=upscale_video_amd
timeout 900 python tools/lib_identity.py $U/libuva.so $U/libuva_w128.so $U/libuva_fence.so $U/libuva_pffb8.so > $O/identity.txt 2>&1; cat $O/identity.txt
bash tools/ab_libs.sh "main w128 fence pffb8" 3 > $O/ab_trunkw.txt 2>&1
cat $O/ab_trunkw.txt
