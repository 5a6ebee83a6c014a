#!/bin/bash
# Run ON THE MI355X BOX: sub10_kernel's wave map once more (S10_MAP=1: the head's three-fragment half on wave 11; 2: the four-fragment
# layers on the older waves of their SIMDs), same bytes required
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab27; mkdir -p $O
U=upscale_video_amd
UVA_IDENTITY_1X=1 UVA_IDENTITY_REPS=1 timeout 600 python tools/lib_identity.py $U/libuva.so $U/libuva_map1.so $U/libuva_map2.so $U/libuva_old10.so > $O/identity_1x.txt 2>&1; cat $O/identity_1x.txt
bash tools/ab_libs.sh "main map1 map2" 4 "1x_hurrdeblur_1080p" > $O/ab_1x.txt 2>&1
cat $O/ab_1x.txt
