#!/bin/bash
# Run ON THE MI355X BOX: the consumers' DMA (default) against the producers' (TW_DMA_B=0, the path of rounds 4-5a) on 300 random geometries,
# the GPU suite, a longer soak
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab20; mkdir -p $O
U=upscale_video_amd
UVA_IDENTITY_REPS=1 UVA_IDENTITY_RANDOM=300 timeout 1500 python tools/lib_identity.py $U/libuva_dmab0.so $U/libuva.so > $O/identity_random.txt 2>&1; cat $O/identity_random.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -n 2 $O/gpu_tests.txt
timeout 900 python tools/soak.py 6000 20 > $O/soak.txt 2>&1; tail -n 5 $O/soak.txt
