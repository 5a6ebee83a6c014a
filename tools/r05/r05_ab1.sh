#!/bin/bash
# Round 5, GPU call 1 (run ON THE BOX): the -m gpu suite on the current build, then same-box A/B of the trunkw_kernel variants
# built beside it (tools/wino_variant.sh): who transforms the raw rows WHERE, and the 2-D Winograd ceiling.
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ab1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
cp gpurun_out/parity_report.json $O/ 2>/dev/null
# parity of the candidate variants (the fp32 + product-mode bars of the parity file, the golden fixtures, random geometries)
for v in rawink rawink_inb2; do
  UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not huge and not too_large" > $O/parity_$v.txt 2>&1; echo "rc $?" >> $O/parity_$v.txt
done
bash tools/ab_libs.sh "main rawink rawink_inb2 inb2 rawink_g14 exp2d1 exp2d2" 3 > $O/ab_trunkw.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr.so python tools/trunkw_anatomy.py > $O/anatomy_main.txt 2>&1
UVA_LIB_PATH=$PWD/upscale_video_amd/libuva_instr_rawink.so python tools/trunkw_anatomy.py > $O/anatomy_rawink.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2>> $O/bench_default.err
tail -3 $O/gpu_tests.txt; cat $O/ab_trunkw.txt
