#!/bin/bash
# Run ON THE MI355X BOX: what bounds a dense block's last convolution (conv5, 192 -> 64)?  g_conv3_sww's timing variants
# (tools/sww_variant.sh name uva_sww.hip -DSWW_DBG=k: wrong results) against the whole kernel and the direct g_conv3_sw<6,1>.
exec < /dev/null
export UVA_DEBUG_SWITCHES=1
O=gpurun_out/${1:-r06e}; mkdir -p $O
timeout 900 python -m pytest tests/test_generic_graph.py -m gpu -x -q -k "winograd or independent_fixture or long_segments" > $O/generic_tests.txt 2>&1; tail -3 $O/generic_tests.txt
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["config"]["kernel_ms_per_frame"]; print("frames/s %.2f  rdb4 %.2f ms  conv5 %.2f ms per frame (69 launches each)" % (d["value"], k["rdb4_kernel"], list(k.values())[1]))'
run() { echo -n "$1: "; shift; env "$@" timeout 300 python bench.py --workload 4x_valar_1080p --steps 6 --warmup 1 --repeats 1 --no-cpu-baseline --no-parity 2>/dev/null | python -c "$P"; }
L=$PWD/upscale_video_amd
{
run "g_conv3_sw<6,1> direct (UVA_GENERIC_WINO=0)      " UVA_GENERIC_WINO=0
run "g_conv3_sww Winograd, whole                      " UVA_GENERIC_WINO=1
run "  1: no residual loads, no stores                " UVA_LIB_PATH=$L/libuva_swwdbg1.so
run "  2: no row DMA behind a segment's first rows    " UVA_LIB_PATH=$L/libuva_swwdbg2.so
run "  3: 1 + 2: no memory operation in the loop      " UVA_LIB_PATH=$L/libuva_swwdbg3.so
run " 16: no fragment reads                           " UVA_LIB_PATH=$L/libuva_swwdbg16.so
run " 27: MFMAs and epilogue arithmetic alone         " UVA_LIB_PATH=$L/libuva_swwdbg27.so
run "  4: no MFMAs, no transform (memory + barriers)  " UVA_LIB_PATH=$L/libuva_swwdbg4.so
run "g_conv3_sw<6,1> direct (again)                   " UVA_GENERIC_WINO=0
run "g_conv3_sww Winograd, whole (again)              " UVA_GENERIC_WINO=1
} > $O/conv5_bound.txt 2>&1
cat $O/conv5_bound.txt
