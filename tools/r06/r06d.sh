#!/bin/bash
# Run ON THE MI355X BOX (gpurun): g_conv3_sww (a dense block's 192 -> 64 convolution as Winograd F(2,3)) -- parity tests, then
# the Valar frame with the new kernel and with the direct one (UVA_GENERIC_WINO=0) in turn, three times, and bench.py's line.
exec < /dev/null
export UVA_DEBUG_SWITCHES=1
O=gpurun_out/${1:-r06d}; mkdir -p $O
timeout 1200 python -m pytest tests/test_generic_graph.py -m gpu -x -q > $O/generic_tests.txt 2>&1; tail -6 $O/generic_tests.txt
for i in 1 2 3; do
  for v in 1 0; do echo -n "UVA_GENERIC_WINO=$v: "; UVA_GENERIC_WINO=$v timeout 300 python tools/valar_bench.py 5 2>/dev/null | head -1; done
done > $O/valar_ab.txt 2>&1; cat $O/valar_ab.txt
for v in 1 0; do UVA_GENERIC_WINO=$v timeout 600 python bench.py --workload 4x_valar_1080p --steps 12 --warmup 2 --no-cpu-baseline > $O/bench_valar_wino$v.json 2> $O/bench_valar_wino$v.err; python -c "
import json; d=json.loads(open('$O/bench_valar_wino$v.json').read().strip().splitlines()[-1]); print('UVA_GENERIC_WINO=$v', d['value'], d['config']['kernel_ms_per_frame'], d['roofline']['frac'])"; done
