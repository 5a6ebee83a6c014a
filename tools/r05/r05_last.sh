#!/bin/bash
# Run ON THE MI355X BOX: the round's last build -- GPU suite, smoke, the default bench line in the driver's form, the 1x line
cd "$(dirname "$0")/.."
O=gpurun_out/r05_last; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
(python -c "import __graft_entry__ as g; g.smoke(); print('smoke: ok')" 2>&1 | tail -2) > $O/smoke.txt; cat $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err; cut -c1-400 $O/bench_driver_form.json
python bench.py --workload 1x_hurrdeblur_1080p --steps 100 --warmup 10 > $O/bench_1x.json 2>> $O/bench.err; cut -c1-300 $O/bench_1x.json
python tools/soak.py 1000 > $O/soak.txt 2>&1; tail -5 $O/soak.txt
