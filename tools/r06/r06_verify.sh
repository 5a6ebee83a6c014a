mkdir -p gpurun_out/r06_verify; O=gpurun_out/r06_verify
(time timeout 2400 python -m pytest tests -m gpu -q) > $O/gpu_tests.txt 2>&1; tail -25 $O/gpu_tests.txt
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-900
python bench.py --workload 1x_hurrdeblur_1080p > $O/bench_1x.json 2>> $O/bench.err; cat $O/bench_1x.json | cut -c1-600
