mkdir -p gpurun_out/r05_verify; O=gpurun_out/r05_verify
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/gpu_tests.txt 2>&1; tail -8 $O/gpu_tests.txt
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-600
