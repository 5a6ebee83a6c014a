cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_generic_graph.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "valar" 2>&1 | tail -3
for i in 1 2; do python tools/valar_bench.py 6 2>&1 | grep frames; done
