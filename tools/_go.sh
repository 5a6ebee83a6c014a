cd /root/repo
timeout 900 python -m pytest tests/test_generic_graph.py -x -q -m gpu 2>&1 | tail -3
for v in 1 0 1 0; do echo "fuse $v: $(UVA_GENERIC_FUSE_ADD=$v timeout 300 python tools/valar_bench.py 3 2>&1 | grep frames)"; done
