for cfg in "0 4 12" "0 6 10" "0 4 16" "0,0 4 8" "0,0 4 12" "0,0 6 10" "0,0,0,0 2 4" "0,0,0,0 3 6" "0,0,0 4 8"; do
  set -- $cfg
  UVA_GPU_PNG=1 UVA_ENCODE_THREADS=$2 UVA_DECODE_THREADS=$3 python tools/png_route_sweep.py --one $1 360 2>&1 | grep frames/s
done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
