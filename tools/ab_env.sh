#!/bin/bash
# Same-box comparison of run-time variants of ONE build: each argument is an environment assignment list
# (quote it), e.g.  tools/ab_env.sh "UVA_TRUNK_FUSION=0" "UVA_TRUNK_FUSION=1".   WORKLOAD, REPS env.
cd "$(dirname "$0")/.."
python -c "from upscale_video_amd import build; build.build_lib()" || exit 1
CMD='P="import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"config\"][\"kernel_ms_per_frame\"], d[\"roofline\"][\"frac\"])"; for i in $(seq 1 '"${REPS:-3}"'); do '
for v in "$@"; do
  CMD+='echo -n "'"$v"': "; env '"$v"' python bench.py --workload '"${WORKLOAD:-2x_compact_1080p}"' --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "$P"; '
done
CMD+='done'
/usr/local/graft/bin/gpurun --timeout ${TIMEOUT:-900} -- "$CMD" 2>&1 | tail -30
