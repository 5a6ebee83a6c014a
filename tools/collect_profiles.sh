#!/bin/bash
# Run ON THE MI355X BOX (through gpurun): the rocprofv3 evidence behind bench.py's roofline object.
#   pass 1: --kernel-trace --stats of the default bench command       -> <tag>_kernel_stats_rocprofv3.csv
#   pass 2-4: PMC counters, each in its own run with --kernel-trace only (never with hip/hsa traces)
#   then tools/summarize_pmc.py                                       -> <tag>_trunk_pmc.json
# usage: tools/collect_profiles.sh r01_d      (outputs under gpurun_out/profiles_<tag>/)
set -u
TAG=${1:-r01_x}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-parity"     # no 128x96 probe launches: every row of the CSVs is a full-size launch
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- $B --steps 120 --warmup 10 > "$OUT/stats.log" 2>&1
cp "$OUT"/stats/p_kernel_stats.csv "$OUT/${TAG}_kernel_stats_rocprofv3.csv" 2>/dev/null || find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_kernel_stats_rocprofv3.csv" \;
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- $B --steps 4 --warmup 2 > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- $B --steps 4 --warmup 2 > "$OUT/pmc_write.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -o p -- $B --steps 4 --warmup 2 > "$OUT/pmc_sq.log" 2>&1
python "$REPO/tools/summarize_pmc.py" "$OUT" "$TAG" "${KERNEL:-trunkw_kernel<64}" > "$OUT/${TAG}_trunk_pmc.json"
(cd "$REPO" && python bench.py --steps 100 --warmup 10 > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err")
# keep the merge-back small: only the summaries travel
rm -rf "$OUT"/stats "$OUT"/pmc_fetch/*kernel_trace* "$OUT"/pmc_write/*kernel_trace* "$OUT"/pmc_sq/*kernel_trace*
ls -la "$OUT"
