#!/bin/bash
# Run ON THE MI355X BOX: are sub10_kernel's LDS bank conflicts (14.5 % of its LDS cycles, the ring stores) time?  The product library against
# a build whose ring stores go to lane-linear, conflict-free addresses (tools/experiments/sub10_linear_stores.patch: wrong results, the same
# instructions): frames/s three times in turn, then the two LDS counters of both.
exec < /dev/null
O=gpurun_out/r06k; mkdir -p $O
R=$PWD
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "frames/s,", d["roofline"]["avg_launch_ms"], "ms per launch")'
{
cd /tmp && export TMPDIR=/tmp
for lib in product s10lin s10head s10both; do
  L=$R/upscale_video_amd/libuva.so; [ $lib != product ] && L=$R/upscale_video_amd/libuva_$lib.so
  rm -rf /tmp/pmc_$lib
  UVA_LIB_PATH=$L timeout 400 rocprofv3 --kernel-trace --stats --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS -d /tmp/pmc_$lib -o pmc --output-format csv -- python $R/tools/time_1x.py 40 > /tmp/pmc_$lib.log 2>&1
  python - /tmp/pmc_$lib $lib <<'PY'
import csv, glob, sys, statistics, collections
vals = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sub10" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: statistics.median(v) for k, v in vals.items()}
print("%-8s" % sys.argv[2], {k: int(v) for k, v in sorted(m.items())}, "conflict share %.1f %%" % (100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
done
for i in 1 2 3; do
  for lib in product s10lin s10head s10both; do
    L=$R/upscale_video_amd/libuva.so; [ $lib != product ] && L=$R/upscale_video_amd/libuva_$lib.so
    echo -n "$lib: "; UVA_LIB_PATH=$L timeout 200 python $R/tools/time_1x.py 3000 2>/dev/null | tail -1
  done
done
} > $R/$O/sub10_lds_conflicts.txt 2>&1
cat $R/$O/sub10_lds_conflicts.txt
