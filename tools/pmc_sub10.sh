#!/bin/bash
# SQ counters of sub10_kernel (the fused 1x net), one rocprofv3 pass per group; prints per-launch medians.
# Run on the GPU box:  tools/pmc_sub10.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-/tmp/pmc_sub10}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_IFETCH SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o pmc --output-format csv -- python $R/bench.py --workload 1x_hurrdeblur_1080p --batch 1 --tile 0 --no-cpu-baseline --steps 10 --warmup 2 > $OUT.log 2>&1 || { echo "group failed: $grp"; tail -3 $OUT.log; }
done
python - "$OUT" <<'PY'
import csv, glob, sys, statistics, collections
vals = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sub10" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals):
    print("%-36s %16.0f" % (k, statistics.median(vals[k])))
PY
