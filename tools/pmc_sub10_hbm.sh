#!/bin/bash
# HBM bytes of sub10_kernel per 1080p launch (FETCH_SIZE / WRITE_SIZE, one rocprofv3 --pmc pass each, --kernel-trace only).
# Run on the GPU box:  bash tools/pmc_sub10_hbm.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B="python $R/bench.py --workload 1x_hurrdeblur_1080p --batch 1 --no-cpu-baseline --steps 6 --warmup 2"
rm -rf /tmp/s10hbm
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/s10hbm/pmc_fetch -o p -- $B > /tmp/s10hbm_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/s10hbm/pmc_write -o p -- $B > /tmp/s10hbm_w.log 2>&1
python - <<'PY'
import csv, glob, statistics
def med(sub, name):
    v = {}
    for f in glob.glob("/tmp/s10hbm/%s/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if "sub10" in r["Kernel_Name"] and r["Counter_Name"] == name:
                v[r["Dispatch_Id"]] = v.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    big = [x for x in v.values() if x > 0.5 * max(v.values())]
    return statistics.median(big), len(big)
f, nf = med("pmc_fetch", "FETCH_SIZE")
w, nw = med("pmc_write", "WRITE_SIZE")
frame = 1920 * 1080 * 3
print("sub10_kernel, 1920x1080 u8 frame (%.2f MB in, %.2f MB out), median of %d / %d full-size launches" % (frame / 1e6, frame / 1e6, nf, nw))
print("  FETCH_SIZE %.0f KB raw -> %.2f MB (x1: 3-byte pixel loads are not the wide coalesced reads the x2 correction is for) .. %.2f MB (x2)" % (f, f * 1024 / 1e6, 2 * f * 1024 / 1e6))
print("  WRITE_SIZE %.0f KB -> %.2f MB" % (w, w * 1024 / 1e6))
print("  per-pair path for comparison: 8 activation images of 99.5 MB written and read again = ~1.6 GB per frame")
PY
