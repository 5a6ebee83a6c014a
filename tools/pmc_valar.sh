#!/bin/bash
# 4x_Valar_v1 at 1920x1080 (config 4 as named, synthetic weights): HBM bytes per frame (FETCH_SIZE / WRITE_SIZE summed
# over every kernel of one frame) and SQ counters of the dense-block kernels.  One rocprofv3 --pmc pass per counter
# group, --kernel-trace only.  Run on the GPU box:  bash tools/pmc_valar.sh [env settings, e.g. UVA_GENERIC_RDB=0]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for kv in "$@"; do export "$kv"; done
B="python $R/tools/valar_bench.py 1"
rm -rf /tmp/valpmc
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/valpmc/fetch -o p -- $B > /tmp/valpmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/valpmc/write -o p -- $B > /tmp/valpmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/valpmc/sq -o p -- $B > /tmp/valpmc_s.log 2>&1
python - "$@" <<'PY'
import csv, glob, collections, sys
print("4x_Valar_v1 1920x1080 -> 7680x4320, reference tiling, synthetic weights; settings:", " ".join(sys.argv[1:]) or "(defaults)")
def load(sub):
    rows = []
    for f in glob.glob("/tmp/valpmc/%s/*counter_collection.csv" % sub):
        rows += list(csv.DictReader(open(f)))
    return rows
def second_frame(rows):
    # valar_bench.py 1 runs two frames (one warm-up): keep the dispatches of the second one
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    first_of = {}
    for r in rows:
        if "g_input_u8" in r["Kernel_Name"]:
            first_of[int(r["Dispatch_Id"])] = 1
    starts = sorted(first_of)
    cut = starts[len(starts) // 2]
    return [r for r in rows if int(r["Dispatch_Id"]) >= cut]
tot = {}
for sub, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = [r for r in second_frame(load(sub)) if r["Counter_Name"] == name]
    per = collections.defaultdict(float)
    for r in rows:
        per[r["Kernel_Name"][:48]] += float(r["Counter_Value"])
    tot[name] = sum(per.values())
    print("%s per frame: %.1f GB raw (KB counter x 1024)" % (name, tot[name] * 1024 / 1e9))
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:6]:
        print("    %-50s %8.2f GB" % (k, v * 1024 / 1e9))
rd, wr = tot["FETCH_SIZE"] * 1024 * 2, tot["WRITE_SIZE"] * 1024
print("HBM traffic per frame: read %.1f GB (FETCH_SIZE x2: gfx950 counts 64 B per 128-B request on wide coalesced reads, "
      "MI355X_MICROARCH.md) + written %.1f GB = %.1f GB" % (rd / 1e9, wr / 1e9, (rd + wr) / 1e9))
rows = second_frame(load("sq"))
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    per[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
print("SQ counters, summed over the frame's launches of a kernel:")
for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:5]:
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0)
    print("    %-50s LDS bank-conflict cycles / LDS active %5.1f %%   waves waiting for an instruction %5.1f %% of wave cycles   "
          "matrix pipes busy %5.1f %% of kernel cycles" % (
              k, 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / lds if lds else 0,
              100 * c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0,
              100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["SQ_BUSY_CYCLES"] / 32 * 1024) if c.get("SQ_BUSY_CYCLES") else 0))
PY
