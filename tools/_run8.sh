cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/p1
rocprofv3 --kernel-trace --output-format csv -d /tmp/p1 -- python tools/valar_bench.py 1 > /tmp/p1.log 2>&1
python3 - $(find /tmp/p1 -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# second frame = second half by count of g_input_u8 launches
idx = [i for i, r in enumerate(rows) if 'g_input_u8' in r['Kernel_Name']]
start = idx[len(idx)//2]
rows = rows[start:]
tot = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name'][:64]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    t = tot.setdefault(n, [0, 0.0]); t[0] += 1; t[1] += d
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
busy = sum(v[1] for v in tot.values())
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-66s n %5d  %9.1f us  %5.1f %%" % (k, v[0], v[1], 100 * v[1] / busy))
print("frame span %.1f us, kernels busy %.1f us, launches %d" % (span, busy, len(rows)))
PY
