cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in ${SLOTS:-1 2 4 16 256}; do
  timeout 60 $R/tools/l2_ring_bench.bin $S 1024
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rp; timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/rp -o p -- $R/tools/l2_ring_bench.bin $S 1024 > /dev/null 2>&1
    python3 - <<PY
import csv,glob
tot=[]
for f in glob.glob('/tmp/rp/**/*counter_collection.csv', recursive=True):
    d={}
    for r in csv.DictReader(open(f)):
        if 'ring' in r['Kernel_Name']:
            d[r['Dispatch_Id']]=d.get(r['Dispatch_Id'],0)+float(r['Counter_Value'])
    tot=list(d.values())
print('   slots=$S $C per launch (KB):', [round(x) for x in tot])
PY
  done
done
