import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2:] or ['g_conv3_sw']
groups = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name'][:60]
    if not any(p in name for p in pat):
        continue
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    groups.setdefault(name, []).append(d)
for k, v in groups.items():
    big = sorted(x for x in v if x > 0.5 * max(v))
    small = sorted(x for x in v if x <= 0.5 * max(v))
    med = lambda a: a[len(a) // 2] if a else 0
    print("%-58s n %5d sum %9.1f us | large launches: n %4d med %7.1f | small: n %4d med %7.1f" % (k, len(v), sum(v), len(big), med(big), len(small), med(small)))
