"""Average every counter of a rocprofv3 --pmc counter_collection.csv per kernel (name filter optional).
    python tools/pmc_raw.py <csv> [substring]"""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if sub not in k:
        continue
    k = k.split("(")[0][-60:]
    a = agg[k][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print("   %-32s n=%4d avg %.4g" % (c, n, v / n))
