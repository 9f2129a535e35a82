"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel (mean per dispatch)."""
import csv, sys, collections
path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path) as f:
    for r in csv.DictReader(f):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
