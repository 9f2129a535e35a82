"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel (mean per dispatch, launch geometry).
usage: pmc_summary.py <p_counter_collection.csv> [kernel-name substring]"""
import csv, sys, collections
path = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
geo = {}
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if only and only not in name:
            continue
        key = name[:72]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        geo[key] = "grid %s wg %s lds %s vgpr %s sgpr %s scratch %s" % (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"],
                                                                     r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
for k, d in acc.items():
    print(k, "|", geo[k])
    for c, v in sorted(d.items()):
        print("   %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
