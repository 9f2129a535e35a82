"""Print the per-kernel average durations of a rocprofv3 *_kernel_stats.csv (measurement aid)."""
import csv, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        if r["Name"].startswith(("void apm_", "apm_")) and "synth" not in r["Name"]:
            print("  %-48s calls %4s avg %10.1f us" % (r["Name"][5:53], r["Calls"], float(r["AverageNs"]) / 1e3))
