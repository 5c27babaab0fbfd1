#!/usr/bin/env python3
"""Per-step kernel times from a rocprofv3 kernel_stats.csv: usage kstats_filter.py <dir> <steps> <substr> ..."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = int(sys.argv[2])
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    if any(k in n for k in sys.argv[3:]):
        print("%8.3f ms/step  calls/step %5.1f  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / steps / 1e6,
              int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, n))
