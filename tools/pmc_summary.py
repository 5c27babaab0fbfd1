#!/usr/bin/env python3
"""Aggregates rocprofv3 csv output (kernel trace + one --pmc counter) per kernel name.
usage: pmc_summary.py <dir> -> prints kernel, calls, total / per-call counter value"""
import csv
import glob
import sys
from collections import defaultdict


def main(d):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?").split("(")[0]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[(k, row["Counter_Name"])] += 1
    print("kernel,counter,calls,total,per_call")
    for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
        for cn, v in agg[k].items():
            c = calls[(k, cn)]
            print(f"{k},{cn},{c},{v:.0f},{v / max(c, 1):.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
