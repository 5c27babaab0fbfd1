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


def collect(d):
    agg = defaultdict(float)
    calls = defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?").split("(")[0]
            agg[(k, row["Counter_Name"])] += float(row["Counter_Value"])
            calls[(k, row["Counter_Name"])] += 1
    return agg, calls


def main_json(dirs):
    """-> {kernel: {"calls", "fetch_kb_per_call", "write_kb_per_call", "hbm_bytes_per_call"}}.
    FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE under-reports 8 B/lane streaming reads by 2x on
    gfx950 (calibrated on k2_radix_hist), so reads are doubled."""
    import json
    out = {}
    for d in dirs:
        agg, calls = collect(d)
        for (k, cn), v in agg.items():
            e = out.setdefault(k, {})
            e["calls"] = calls[(k, cn)]
            if cn == "FETCH_SIZE":
                e["fetch_kb_per_call"] = round(v / calls[(k, cn)], 1)
            if cn == "WRITE_SIZE":
                e["write_kb_per_call"] = round(v / calls[(k, cn)], 1)
    for k, e in out.items():
        e["hbm_bytes_per_call"] = round((2.0 * e.get("fetch_kb_per_call", 0.0) + e.get("write_kb_per_call", 0.0)) * 1024)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    if sys.argv[1] == "--json":
        main_json(sys.argv[2:])
    else:
        main(sys.argv[1])
