#!/usr/bin/env python3
"""Prints the first N durations of a kernel from a rocprofv3 kernel-trace csv directory."""
import csv, glob, sys
d, name, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(name, [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows[:n]], "us")
