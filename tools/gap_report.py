#!/usr/bin/env python3
"""Idle time between kernels: reads a rocprofv3 kernel-trace CSV, takes the last compress step
(from the last k1_heads_p1 to the last k8_frame) and reports busy union, gaps and the largest gaps.
usage: gap_report.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
starts = [k for k, r in enumerate(rows) if r[2].startswith("k1_heads_p1")]
ends = [k for k, r in enumerate(rows) if r[2].startswith("k8_frame")]
a, b = starts[-1], ends[-1]
seg = rows[a:b + 1]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy, cur_e, gaps = 0, t0, []
prev = None
for s, e, n in seg:
    if s > cur_e:
        gaps.append((s - cur_e, prev, n))
        busy += 0
        cur_s = s
    if e > cur_e:
        busy += e - max(s, cur_e)
        cur_e = e
        prev = n
print(f"step wall {1e-6 * (t1 - t0):.3f} ms, busy union {1e-6 * busy:.3f} ms, idle {1e-6 * (t1 - t0 - busy):.3f} ms in {len(gaps)} gaps, {len(seg)} kernels")
print(f"sum of kernel durations {1e-6 * sum(e - s for s, e, _ in seg):.3f} ms")
for g, p, n in sorted(gaps, reverse=True)[:25]:
    print(f"  {1e-3 * g:8.1f} us  after {p}  before {n}")
