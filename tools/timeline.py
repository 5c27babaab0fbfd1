#!/usr/bin/env python3
"""Timeline of ONE bench step from a rocprofv3 kernel trace: GPU-busy time, idle gaps (host round
trips), per-kernel busy share, concurrency.  usage: timeline.py <kernel_trace.csv> [step_index]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows]
ev.sort()
# steps start at k1_heads_p1
starts = [i for i, e in enumerate(ev) if e[2].startswith("k1_heads_p1")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
lo = starts[k]
hi = starts[k + 1] if k + 1 < len(starts) else len(ev)
step = [e for e in ev[lo:hi] if not e[2].startswith("kd_")]
t0 = step[0][0]
t1 = max(e[1] for e in step)
print(f"step {k}: {len(step)} kernels, span {(t1 - t0) / 1e6:.3f} ms")
# union of busy intervals
busy = 0
cur_s, cur_e = step[0][0], step[0][1]
gaps = []
for s, e, n in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"GPU busy (union) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
gaps.sort(reverse=True)
for g, at, n in gaps[:12]:
    print(f"   gap {g / 1e3:8.1f} us at {at / 1e6:7.3f} ms before {n}")
# exclusive time: time when a kernel is the only one running, and total
tot = defaultdict(int)
for s, e, n in step:
    tot[n] += e - s
print("sum of kernel durations %.3f ms" % (sum(tot.values()) / 1e6))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / 1e6:7.3f} ms  {n}")
# coarse timeline: phase boundaries
marks = ["k1_heads_p1", "k2_reset", "k2_keys", "k2_emit_bwt", "k3_mtf_recency", "k4_seed", "k4_plan", "k8_frame"]
for m in marks:
    for s, e, n in step:
        if n.startswith(m):
            print(f"  first {m:16s} at {(s - t0) / 1e6:7.3f} ms")
            break

# optional: list every kernel of a time window (ms from the step start): TL_FROM / TL_TO
import os
if os.environ.get("TL_FROM"):
    a, b = float(os.environ["TL_FROM"]) * 1e6, float(os.environ.get("TL_TO", "1e9")) * 1e6
    for s_, e_, n_ in step:
        if a <= s_ - t0 <= b:
            print(f"  {(s_ - t0) / 1e6:8.3f} +{(e_ - s_) / 1e3:8.1f} us  {n_[:50]}")
