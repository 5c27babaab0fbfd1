#!/bin/bash
# kernel trace of one decode (tools/decode_once.py): start / end of every kd_* launch relative to the first
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2d/dectrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/decode_once.py > $OUT/log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/t/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("kd_", "k1_crc"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last decode: from the last kd_scan_magic on
i0 = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("kd_scan_magic"))
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e6:9.3f} .. {e / 1e6:9.3f} ms  {(e - s) / 1e6:8.3f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'].split('(')[0]}  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}")
PY
rm -rf $OUT/t
