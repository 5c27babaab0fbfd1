#!/bin/bash
# kernel-time stats of the decoder alone (rocprofv3 kernel trace of tools/decode_once.py); top kernels
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2/dec_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/decode_once.py > $OUT/stats.log 2>&1
cd $ROOT
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:24]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.3f} ms  calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:70]}")
PY
tail -3 $OUT/stats.log
