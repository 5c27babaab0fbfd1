#!/bin/bash
# kernel-time stats of bench.py under rocprofv3 (kernel trace only); prints the top kernels
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3q/ks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-host > $OUT/stats.log 2>&1
cd $ROOT
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/3e6:8.3f} ms/step  calls {int(r['Calls'])/3:6.1f}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:70]}")
PY
grep -h '"metric"' $OUT/stats.log | tail -1 | cut -c1-200
