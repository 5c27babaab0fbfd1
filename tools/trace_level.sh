#!/bin/bash
# kernel timeline of one step at another level: tools/trace_level.sh <level>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LV=${1:-1}
OUT=$ROOT/gpurun_out/r3q/trace_lv$LV
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/bench.py --level $LV --steps 3 --warmup 1 --no-cpu --no-host --no-profile > $OUT/log 2>&1
cd $ROOT
F=$(ls $OUT/t/*/*kernel_trace.csv | head -1)
python3 tools/timeline.py $F
rm -rf $OUT/t
