#!/bin/bash
# kernel timeline of one small one-shot step (default 8 MiB): where a small batch's time goes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SEG=${1:-8}
OUT=$ROOT/gpurun_out/r3q/trace_small_$SEG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/bench.py --segments $SEG --steps 3 --warmup 1 --no-cpu --no-host --no-profile > $OUT/log 2>&1
cd $ROOT
F=$(ls $OUT/t/*/*kernel_trace.csv | head -1)
TL_FROM=0 TL_TO=100 python3 tools/timeline.py $F
rm -rf $OUT/t
