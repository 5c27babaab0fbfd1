#!/bin/bash
# One rocprofv3 --pmc pass with SQ counters over bench.py (1 step); prints the counters of kernels
# whose name matches $1.   usage: tools/pmc_sq.sh <kernel substring> <counter> [counter ...]  (<= 8 SQ_)
set -e
PAT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3q/pmc_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-host --no-profile > $OUT/log.txt 2>&1
cd $ROOT
python3 tools/pmc_summary.py $OUT | grep -E "kernel,|$PAT" > gpurun_out/r3q/pmc_sq.csv
rm -rf $OUT
cat gpurun_out/r3q/pmc_sq.csv
