#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-time stats and the two PMC passes for bench.py, each
# in its own rocprofv3 run.  Results land in gpurun_out/prof_<tag>/; copy what is to be judged
# into profiles/.   usage: tools/profile_round.sh <tag>
set -e
TAG=${1:-rX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-host > $OUT/stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-host --no-profile > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-host --no-profile > $OUT/pmc_write.log 2>&1
cd $ROOT
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 tools/pmc_summary.py $OUT/pmc_fetch > $OUT/pmc_fetch_size_kb.csv
python3 tools/pmc_summary.py $OUT/pmc_write > $OUT/pmc_write_size_kb.csv
python3 tools/pmc_summary.py --json $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_under_rocprof.json || true
head -12 $OUT/kernel_stats.csv
