#!/bin/bash
# the multi-device parity test N times in fresh processes (flake hunt); stops at the first failure
N=${1:-10}
mkdir -p gpurun_out/r3q
for i in $(seq 1 $N); do
  timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 150 --timeout-method thread -m gpu -k "multi_device" > gpurun_out/r3q/md_$i.txt 2>&1 || { echo "run $i FAILED"; tail -30 gpurun_out/r3q/md_$i.txt; exit 1; }
  echo "run $i ok"
done
