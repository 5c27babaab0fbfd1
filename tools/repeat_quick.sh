#!/bin/bash
# quick_check's pytest line N times (flake hunt); stops at the first failure
N=${1:-4}
mkdir -p gpurun_out/r3q
for i in $(seq 1 $N); do
  timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -m gpu -k "not config4 and not failed_alloc and not mixed256" > gpurun_out/r3q/rq_$i.txt 2>&1
  rc=$?
  echo "run $i rc=$rc: $(tail -1 gpurun_out/r3q/rq_$i.txt)"
  if [ $rc -ne 0 ]; then grep -E "^FAILED|^E  " gpurun_out/r3q/rq_$i.txt | head; exit 1; fi
done
