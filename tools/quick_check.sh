#!/bin/bash
# quick GPU regression + bench: parity subset, then bench N=1 (no CPU leg)
set -e
mkdir -p gpurun_out/r3q
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -m gpu -k "not config4 and not failed_alloc and not mixed256" > gpurun_out/r3q/quick_parity.log 2>&1 || { tail -30 gpurun_out/r3q/quick_parity.log; exit 1; }
tail -2 gpurun_out/r3q/quick_parity.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-host "$@" > gpurun_out/r3q/quick_bench.json 2> gpurun_out/r3q/quick_bench.err || { tail -20 gpurun_out/r3q/quick_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3q/quick_bench.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "ok", d["config"].get("roundtrip_ok"))
print("roofline", d.get("roofline"))
print("stages", d.get("stage_ms_per_step"))
print("bwt", d.get("bwt_stage"))
PY
