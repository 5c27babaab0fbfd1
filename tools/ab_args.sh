#!/bin/bash
# A/B of bench.py arguments at N=1: usage tools/ab_args.sh "--batch 111" "--batch 150" ...
mkdir -p gpurun_out/r2
for a in "$@"; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-host $a > gpurun_out/r2/ab.json 2> gpurun_out/r2/ab.err || { tail -5 gpurun_out/r2/ab.err; exit 1; }
  python - "$a" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r2/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "ok", d["config"].get("roundtrip_ok"))
PY
done
