#!/bin/bash
# A/B of environment knobs on bench.py N=1: usage tools/ab_env.sh "VAR=a" "VAR=b" ...
mkdir -p gpurun_out/r3q
for kv in "$@"; do
  env $kv timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-host > gpurun_out/r3q/ab.json 2> gpurun_out/r3q/ab.err || { tail -5 gpurun_out/r3q/ab.err; exit 1; }
  python - "$kv" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3q/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "ok", d["config"].get("roundtrip_ok"), d.get("stage_ms_per_step"))
PY
done
