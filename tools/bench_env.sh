#!/bin/bash
# runs bench.py (no CPU baseline) for "CHUNK:STREAMS" settings given on the command line; results in gpurun_out/env.log
mkdir -p gpurun_out; : > gpurun_out/env.log
for v in "$@"; do
  export SASKTRAN2_AMD_CHUNK=${v%%:*} SASKTRAN2_AMD_STREAMS=${v##*:}
  echo "== chunk=$SASKTRAN2_AMD_CHUNK streams=$SASKTRAN2_AMD_STREAMS" >> gpurun_out/env.log
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]),"wl/s", d["ms_per_step"], "valid", d.get("valid"), "scratch GiB", round(d.get("scratch_gib",0),2))
' >> gpurun_out/env.log || exit 1
done
cat gpurun_out/env.log
