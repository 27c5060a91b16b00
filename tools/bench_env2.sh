#!/bin/bash
# usage: bench_env2.sh CHUNK:STREAMS:SPLIT ...
mkdir -p gpurun_out; : > gpurun_out/env2.log
for v in "$@"; do
  IFS=: read c s sp <<< "$v"
  export SASKTRAN2_AMD_CHUNK=$c SASKTRAN2_AMD_STREAMS=$s
  if [ "$sp" = auto ]; then unset SASKTRAN2_AMD_SPLIT; else export SASKTRAN2_AMD_SPLIT=$sp; fi
  echo "== chunk=$c streams=$s split=$sp" >> gpurun_out/env2.log
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]),"wl/s", round(d["ms_per_step"],1))
' >> gpurun_out/env2.log || exit 1
done
cat gpurun_out/env2.log
