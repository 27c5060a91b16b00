#!/bin/bash
# usage: bench_split.sh NW:SPLIT ...
mkdir -p gpurun_out; : > gpurun_out/split.log
for v in "$@"; do
  IFS=: read n sp <<< "$v"
  export SASKTRAN2_AMD_SPLIT=$sp
  echo "== wavelengths=$n split=$sp" >> gpurun_out/split.log
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --wavelengths $n 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]),"wl/s", round(d["ms_per_step"],1), "ms")
' >> gpurun_out/split.log || exit 1
done
cat gpurun_out/split.log
