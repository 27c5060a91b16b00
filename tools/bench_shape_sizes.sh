#!/bin/bash
# usage: bench_shape_sizes.sh NW:CHUNK:STREAMS ...
mkdir -p gpurun_out; : > gpurun_out/shape_sizes.log
for v in "$@"; do
  IFS=: read n c s <<< "$v"
  export SASKTRAN2_AMD_CHUNK=$c SASKTRAN2_AMD_STREAMS=$s
  echo "== wavelengths=$n chunk=$c streams=$s" >> gpurun_out/shape_sizes.log
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --wavelengths $n 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]),"wl/s", round(d["ms_per_step"],1), "ms")
' >> gpurun_out/shape_sizes.log || exit 1
done
cat gpurun_out/shape_sizes.log
