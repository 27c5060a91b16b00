#!/bin/bash
# runs bench.py (no CPU baseline) for each tuning variant given on the command line; results in gpurun_out/variants.log
mkdir -p gpurun_out; : > gpurun_out/variants.log
for v in "$@"; do
  if [ "$v" = base ]; then unset SASKTRAN2_AMD_LIB; else export SASKTRAN2_AMD_LIB=$PWD/sasktran2_amd/_lib/variants/$v.so; fi
  echo "== $v" >> gpurun_out/variants.log
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]),"wl/s", {k:round(v["total_ms"],1) for k,v in d["kernels_ms"].items() if v["total_ms"]>5})
' >> gpurun_out/variants.log || exit 1
done
cat gpurun_out/variants.log
