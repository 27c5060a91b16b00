#!/bin/bash
# usage: bench_k1split.sh variant:split ...   (single stream, 4096-wavelength launches)
mkdir -p gpurun_out; : > gpurun_out/k1split.log
export SASKTRAN2_AMD_STREAMS=1 SASKTRAN2_AMD_CHUNK=4096
for vs in "$@"; do
  v=${vs%%:*}; export SASKTRAN2_AMD_K1SPLIT=${vs##*:}
  if [ "$v" = base ]; then unset SASKTRAN2_AMD_LIB; else export SASKTRAN2_AMD_LIB=$PWD/sasktran2_amd/_lib/variants/$v.so; fi
  echo "== $vs" >> gpurun_out/k1split.log
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]),"wl/s k1", round(d["kernels_ms"]["k1_layer_solution"]["total_ms"],1))
' >> gpurun_out/k1split.log || exit 1
done
cat gpurun_out/k1split.log
