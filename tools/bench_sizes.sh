#!/bin/bash
# throughput of one GPU at the per-rank wavelength counts of the 1/2/4/8-GPU strong-scaling runs
mkdir -p gpurun_out; : > gpurun_out/sizes.log
for n in "$@"; do
  echo "== wavelengths=$n" >> gpurun_out/sizes.log
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --wavelengths $n 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]),"wl/s", round(d["ms_per_step"],1), "ms", d["config"]["wavelengths_per_launch"], "x", d["config"]["blocks_in_flight"])
' >> gpurun_out/sizes.log || exit 1
done
cat gpurun_out/sizes.log
