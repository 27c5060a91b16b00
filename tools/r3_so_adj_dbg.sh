#!/bin/bash
for d in 0; do
  SASKTRAN2_AMD_SO_ADJ_DEBUG=$d python bench.py --workload c5 --wavelengths 1024 --steps 1 --warmup 0 --weighting-functions --no-cpu-baseline > gpurun_out/dbg_tmp.json 2>/dev/null
  python - $d <<'PY'
import json, sys
d = json.loads(open("gpurun_out/dbg_tmp.json").read().strip().splitlines()[-1])
print("dbg", sys.argv[1], round(d["ms_per_step"], 1), flush=True)
PY
done
