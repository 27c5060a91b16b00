#!/bin/bash
# time of one C5 pass with weighting functions at the given wavelength count (default 1024): tools/r3_c5wf_time.sh [nw]
python bench.py --workload c5 --wavelengths ${1:-1024} --steps 1 --warmup 0 --weighting-functions --no-cpu-baseline > gpurun_out/t_tmp.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/t_tmp.json").read().strip().splitlines()[-1])
print("c5 wf:", round(d["ms_per_step"], 1), "ms", flush=True)
PY
