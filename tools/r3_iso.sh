#!/bin/bash
# isolated per-kernel times (ms per step) of the C2 size with the given environment: tools/r3_iso.sh "ENV=.." ...
for envs in "$@"; do
  env $envs python bench.py --no-cpu-baseline --no-host-path --steps 10 --warmup 3 > gpurun_out/iso_tmp.json 2>/dev/null
  python - "$envs" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/iso_tmp.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), {k: round(v["total_ms"], 2) for k, v in d["kernels_ms_isolated"].items() if v["launches"] > 1}, flush=True)
PY
done
