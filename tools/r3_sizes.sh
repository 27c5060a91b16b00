#!/bin/bash
# per-rank sizes of the strong-scaling run on one GPU: tools/r3_sizes.sh "ENV=.. ENV=.." n1 n2 ...
envs=$1; shift
for n in "$@"; do
  env $envs python bench.py --no-cpu-baseline --no-host-path --steps 10 --warmup 3 --wavelengths $n > gpurun_out/r3_size_tmp.json 2>/dev/null
  python - $n "$envs" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r3_size_tmp.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], round(d["ms_per_step"], 2), {k: round(v["total_ms"] / d["steps"], 1) for k, v in d["kernels_ms"].items() if v["total_ms"] / d["steps"] > 0.5}, flush=True)
PY
done
