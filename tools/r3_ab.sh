#!/bin/bash
# A/B runs of the C2 bench with environment switches: tools/r3_ab.sh TAG "ENV=.. ENV=.." [TAG2 "ENV.."] ...
# prints value / ms per step / per-kernel ms (pipelined per step and isolated)
while [ $# -gt 0 ]; do
  tag=$1; envs=$2; shift 2
  env $envs python bench.py --no-cpu-baseline --no-host-path --steps ${STEPS:-5} --warmup 2 > gpurun_out/r3_bench_$tag.json 2> gpurun_out/r3_bench_$tag.err
  python - $tag <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r3_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"]), round(d["ms_per_step"], 1), {k: round(v["total_ms"] / d["steps"], 1) for k, v in d["kernels_ms"].items() if v["total_ms"] / d["steps"] > 1},
          "ISO", {k: round(v["total_ms"], 1) for k, v in d["kernels_ms_isolated"].items() if v["total_ms"] > 1},
          "parity", d.get("parity_radiance_max_rel"), flush=True)
except Exception as e:
    print(n, "fail", e, open(f"gpurun_out/r3_bench_{n}.err").read()[-800:], flush=True)
PY
done
