#!/bin/bash
# bench.py at several spectrum sizes for tuning variants: tools/bench_variant_sizes.sh "1250 10000" base k1w4 ...
sizes=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset SASKTRAN2_AMD_LIB; else export SASKTRAN2_AMD_LIB=$PWD/sasktran2_amd/_lib/variants/$v.so; fi
  for n in $sizes; do
    timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --wavelengths $n 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], sys.argv[2], round(d["ms_per_step"],2), "ms")
' $v $n
  done
done
