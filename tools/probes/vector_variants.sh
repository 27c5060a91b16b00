#!/bin/bash
# per-kernel times of the polarised path for tuning variants of the library (experiment switches): tools/probes/vector_variants.sh base v1 v2 ...
for v in "$@"; do
  if [ "$v" = base ]; then unset SASKTRAN2_AMD_LIB; else export SASKTRAN2_AMD_LIB=$PWD/sasktran2_amd/_lib/variants/$v.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/probes/vector_rate.py 2>&1 | grep -v "^W2026\|^E2026"
done
