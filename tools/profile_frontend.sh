#!/bin/bash
# Profiling of the optical-property front end (bench.py --workload frontend) on the GPU box: kernel trace of the timed run + separate PMC
# passes (HBM bytes) of one step.
# usage (inside gpurun): tools/profile_frontend.sh <tag>   -> gpurun_out/prof_<tag>/...
set -e
tag=${1:-frontend}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $root/bench.py --workload frontend --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $out/trace -o r -- $BENCH --steps 20 --warmup 3 > $out/trace.log 2>&1
echo "trace done"
PMCARGS="--steps 1 --warmup 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- $BENCH $PMCARGS > $out/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o r -- $BENCH $PMCARGS > $out/write.log 2>&1
echo "write done"
cd $root
PMC_WAVELENGTHS=10000 python3 tools/summarize_profile.py $out > $out/summary.json
tail -c 3000 $out/summary.json
