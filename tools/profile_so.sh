#!/bin/bash
# Profiling of the exact single-scatter / successive-orders path (bench.py --workload c5) on the GPU box: kernel trace of the full
# 20 000-wavelength run + separate PMC passes (HBM bytes, FP64 ops, wave state) at 4096 wavelengths per launch.
# usage (inside gpurun): tools/profile_so.sh <tag>   -> gpurun_out/prof_<tag>/...
set -e
tag=${1:-so}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $root/bench.py --workload c5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $out/trace -o r -- $BENCH --steps 2 --warmup 1 > $out/trace.log 2>&1
echo "trace done"
PMCARGS="--steps 1 --warmup 0 --wavelengths 4096"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- $BENCH $PMCARGS > $out/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o r -- $BENCH $PMCARGS > $out/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace -d $out/f64 -o r -- $BENCH $PMCARGS > $out/f64.log 2>&1
echo "f64 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace -d $out/wave -o r -- $BENCH $PMCARGS > $out/wave.log 2>&1
echo "wave done"
cd $root
python3 tools/summarize_profile.py $out > $out/summary.json
tail -c 1200 $out/summary.json
