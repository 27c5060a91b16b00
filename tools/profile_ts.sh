#!/bin/bash
# Profiling of the two-stream path (bench.py --workload twostream) on the GPU box: kernel trace of the full 220 000-wavelength run +
# separate PMC passes (HBM bytes, FP64 ops, wave state) of the same run (one launch per kernel).
# usage (inside gpurun): tools/profile_ts.sh <tag>   -> gpurun_out/prof_<tag>/...
set -e
tag=${1:-ts}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $root/bench.py --workload twostream --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $out/trace -o r -- $BENCH --steps 3 --warmup 1 > $out/trace.log 2>&1
echo "trace done"
PMCARGS="--steps 1 --warmup 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- $BENCH $PMCARGS > $out/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o r -- $BENCH $PMCARGS > $out/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace -d $out/f64 -o r -- $BENCH $PMCARGS > $out/f64.log 2>&1
echo "f64 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace -d $out/wave -o r -- $BENCH $PMCARGS > $out/wave.log 2>&1
echo "wave done"
cd $root
PMC_WAVELENGTHS=220000 python3 tools/summarize_profile.py $out > $out/summary.json
tail -c 2500 $out/summary.json
