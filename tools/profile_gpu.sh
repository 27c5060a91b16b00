#!/bin/bash
# Round profiling on the GPU box: kernel trace (durations) + separate PMC passes (HBM bytes, FP64 ops, wave state).
# usage (inside gpurun): tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/{trace,fetch,write,f64,mfma,wave}/...
# One rocprofv3 command per pass, the program itself right after "--" (no wrappers), counters never combined with tracing.
set -e
tag=${1:-run}
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $root/bench.py --no-cpu-baseline --no-host-path"
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $out/trace -o r -- $BENCH --steps 2 --warmup 1 > $out/trace.log 2>&1
echo "trace done"
PMCARGS="--steps 1 --warmup 0 --wavelengths 4096"
export SASKTRAN2_AMD_STREAMS=1 SASKTRAN2_AMD_CHUNK=4096
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- $BENCH $PMCARGS > $out/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o r -- $BENCH $PMCARGS > $out/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace -d $out/f64 -o r -- $BENCH $PMCARGS > $out/f64.log 2>&1
echo "f64 done"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $out/mfma -o r -- $BENCH $PMCARGS > $out/mfma.log 2>&1
echo "mfma done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace -d $out/wave -o r -- $BENCH $PMCARGS > $out/wave.log 2>&1
echo "wave done"
cd $root
python3 tools/summarize_profile.py $out > $out/summary.json
# gpurun merges at most 64 MiB back: keep the summary, the kernel statistics of the trace and the logs, drop the raw databases / traces
cp $out/trace/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || true
rm -rf $out/trace $out/fetch $out/write $out/f64 $out/mfma $out/wave
tail -c 1500 $out/summary.json
