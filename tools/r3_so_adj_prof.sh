#!/bin/bash
# per-kernel times of the C5 weighting-function pass: tools/r3_so_adj_prof.sh nw
nw=${1:-1024}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r3_so_adj
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 bench.py --workload c5 --wavelengths $nw --steps 1 --warmup 0 --weighting-functions --no-cpu-baseline > $out/bench.log 2>&1
f=$(find $out/raw -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv
rm -rf $out/raw
head -14 $out/kernel_stats.csv | cut -c1-200
