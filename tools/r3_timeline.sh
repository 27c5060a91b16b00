#!/bin/bash
# kernel timeline of a small launch: tools/r3_timeline.sh tag nw  (env passed through)
tag=$1; nw=$2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r3_tl_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 bench.py --no-cpu-baseline --no-host-path --steps 3 --warmup 2 --wavelengths $nw > $out/bench.log 2>&1
f=$(find $out/raw -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $out/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last kern_k0t backwards
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("kern_k0t") or "kern_k0t" in n]
i0, i1 = idx[3], idx[4]   # the fourth pass: pipelined (the last ones are the serialised per-kernel timing passes)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    n = r["Kernel_Name"].split("(")[0][:40]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} q{r.get("Queue_Id", "?")} {n}')
PY
rm -rf $out/raw
