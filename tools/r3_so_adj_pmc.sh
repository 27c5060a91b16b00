#!/bin/bash
# wave-state counters of the successive-orders adjoint kernels: tools/r3_so_adj_pmc.sh nw
nw=${1:-1024}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r3_so_adj_pmc
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $out/wave -- python3 bench.py --workload c5 --wavelengths $nw --steps 1 --warmup 0 --weighting-functions --no-cpu-baseline > $out/wave.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py --workload c5 --wavelengths $nw --steps 1 --warmup 0 --weighting-functions --no-cpu-baseline > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py --workload c5 --wavelengths $nw --steps 1 --warmup 0 --weighting-functions --no-cpu-baseline > $out/write.log 2>&1
python3 - $out <<'PY'
import csv, glob, re, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for sub in ("wave", "fetch", "write"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_so_\w+)", r["Kernel_Name"])
            if not m: continue
            res[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE", "WRITE_SIZE"): cnt[(m.group(1), r["Counter_Name"])] += 1
rows = []
for k, v in res.items():
    wc = v.get("SQ_WAVE_CYCLES", 0)
    rows.append({"kernel": k, "wait_any_over_wave_cycles": v.get("SQ_WAIT_ANY", 0) / wc if wc else None, "valu_active_over_wave_cycles": v.get("SQ_ACTIVE_INST_VALU", 0) / wc if wc else None,
                 "fetch_kb_total_x2_corrected": v.get("FETCH_SIZE", 0) * 2, "write_kb_total": v.get("WRITE_SIZE", 0), "dispatches": cnt[(k, "SQ_WAVES")]})
rows.sort(key=lambda r: -(r["fetch_kb_total_x2_corrected"] + r["write_kb_total"]))
json.dump(rows, open(f"{out}/summary.json", "w"), indent=1)
for r in rows[:10]: print(r)
PY
rm -rf $out/wave $out/fetch $out/write
