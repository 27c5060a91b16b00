#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile_gpu.sh into one JSON document (kept under profiles/).

kernel trace: per-kernel calls / total / average duration of the timed bench run (default launch shape);
PMC passes (4096 wavelengths per launch, one block in flight): per kernel and per launch
  HBM-side bytes (FETCH_SIZE in KB, doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950; WRITE_SIZE in KB),
  FP64 operations executed (VALU: 64 lanes x (2 FMA + ADD + MUL + TRANS) per wave instruction, an upper bound that counts
  inactive lanes; matrix cores: SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flops), and the wave-state ratios."""
import csv
import glob
import sqlite3
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pattern):
    hits = glob.glob(os.path.join(root, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    m = re.search(r"(kern_\w+|k_so_\w+|k_ts_\w+|k_fe_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


out = {"source": os.path.basename(root)}
f = find("trace", "*kernel_stats.csv")
fdb = find("trace", "*results.db")
if f:
    rows = list(csv.DictReader(open(f)))
    out["kernel_trace"] = [{"kernel": short(r["Name"]), "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                            "avg_us": float(r["AverageNs"]) / 1e3, "percent": float(r["Percentage"])} for r in rows[:14]]
elif fdb:  # rocpd database (default output format of this rocprofv3): view top_kernels, durations in microseconds
    con = sqlite3.connect(fdb)
    out["kernel_trace"] = [{"kernel": short(n), "calls": c, "total_ms": t / 1e3, "avg_us": a, "percent": pc}
                           for n, c, t, a, pc in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 14")]


def counters(sub):
    f = find(sub, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    if not f:
        fdb = find(sub, "*results.db")
        if fdb:
            con = sqlite3.connect(fdb)
            for name, cname, val, did in con.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
                k = short(name)
                acc[k][cname] += float(val)
                launches[k].add(did)
        return acc, launches
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return acc, launches


pmc = defaultdict(dict)
for sub in ("fetch", "write", "f64", "mfma", "wave"):
    acc, launches = counters(sub)
    for k, cs in acc.items():
        n = max(1, len(launches[k]))
        pmc[k]["launches"] = n
        for c, v in cs.items():
            pmc[k][c + "_per_launch"] = v / n
rows = []
for k, d in pmc.items():
    r = {"kernel": k, "launches": d.get("launches")}
    if "FETCH_SIZE_per_launch" in d:
        r["hbm_read_bytes_per_launch_x2_corrected"] = d["FETCH_SIZE_per_launch"] * 1024 * 2
    if "WRITE_SIZE_per_launch" in d:
        r["hbm_write_bytes_per_launch"] = d["WRITE_SIZE_per_launch"] * 1024
    valu = sum(d.get(f"SQ_INSTS_VALU_{n}_F64_per_launch", 0) * w for n, w in (("FMA", 2), ("ADD", 1), ("MUL", 1), ("TRANS", 1)))
    if valu:
        r["fp64_valu_flops_per_launch_upper_bound"] = valu * 64
    if "SQ_INSTS_VALU_MFMA_MOPS_F64_per_launch" in d:
        r["fp64_mfma_flops_per_launch"] = d["SQ_INSTS_VALU_MFMA_MOPS_F64_per_launch"] * 512
    if d.get("SQ_BUSY_CYCLES_per_launch"):
        r["mfma_busy_over_sq_busy"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES_per_launch", 0) / d["SQ_BUSY_CYCLES_per_launch"]
    if d.get("SQ_WAVE_CYCLES_per_launch"):
        r["wait_any_over_wave_cycles"] = d.get("SQ_WAIT_ANY_per_launch", 0) / d["SQ_WAVE_CYCLES_per_launch"]
        r["valu_active_over_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU_per_launch", 0) / d["SQ_WAVE_CYCLES_per_launch"]
    rows.append(r)
rows.sort(key=lambda r: -(r.get("hbm_read_bytes_per_launch_x2_corrected") or 0))
if os.environ.get("PMC_WAVELENGTHS"):   # PMC passes taken at another launch size (tools/profile_ts.sh)
    out["wavelengths_per_launch"] = int(os.environ["PMC_WAVELENGTHS"])
    out["pmc"] = rows
else:
    out["pmc_4096_wavelengths_per_launch"] = rows
print(json.dumps(out, indent=1))
