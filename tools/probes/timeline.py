"""Concurrency of the kernels of one bench step from a rocprofv3 kernel trace (CSV): per kernel the busy time, the union of all
kernels' busy intervals, the time with 1 / 2 / 3+ kernels in flight, and the idle gaps.  Usage: timeline.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * skip      # the last part of the run: the timed steps
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
pts = []
for s, e, _ in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
level, last, hist = 0, pts[0][0], defaultdict(int)
for t, d in pts:
    hist[min(level, 4)] += t - last
    level += d
    last = t
busy = defaultdict(int)
for s, e, n in ev:
    busy[n.split("(")[0][-40:]] += e - s
print(f"window {span / 1e6:.1f} ms, kernels {len(ev)}")
for k in sorted(hist):
    print(f"  {k}{'+' if k == 4 else ''} kernels in flight: {hist[k] / 1e6:8.2f} ms ({100 * hist[k] / span:.1f} %)")
for n, t in sorted(busy.items(), key=lambda kv: -kv[1])[:8]:
    print(f"  {n:42s} {t / 1e6:8.2f} ms")
