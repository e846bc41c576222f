#!/usr/bin/env python3
"""Read a rocprofv3 --kernel-trace CSV and report, for the last N seconds of kernels: wall time, time with at least one kernel
running (union of intervals), time with >= 2 running, and per-queue busy time.  Usage: trace_gaps.py <dir> [tail_fraction]"""
import csv
import glob
import sys

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - frac)
rows = [r for r in rows if r[0] >= t_lo]
wall = rows[-1][1] - rows[0][0]
ev = []
for s, e, q, n in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy1 = busy2 = 0
depth = 0
last = ev[0][0]
for t, dlt in ev:
    if depth >= 1:
        busy1 += t - last
    if depth >= 2:
        busy2 += t - last
    depth += dlt
    last = t
perq = {}
for s, e, q, n in rows:
    perq[q] = perq.get(q, 0) + (e - s)
print(f"kernels {len(rows)}  wall {wall/1e6:.3f} ms  >=1 running {busy1/1e6:.3f} ms ({100*busy1/wall:.1f} %)  >=2 running {busy2/1e6:.3f} ms ({100*busy2/wall:.1f} %)")
for q, b in sorted(perq.items()):
    print(f"  queue {q}: busy {b/1e6:.3f} ms ({100*b/wall:.1f} % of wall)")
byk = {}
for s, e, q, n in rows:
    k = n.split("(")[0][:60]
    byk[k] = byk.get(k, 0) + (e - s)
for k, b in sorted(byk.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {b/1e6:9.3f} ms  {k}")
