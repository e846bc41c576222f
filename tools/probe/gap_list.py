"""Idle gaps of a rocprofv3 --kernel-trace run: python gap_list.py <dir> [min_us]; prints the largest gaps of the second half with the kernels around them."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sr::", "")[:60]))
rows.sort()
rows = rows[len(rows) // 2:]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
end = rows[0][1]
prev = rows[0][2]
gaps = []
for s, e, n in rows[1:]:
    if s > end and (s - end) / 1e3 >= min_us:
        gaps.append(((s - end) / 1e3, prev, n))
    if e > end:
        end, prev = e, n
tot = sum(g[0] for g in gaps)
print(f"{len(gaps)} gaps >= {min_us} us, total {tot / 1e3:.3f} ms over span {(rows[-1][1] - rows[0][0]) / 1e6:.3f} ms")
agg = {}
for g, a, b in gaps:
    k = (a, b)
    agg.setdefault(k, [0, 0.0])
    agg[k][0] += 1
    agg[k][1] += g
for (a, b), (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {cnt:4d} x {t / cnt:8.1f} us  after {a}  before {b}")
