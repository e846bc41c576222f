#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into the small files kept under profiles/.

    python tools/pmc_summary.py stats   <dir with *_kernel_stats.csv>  > profiles/rN_..._kernel_stats.csv
    python tools/pmc_summary.py traffic <fetch dir> <write dir> <out.json>
    python tools/pmc_summary.py durations <dir with *kernel_trace.csv>  > profiles/rN_..._launch_durations.csv
    python tools/pmc_summary.py counters <dir> <counter> [<counter> ...]  (per kernel and grid: mean per launch)

`durations` groups the kernel trace by (kernel, grid size): rocprofv3's own --stats summary groups by name only, which
mixes e.g. the whole-batch and the sub-batch launches of one kernel.

`traffic` reads two separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950: 3 + 2 of the 4 TCC slots) and writes per-kernel HBM bytes per launch.  Corrections follow
MI355X_MICROARCH.md (HBM): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE
counts a 128-byte request as 64 bytes for wide coalesced streaming reads, so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]            # drop the argument list, keep the template arguments
    return name.replace("sr::", "")


def per_kernel(path, counter):
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    seen = set()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = (f, r["Dispatch_Id"])
            name = short(r["Kernel_Name"]) + " @grid" + r["Grid_Size"]
            if key in seen:          # one row per dimension instance: sum them
                acc[name][0] += float(r["Counter_Value"])
                continue
            seen.add(key)
            a = acc[name]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / max(1, v[1]), v[1]) for k, v in acc.items()}


def main():
    if sys.argv[1] == "stats":
        files = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_stats.csv"), recursive=True)
        for f in files:
            sys.stdout.write(open(f).read())
        return
    if sys.argv[1] == "durations":
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                grid = r.get("Grid_Size") or str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
                acc[short(r["Kernel_Name"]) + " @grid" + grid].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print("kernel @grid,launches,avg_us,min_us,max_us,total_ms")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            print(f'"{k}",{len(v)},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f},{sum(v) / 1e3:.3f}')
        return
    if sys.argv[1] == "counters":
        for ctr in sys.argv[3:]:
            for k, (v, n) in sorted(per_kernel(sys.argv[2], ctr).items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                print(f"{ctr:32s} {k[:90]:90s} launches {n:5d} mean per launch {v:.4g}")
        return
    fetch = per_kernel(sys.argv[2], "FETCH_SIZE")
    write = per_kernel(sys.argv[3], "WRITE_SIZE")
    out = {"_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); "
                    "FETCH_SIZE KiB doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B), "
                    "WRITE_SIZE KiB as reported", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fk, n = fetch.get(k, (0.0, 0))
        wk, _ = write.get(k, (0.0, 0))
        out["kernels"][k] = {"launches": n, "fetch_bytes": round(fk * 1024 * 2), "write_bytes": round(wk * 1024),
                             "hbm_bytes": round(fk * 1024 * 2 + wk * 1024)}
    # the kernels behind bench.py's roofline line (`srnet.res`), so that the bench and this file cannot drift apart
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    try:
        from bench import TRUNK_KERNELS
        out["trunk_kernel"] = {d: t for d, t in TRUNK_KERNELS.items() if all(any(k.startswith(x) for k in out["kernels"]) for x in t)}
    except ImportError:
        pass
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
