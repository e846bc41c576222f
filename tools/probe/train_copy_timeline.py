"""One training step of a rocprofv3 --kernel-trace --memory-copy-trace run: the copies and the first kernel of every phase, relative to fnet_input."""
import csv
import glob
import sys

d = sys.argv[1]
cp = list(csv.DictReader(open(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0])))
kn = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY") for r in cp if r["Direction"].endswith("HOST_TO_DEVICE")]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sr::", "")[:40]) for r in kn]
ev.sort()
idx = [i for i, e in enumerate(ev) if "fnet_input" in e[2]]
i0, i1 = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = ev[i0][0]
print(f"step length {(ev[i1][0] - t0) / 1e3:.0f} us")
for s, e, name in ev[max(0, i0 - 40):i1]:
    if name == "COPY" or any(k in name for k in ("fnet_input", "warp_generic", "mse_grad", "adam", "repack")):
        if (e - s) > 5000 or name != "COPY":
            print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:6.1f})  {name}")
