#!/usr/bin/env python3
"""Time single conv layers of the 180p workload (16 clips, 184x320 LR) on the GPU.

    python tools/conv_bench.py [--dtype bf16x3] [--iters 20] [--only res]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import sr_hip  # noqa: E402

PEAK = {"fp32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0}
LAYERS = {  # name: (H, W, Cin, Cout, transposed, residual)
    "res": (184, 320, 64, 64, False, True),
    "head": (184, 320, 51, 64, False, False),
    "up1": (184, 320, 64, 64, True, False),
    "up2": (368, 640, 64, 64, True, False),
    "out": (736, 1280, 64, 3, False, False),
    "fnet2": (184, 320, 32, 32, False, False),
    "fnet8": (23, 40, 256, 256, False, False),
    "fnet13": (184, 320, 64, 32, False, False),
}

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16x3")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
for name, (H, W, ci, co, tr, res) in LAYERS.items():
    if a.only and name not in a.only.split(","):
        continue
    for dt in a.dtype.split(","):
        ms = sr_hip.bench_conv(a.batch, H, W, ci, co, dtype=dt, transposed=tr, residual=res, iters=a.iters)
        fl = a.batch * H * W * 9 * ci * co * 2
        tf = fl / ms / 1e9
        print(f"{name:7s} {dt:7s} {ms:8.4f} ms  {tf:8.1f} TFLOP/s algorithmic  ({100 * tf / PEAK[dt]:.1f}% of {PEAK[dt]:.0f})", flush=True)
