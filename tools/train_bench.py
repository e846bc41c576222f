"""Times the native FRVSR training step (sr_train_on_batch) on synthetic data: reference config 4 is
batch_size 4 clips x c_frames 10 = 40 samples of 64x64 LR / 256x256 HR (FRVSR_example.yaml, FRVSR.py:154-156).
Prints samples/s, ms per step and the per-kernel-class breakdown from the ctx profiler.

    python tools/train_bench.py [--batch 40] [--lr 64] [--dtype bf16x3|fp32] [--steps 10]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import sr_hip  # noqa: E402
from layers import glorot_init  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=40)
    ap.add_argument("--lr", type=int, default=64)
    ap.add_argument("--dtype", default="bf16x3")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    B, h = a.batch, a.lr
    rng = np.random.default_rng(0)
    data = [rng.random((B, h, h, 3), dtype=np.float32), rng.random((B, h, h, 3), dtype=np.float32),
            rng.random((B, 4 * h, 4 * h, 3), dtype=np.float32), rng.random((B, 4 * h, 4 * h, 3), dtype=np.float32)]
    ctx = sr_hip.Context(h, h, max_batch=B, dtype=a.dtype)
    ctx.load_weights(glorot_init(7))
    for _ in range(2):
        loss = ctx.train_on_batch(*data)
    if a.profile:
        ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = ctx.train_on_batch(*data)
    ctx.sync()
    dt = (time.perf_counter() - t0) / a.steps
    flop = 3 * 2210688.0 * h * h * B          # forward + data gradient + weight gradient, SURVEY.md 8 d per LR pixel
    print(f"train step B={B} {h}x{h} {a.dtype}: {dt * 1e3:.2f} ms/step incl. H2D of the batch, {B / dt:.0f} samples/s, "
          f"~{flop / dt / 1e12:.1f} TFLOP/s algorithmic (3x forward), loss {loss[0]:.5f}")
    if a.profile:
        rows = sorted(ctx.profile_read(), key=lambda r: -r["ms_total"])
        tot = sum(r["ms_total"] for r in rows)
        for r in rows:
            tf = r["flops_per_launch"] * r["launches"] / max(r["ms_total"], 1e-9) / 1e9
            print(f"  {r['name']:18s} {r['ms_total'] / a.steps:8.3f} ms/step  {100 * r['ms_total'] / tot:5.1f} %  "
                  f"{r['launches'] // a.steps:4d} launches/step  {tf:7.1f} TFLOP/s")
        print(f"  kernels total {tot / a.steps:.2f} ms/step")


if __name__ == "__main__":
    main()
