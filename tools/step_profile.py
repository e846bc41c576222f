#!/usr/bin/env python3
"""The benched step (bench.py's workload and timing) without the extras: frames/s of N timed steps, then the
per-kernel milliseconds of a step from HIP events on the launch stream (one lane, every kernel alone on the device).
For A/B runs of environment switches:   SR_NO_PERSIST_UP2=1 python tools/step_profile.py"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts the package directory on sys.path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16x3")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--e2e", action="store_true", help="also bench.py's end_to_end leg (host float32 frames in, uint8 frames out)")
    args = ap.parse_args()
    import torch
    import sr_hip
    import synth
    torch.cuda.set_device(0)
    weights = synth.make_weights(7)
    lr = bench.make_inputs(0, bench.BATCH)
    lr_dev = torch.from_numpy(np.ascontiguousarray(lr)).cuda()
    ctx = sr_hip.Context(bench.H_PAD, bench.W_LR, max_batch=bench.BATCH, dtype=args.dtype, device=0)
    ctx.load_weights(weights)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dt = bench.time_steps(torch, None, [ctx], [lr_dev], args.steps, args.warmup, 1)
    out = {"frames_per_s": round(bench.BATCH * args.steps / dt, 1), "ms_per_step": round(1e3 * dt / args.steps, 4)}
    if not args.no_profile:
        n = 10
        prof = bench.kernel_profile(torch, ctx, lr_dev, n)
        out["kernel_ms_per_step"] = {p["name"]: round(p["ms_total"] / n, 4) for p in prof}
        out["kernel_ms_per_step"]["_sum"] = round(sum(p["ms_total"] for p in prof) / n, 4)
    ctx.close()
    if args.e2e:
        out["end_to_end"] = bench.end_to_end(torch, sr_hip, weights, lr, args.dtype, 0, 50)
        out["end_to_end"].pop("what")
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
