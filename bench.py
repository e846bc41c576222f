#!/usr/bin/env python3
"""Headline benchmark: FRVSR x4 180p -> 720p inference, SR frames/sec (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: 16 synthetic clips x 10 frames, 320x180 LR ->
1280x720 HR, advanced in lock-step on one MI355X (batch = clips, time = sequential).
180 rows are zero-padded to 184 (FNet needs multiples of 8),
the estimate is produced on the 736x1280 grid and only 720 rows count.
A "step" is one frame of all 16 clips = 16 SR frames.  Inputs are resident in HBM
before the timed region; the recurrent state never leaves the device.
For N > 1 (launched by torch.distributed.run) every rank runs its own 16 clips
(clip sharding, no data-path collective; "weak" scaling) and the time is the MAX
over ranks between two barriers.

Default precision is bf16x3: fp32 activations in HBM, every product computed as
three bf16 MFMAs on hi/lo-split operands with fp32 accumulation.  It meets the
fp32 parity bar (max-abs <= 1e-3, PSNR delta <= 0.01 dB; tests/test_gpu_step.py).
Plain bf16 and exact fp32 are measured too and reported under "other_dtypes".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "super-resolution_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

H_LR, W_LR, H_PAD = 180, 320, 184
CLIPS, FRAMES = 16, 10
FLOP_PER_LR_PX = 2210688.0            # 1,105,344 MAC per LR pixel per frame
RES_CONV_FLOP_PER_LR_PX = 73728.0     # one 64->64 3x3 conv: 36,864 MAC per LR pixel
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0}   # MI355X_MICROARCH.md dense peaks


def make_inputs(rank):
    """[FRAMES, CLIPS, 184, 320, 3] float32 in [0,1], rows 180..183 zero: per clip a smooth random
    image (bilinear blow-up of coarse noise) drifting by a few pixels per frame, plus fine noise."""
    lr = np.zeros((FRAMES, CLIPS, H_PAD, W_LR, 3), np.float32)
    for c in range(CLIPS):
        rng = np.random.default_rng(1000 * rank + c)
        coarse = rng.random((H_LR // 12 + 4, W_LR // 12 + 4, 3)).astype(np.float32)
        big = np.kron(coarse, np.ones((12, 12, 1), np.float32))          # piecewise constant ...
        k = np.ones(9, np.float32) / 9                                    # ... smoothed by a 9-tap box filter
        big = np.apply_along_axis(lambda v: np.convolve(v, k, "same"), 0, big)
        big = np.apply_along_axis(lambda v: np.convolve(v, k, "same"), 1, big)
        dy, dx = rng.integers(-2, 3, 2)
        for t in range(FRAMES):
            oy, ox = 12 + dy * t % 12, 12 + dx * t % 12
            frame = big[oy:oy + H_LR, ox:ox + W_LR] + 0.05 * rng.standard_normal((H_LR, W_LR, 3)).astype(np.float32)
            lr[t, c, :H_LR] = np.clip(frame, 0.0, 1.0)
    return lr


def make_weights(seed=7):
    """Seeded random weights in the reference's tensor layout (76 Keras-named tensors; demo.h5 is
    absent): Glorot-uniform kernels, small biases, the flow head scaled up and the last bias shifted
    so that flows reach a few tenths of a pixel and outputs spread over [0,1]."""
    import layers
    rng = np.random.default_rng(seed)
    out = {}
    for i, (name, shape) in enumerate(layers.weight_specs()):
        if len(shape) == 4:
            limit = np.sqrt(6.0 / (9 * shape[2] + 9 * shape[3]))
            w = rng.uniform(-limit, limit, size=shape)
        else:
            w = rng.uniform(-0.05, 0.05, size=shape)
        if i == 26:
            w = w * 8.0
        if i == 75:
            w = w + 0.45
        out[name] = w.astype(np.float32)
    return out


def run_steps(ctx, lr_dev, n, t0_frame=0):
    for i in range(n):
        f = (t0_frame + i) % FRAMES
        if f == 0:
            ctx.clip_begin(CLIPS)
        ctx.clip_push_dev(lr_dev[f].data_ptr())


def time_steps(torch, dist, ctx, lr_dev, steps, warmup, world):
    run_steps(ctx, lr_dev, warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(ctx, lr_dev, steps, t0_frame=warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def kernel_profile(torch, ctx, lr_dev, steps):
    """Per-kernel durations from HIP events recorded on the launch stream."""
    ctx.profile_reset()
    ctx.profile_enable(True)
    run_steps(ctx, lr_dev, steps)
    torch.cuda.synchronize()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    return prof


def pmc_traffic(dtype):
    """HBM bytes per launch of the res-block conv from the committed PMC passes (counters need
    their own rocprofv3 runs, so they cannot be read live here); None when no pass is on file."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")))
        tag = {"bf16x3": "conv3x3_lds_kernel<2, 2, 1, true>", "bf16": "conv3x3_lds_kernel<1, 2, 2, true>",
               "fp32": "conv3x3_lds_kernel<0, 2, 2, true>"}[dtype]
        best = None
        for k, v in d["kernels"].items():   # the trunk's grid is the one with the most launches
            if k.startswith(tag) and (best is None or v["launches"] > best["launches"]):
                best = v
        return best["hbm_bytes"] if best else None
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(weights, lr):
    """The torch-CPU restatement (oracle/ref_torch.py; kind 'port': the reference's
    TF1/Keras path cannot run here) on a bounded sample: 1 clip, 1 warm-up + 3 timed
    frames at 184x320.  The only place where bench.py touches oracle/."""
    import torch
    from oracle import ref_torch as rt
    tw = rt.TorchWeights(weights)
    clip = lr[:4, :1]
    with torch.no_grad():
        prev = torch.zeros(1, 3, H_PAD, W_LR)
        est = torch.zeros(1, 3, 4 * H_PAD, 4 * W_LR)
        times = []
        for t in range(4):
            cur = rt.to_nchw(clip[t])
            t0 = time.perf_counter()
            est, _ = rt.generator_step_t(tw, cur, prev, est)
            times.append(time.perf_counter() - t0)
            prev = cur
    dt = float(np.mean(times[1:]))
    return {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle/ref_torch.py (torch-CPU fp32, oneDNN), 1 clip, 3 timed frames after 1 warm-up, "
                      "184x320 -> 736x1280; the reference TF1/Keras CPU path cannot run here"}


def train_step_rate(weights, dtype, B=40, h=64, steps=5):
    """Extra line (not the headline): the native training step (SURVEY.md section 8 f rank 1) at the reference's
    config 4 -- train_on_batch on batch_size 4 x c_frames 10 = 40 samples of 64x64 LR (FRVSR_example.yaml,
    FRVSR.py:154-156), host numpy batch in, Adam step and weight re-pack included."""
    import sr_hip
    rng = np.random.default_rng(0)
    data = [rng.random((B, h, h, 3), dtype=np.float32), rng.random((B, h, h, 3), dtype=np.float32),
            rng.random((B, 4 * h, 4 * h, 3), dtype=np.float32), rng.random((B, 4 * h, 4 * h, 3), dtype=np.float32)]
    ctx = sr_hip.Context(h, h, max_batch=B, dtype=dtype)
    ctx.load_weights(weights)
    for _ in range(2):
        ctx.train_on_batch(*data)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = ctx.train_on_batch(*data)
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    # the whole loop body (FRVSR.py:103-156) from uint8 HR clips: LR synthesis, the no-gradient recurrence for the
    # previous estimates and the step, device-resident (sr_train_on_clips)
    clips = rng.integers(0, 256, (4, B // 4, 4 * h, 4 * h, 3), dtype=np.uint8)
    for _ in range(2):
        ctx.train_on_clips(clips)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.train_on_clips(clips)
    ctx.sync()
    dt_it = (time.perf_counter() - t0) / steps
    ctx.close()
    return {"workload": f"FRVSR train_on_batch, {B} samples of {h}x{h} LR -> {4 * h}x{4 * h} HR, Adam, host batch in",
            "dtype": dtype, "ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 1),
            "algorithmic_tflops": round(3 * FLOP_PER_LR_PX * h * h * B / dt / 1e12, 1), "loss": round(float(loss[0]), 5),
            "iteration_from_clips": {"workload": f"4 uint8 clips x {B // 4} frames in: LR synthesis + recurrence + step on the device",
                                     "ms_per_iteration": round(dt_it * 1e3, 3), "samples_per_s": round(B / dt_it, 1)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16x3", choices=["fp32", "bf16", "bf16x3"])
    ap.add_argument("--no-extras", action="store_true", help="skip other dtypes, kernel profile and CPU baseline")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libsr_hip has no CPU fallback")
    # one process per GPU; SR_BENCH_BACKEND=gloo lets several ranks rehearse on a one-GPU box
    backend = os.environ.get("SR_BENCH_BACKEND", "nccl")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import sr_hip
    weights = make_weights(7)
    lr = make_inputs(rank)
    lr_dev = torch.from_numpy(lr).cuda()
    stream = torch.cuda.current_stream().cuda_stream

    def make_ctx(dtype):
        ctx = sr_hip.Context(H_PAD, W_LR, max_batch=CLIPS, dtype=dtype, device=local_rank)
        ctx.load_weights(weights)
        ctx.set_stream(stream)
        return ctx

    ctx = make_ctx(args.dtype)
    dt = time_steps(torch, dist, ctx, lr_dev, args.steps, args.warmup, world)
    frames = world * CLIPS * args.steps
    fps = frames / dt
    out = {
        "metric": "SR frames/sec (FRVSR x4, 180p->720p)", "value": round(fps, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "FRVSR x4 inference, 16 synthetic clips x 10 frames 320x180 -> 1280x720 per GPU "
                               "(BASELINE.json configs[1]); LR zero-padded to 184 rows, state resident on device",
                   "clips_per_gpu": CLIPS, "frames_per_clip": FRAMES, "lr": [H_LR, W_LR], "hr": [720, 1280],
                   "weights": "seeded random init, demo.h5 tensor layout", "parallelism": f"clip-shard x{world}"},
        "whole_step_tflops": round(fps * FLOP_PER_LR_PX * H_LR * W_LR / 1e12, 2),
    }

    if rank == 0 and not args.no_extras:
        # dominant kernel: the 64->64 3x3 res-block conv (20 launches per step, 66.7 % of the FLOPs)
        prof = kernel_profile(torch, ctx, lr_dev, max(2, min(args.steps, 10)))
        res = next(p for p in prof if p["name"] == "srnet.res")
        avg_ms = res["ms_total"] / res["launches"]
        alg_flops = RES_CONV_FLOP_PER_LR_PX * CLIPS * H_LR * W_LR      # credited on the unpadded 180 rows
        achieved = alg_flops / (avg_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        out["roofline"] = {"bound": "mfma", "kernel": "conv3x3_lds_kernel (srnet.res, 64->64 3x3, 20 launches/step)",
                           "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(achieved / peak, 4), "traffic": pmc_traffic(args.dtype),
                           "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate "
                                           "passes, profiles/r1_hbm_traffic.json); algorithmic 723 MB",
                           "avg_launch_ms": round(avg_ms, 4), "launches_timed": res["launches"]}
        tot = sum(p["ms_total"] for p in prof)
        out["kernel_ms_per_step"] = {p["name"]: round(p["ms_total"] / max(1, res["launches"] // 20), 4) for p in prof}
        out["kernel_ms_per_step"]["_sum"] = round(tot / max(1, res["launches"] // 20), 4)
        pre = next(p for p in prof if p["name"] == "pre_srnet")
        pre_ms = pre["ms_total"] / pre["launches"]
        out["hbm_kernel"] = {"kernel": "pre_srnet_kernel", "bound": "hbm",
                             "achieved": round(pre["bytes_per_launch"] / (pre_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                             "unit": "GB/s", "avg_launch_ms": round(pre_ms, 4)}
        tail = next((p for p in prof if p["name"] == "srnet.out"), None)
        if tail:   # the 9-tap gather after the fused transposed conv: 128 B in + 12 B out per HR pixel
            t_ms = tail["ms_total"] / tail["launches"]
            out["hbm_kernel2"] = {"kernel": "tap_gather_kernel (srnet.out)", "bound": "hbm",
                                  "achieved": round(tail["bytes_per_launch"] / (t_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                                  "unit": "GB/s", "avg_launch_ms": round(t_ms, 4)}
        ctx.close()
        others = {}
        for d in ("fp32", "bf16", "bf16x3"):
            if d == args.dtype or world > 1:
                continue
            c2 = make_ctx(d)
            t2 = time_steps(torch, dist, c2, lr_dev, max(2, args.steps // 2), 2, 1)
            others[d] = round(CLIPS * max(2, args.steps // 2) / t2, 2)
            c2.close()
        out["other_dtypes"] = {"unit": "frames/s", **others}
        if world == 1:
            out["train_step"] = train_step_rate(weights, args.dtype if args.dtype != "bf16" else "bf16x3")
            out["cpu_baseline"] = cpu_baseline(weights, lr)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
