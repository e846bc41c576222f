#!/usr/bin/env python3
"""Headline benchmark: FRVSR x4 180p -> 720p inference, SR frames/sec (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: 16 synthetic clips x 10 frames per GPU, 320x180 LR -> 1280x720 HR,
advanced in lock-step on one MI355X (batch = clips, time = sequential).  180 rows are zero-padded to 184
(FNet needs multiples of 8), the estimate is produced on the 736x1280 grid and
only 720 rows count.  A "step" is one frame of all 16 clips = 16 SR frames.  Inputs (synth.synth_clip,
seed 1000 + global clip id: the clips the parity tests use) are resident in HBM before the timed region;
the recurrent state never leaves the device.

--gpus N: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this
process is one rank; called plainly with N > 1 it starts the N ranks itself (fresh child processes, before
anything here touches the GPU) and relays rank 0's JSON line.  Every rank runs its own clips (clip
sharding, no data-path collective); RCCL carries only the barrier and the MAX over ranks of the time.
Default: 16 clips per GPU ("weak").  --clips-total C (BASELINE configs[2]: 128): C clips over the N ranks in
contiguous blocks, each rank advancing its block in lock-step batches of 16 ("strong").

Default precision is f16mx6: bf16x3 -- 32-bit activations in HBM, every product computed as three bf16
MFMAs on hi/lo-split operands with fp32 accumulation -- except in the 20 convs of SRNet's residual blocks, where the product
is one f16 MFMA plus two MX-scaled FP6 MFMAs for the cross terms and the 32 bits of an activation are f16 + an FP6 residue.
Both meet the fp32 parity bar (max-abs <= 1e-3, PSNR delta <= 0.01 dB; tests/test_gpu_bench_config.py runs this exact
configuration for every dtype: f16mx6 1e-4, bf16x3 4e-5 against the oracle).  bf16x3, plain bf16 and exact fp32 are measured
too and reported under "other_dtypes" on the same box.  After the timed region the first frames of clip 0 are compared with
the CPU oracle ("verified").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "super-resolution_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

H_LR, W_LR, H_PAD = 180, 320, 184
BATCH, FRAMES = 16, 10                # clips advanced in lock-step by one context; frames per clip
FLOP_PER_LR_PX = 2210688.0            # 1,105,344 MAC per LR pixel per frame
RES_CONV_FLOP_PER_LR_PX = 73728.0     # one 64->64 3x3 conv: 36,864 MAC per LR pixel
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0, "f16mx6": 2500.0}   # MI355X_MICROARCH.md dense peaks (f16 = bf16)
VERIFY_TOL = {"fp32": 1e-3, "bf16x3": 1e-3, "f16mx6": 1e-3, "bf16": 0.15}


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (nothing in this process has touched
    the GPU), wait for them, relay rank 0's output.  Returns the exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def make_inputs(first_clip, n_clips):
    """[FRAMES, n_clips, 184, 320, 3] float32 in [0,1], rows 180..183 zero."""
    import synth
    lr = np.zeros((FRAMES, n_clips, H_PAD, W_LR, 3), np.float32)
    for c in range(n_clips):
        lr[:, c, :H_LR] = synth.synth_clip(first_clip + c, FRAMES, H_LR, W_LR)
    return lr


def run_steps(ctxs, lr_devs, n, t0_frame=0):
    for i in range(n):
        f = (t0_frame + i) % FRAMES
        for ctx, lr_dev in zip(ctxs, lr_devs):
            if f == 0:
                ctx.clip_begin(BATCH)
            ctx.clip_push_dev(lr_dev[f].data_ptr())


def time_steps(torch, dist, ctxs, lr_devs, steps, warmup, world):
    run_steps(ctxs, lr_devs, warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(ctxs, lr_devs, steps, t0_frame=warmup)
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
    run_steps([ctx], [lr_dev], steps)
    torch.cuda.synchronize()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    return prof


# the kernels behind `srnet.res` per dtype, as rocprofv3 names them (prefix match; tools/pmc_summary.py writes the same map
# into the traffic file as `trunk_kernel`).  bf16x3: the ReLU and the residual instantiation of conv_trunk.hip alternate,
# the reported figure is their mean.
TRUNK_KERNELS = {"bf16x3": ["trunk_conv_kernel<false, true, 0, false>", "trunk_conv_kernel<true, false, 0, false>"],
                 "f16mx6": ["trunk6_conv_kernel<false, true, 0, false>", "trunk6_conv_kernel<true, false, 0, false>"],
                 "bf16": ["conv3x3_lds_kernel<1, 2, 2, true"], "fp32": ["conv3x3_lds_kernel<0, 2, 2, true"]}


def newest_traffic_file(profiles_dir):
    """profiles/rN_hbm_traffic.json with the largest N, or None."""
    best = None
    for f in os.listdir(profiles_dir) if os.path.isdir(profiles_dir) else []:
        if f.startswith("r") and f.endswith("_hbm_traffic.json") and f[1:-len("_hbm_traffic.json")].isdigit():
            n = int(f[1:-len("_hbm_traffic.json")])
            if best is None or n > best[0]:
                best = (n, f)
    return best[1] if best else None


def pmc_traffic(dtype, profiles_dir=None):
    """HBM bytes per launch of the res-block conv from the NEWEST committed PMC pass (counters need their own
    rocprofv3 runs, so they cannot be read live here).  Returns (bytes, file, note): bytes is None -- never an
    older round's figure for another kernel -- when that file does not hold the kernels timed here."""
    profiles_dir = profiles_dir or os.path.join(ROOT, "profiles")
    name = newest_traffic_file(profiles_dir)
    if not name:
        return None, None, "no profiles/rN_hbm_traffic.json on file"
    try:
        d = json.load(open(os.path.join(profiles_dir, name)))
        tags = d.get("trunk_kernel", {}).get(dtype) or TRUNK_KERNELS[dtype]
        tags = tags if isinstance(tags, list) else [tags]
        found = []
        for t in tags:
            best = None
            for k, v in d["kernels"].items():   # the trunk's grid is the one with the most launches
                if k.startswith(t) and (best is None or v["launches"] > best["launches"]):
                    best = v
            if best is None:
                return None, name, f"{name} holds no kernel named {t}*"
            found.append(best["hbm_bytes"])
        return round(sum(found) / len(found)), name, f"mean over {len(found)} kernel(s): " + ", ".join(tags)
    except (OSError, KeyError, ValueError) as e:
        return None, name, f"{name}: {type(e).__name__}: {e}"


def cpu_baseline(weights, lr0):
    """The torch-CPU restatement (oracle/ref_torch.py; kind 'port': the reference's TF1/Keras path cannot run
    here) on a bounded sample of the same workload: clip 0, 184x320, 1 warm-up + 3 timed frames per run,
    median of 3 runs on all host threads, plus one timed frame on a single thread.  The only place where
    bench.py touches oracle/; its frames double as the reference of the GPU self-check ("verified")."""
    import torch
    from oracle import ref_torch as rt
    tw = rt.TorchWeights(weights)

    def run(nframes):
        times, outs = [], []
        with torch.no_grad():
            prev = torch.zeros(1, 3, H_PAD, W_LR)
            est = torch.zeros(1, 3, 4 * H_PAD, 4 * W_LR)
            for t in range(nframes):
                cur = rt.to_nchw(lr0[t])
                t0 = time.perf_counter()
                est, _ = rt.generator_step_t(tw, cur, prev, est)
                times.append(time.perf_counter() - t0)
                outs.append(rt.to_nhwc(est))
                prev = cur
        return times, outs

    threads = torch.get_num_threads()
    rates, frames = [], None
    for _ in range(3):
        times, frames = run(4)
        rates.append(3.0 / sum(times[1:]))
    torch.set_num_threads(1)
    t1, _ = run(2)
    torch.set_num_threads(threads)
    return {"value": round(float(np.median(rates)), 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "runs": [round(r, 4) for r in rates], "single_thread": round(1.0 / t1[1], 4),
            "sample": "oracle/ref_torch.py (torch-CPU fp32, oneDNN), clip 0 at 184x320 -> 736x1280: median of 3 runs of "
                      "1 warm-up + 3 timed frames on all threads; single_thread = 1 timed frame after 1 warm-up; "
                      "the reference TF1/Keras CPU path cannot run here"}, frames


def gpu_first_frames(torch, ctx, lr_dev, n):
    """Estimates of clip 0 for frames 0..n-1 from a fresh recurrence of the whole batch, [n, 736, 1280, 3]."""
    out = torch.empty((BATCH, 4 * H_PAD, 4 * W_LR, 3), dtype=torch.float32, device="cuda")
    ctx.clip_begin(BATCH)
    got = []
    for t in range(n):
        ctx.clip_push_dev(lr_dev[t].data_ptr(), out.data_ptr())
        torch.cuda.synchronize()
        got.append(out[0].cpu().numpy())
    return np.stack(got)


def end_to_end(torch, sr_hip, weights, lr, dtype, device, steps):
    """The reference's own timing bracket (FRVSR.py:292-296: host arrays in, host arrays out) plus its
    utils.denormalize: every step copies the 16 LR frames (float32, 180 rows: the ctx pads on the device) from
    page-locked host memory and brings the 16 estimates back as uint8 720p frames, the copy of frame t
    running under the kernels of frame t+1."""
    ctx = sr_hip.Context(H_LR, W_LR, max_batch=BATCH, dtype=dtype, device=device, pad=True)
    ctx.load_weights(weights)
    lr_host = torch.from_numpy(np.ascontiguousarray(lr[:, :, :H_LR])).pin_memory()
    outs = [torch.empty((BATCH, 4 * H_LR, 4 * W_LR, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]

    def run(n, f0=0):
        for i in range(n):
            f = (f0 + i) % FRAMES
            if f == 0:
                ctx.clip_begin(BATCH)
            ctx.clip_push_u8(lr_host[f].data_ptr(), outs[i % 2].data_ptr())
            ctx.u8_wait(1)      # frame i-1 is in host memory: this is where a caller would consume it
        ctx.u8_wait(0)

    run(FRAMES)
    ctx.sync()
    t0 = time.perf_counter()
    run(steps)
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.close()
    return {"value": round(BATCH * steps / dt, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps,
            "what": "per step: 16 LR frames float32 H2D from page-locked memory (11 MB) + generator step + 16 uint8 720p "
                    "frames D2H (44 MB) on a copy stream; padded 180-row ctx (sr_create_padded + sr_clip_push_u8)"}


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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="f16mx6", choices=["fp32", "bf16", "bf16x3", "f16mx6"])
    ap.add_argument("--clips-total", type=int, default=0,
                    help="total clips over all ranks (BASELINE configs[2]: 128), in lock-step batches of 16 per rank; "
                         "default 0 = 16 clips per GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip other dtypes, kernel profile, end-to-end and CPU baseline")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only (gloo, no GPU): every rank joins, rank 0 prints the rank count; tests the launcher")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.clips_total and args.clips_total % (world * BATCH):
        raise SystemExit(f"--clips-total must be a multiple of {BATCH} x the number of GPUs")

    import torch
    import torch.distributed as dist
    if args.launch_check:
        n = world
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.ones(1)
            dist.all_reduce(t)
            n = int(t.item())
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": n, "world_size_env": world}), flush=True)
        return
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
    import synth
    weights = synth.make_weights(7)
    per_rank = args.clips_total // world if args.clips_total else BATCH
    nbatch = per_rank // BATCH
    lr = make_inputs(rank * per_rank, per_rank)                         # clip ids are global: rank r owns a contiguous block
    lr_devs = [torch.from_numpy(np.ascontiguousarray(lr[:, b * BATCH:(b + 1) * BATCH])).cuda() for b in range(nbatch)]
    stream = torch.cuda.current_stream().cuda_stream

    def make_ctx(dtype):
        ctx = sr_hip.Context(H_PAD, W_LR, max_batch=BATCH, dtype=dtype, device=local_rank)
        ctx.load_weights(weights)
        ctx.set_stream(stream)
        return ctx

    ctxs = [make_ctx(args.dtype) for _ in range(nbatch)]
    dt = time_steps(torch, dist, ctxs, lr_devs, args.steps, args.warmup, world)
    frames = world * per_rank * args.steps
    fps = frames / dt
    out = {
        "metric": "SR frames/sec (FRVSR x4, 180p->720p)", "value": round(fps, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
        "scaling": "strong" if args.clips_total else "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"FRVSR x4 inference, {per_rank} synthetic clips x {FRAMES} frames 320x180 -> 1280x720 per GPU "
                               f"in lock-step batches of {BATCH} (BASELINE.json configs[{2 if args.clips_total else 1}]); "
                               "LR zero-padded to 184 rows, state resident on device",
                   "clips_per_gpu": per_rank, "clips_total": world * per_rank, "frames_per_clip": FRAMES,
                   "lr": [H_LR, W_LR], "hr": [720, 1280], "inputs": "synth.synth_clip(seed 1000 + clip id)",
                   "weights": "seeded random init, demo.h5 tensor layout", "parallelism": f"clip-shard x{world}"},
        "whole_step_tflops": round(fps * FLOP_PER_LR_PX * H_LR * W_LR / 1e12, 2),
    }
    if world > 1:
        out["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else 0
        out["collective_backend"] = dist.get_backend()

    ctx = ctxs[0]
    for c in ctxs[1:]:
        c.close()
    if rank == 0 and not args.no_extras:
        # dominant kernel: the 64->64 3x3 res-block conv (20 launches per step, 66.7 % of the FLOPs)
        prof = kernel_profile(torch, ctx, lr_devs[0], max(2, min(args.steps, 10)))
        res = next(p for p in prof if p["name"] == "srnet.res")
        convs_per_launch = max(1.0, res["flops_per_launch"] / (RES_CONV_FLOP_PER_LR_PX * BATCH * H_PAD * W_LR))
        avg_ms = res["ms_total"] / res["launches"]
        alg_flops = RES_CONV_FLOP_PER_LR_PX * BATCH * H_LR * W_LR * round(convs_per_launch)   # credited on the unpadded 180 rows
        achieved = alg_flops / (avg_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        traffic, traffic_src, traffic_note = pmc_traffic(args.dtype)
        # algorithmic bytes per launch: 256 B in + 256 B out per LR pixel (64 channels x 4 B), + 256 B of residual on every
        # second launch -- on the padded 184 rows the kernel really moves, like `traffic`
        alg_bytes = round(BATCH * H_PAD * W_LR * 256 * 2.5 * round(convs_per_launch))
        out["roofline"] = {"bound": "mfma", "kernel": "srnet.res: the SRNet trunk's 64->64 3x3 convs "
                                                       f"({round(convs_per_launch)} conv(s) per launch, {res['launches'] // max(1, min(args.steps, 10))} launches/step)",
                           "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(achieved / peak, 4), "traffic": traffic, "algorithmic_bytes": alg_bytes,
                           "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
                           "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes, "
                                           f"profiles/{traffic_src}); algorithmic_bytes = mean of ReLU (512 B/px) and residual (768 B/px) launches",
                           "traffic_note": traffic_note,
                           "avg_launch_ms": round(avg_ms, 4), "launches_timed": res["launches"]}
        nsteps_prof = max(2, min(args.steps, 10))
        tot = sum(p["ms_total"] for p in prof)
        out["kernel_ms_per_step"] = {p["name"]: round(p["ms_total"] / nsteps_prof, 4) for p in prof}
        out["kernel_ms_per_step"]["_sum"] = round(tot / nsteps_prof, 4)
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
        if world == 1:
            gpu_frames = gpu_first_frames(torch, ctx, lr_devs[0], 4)
        ctx.close()
        others = {}
        for d in ("fp32", "bf16", "bf16x3", "f16mx6"):
            if d == args.dtype or world > 1:
                continue
            c2 = make_ctx(d)
            n2 = max(2, args.steps // 4)
            t2 = time_steps(torch, dist, [c2], lr_devs[:1], n2, 2, 1)
            others[d] = round(BATCH * n2 / t2, 2)
            c2.close()
        out["other_dtypes"] = {"unit": "frames/s", **others}
        if world == 1:
            out["end_to_end"] = end_to_end(torch, sr_hip, weights, lr[:, :BATCH], args.dtype, local_rank, max(FRAMES, args.steps // 2))
            out["train_step"] = train_step_rate(weights, args.dtype if args.dtype in ("fp32", "bf16x3") else "bf16x3")
            # BASELINE.json configs[4] (TecoGAN adversarial step), one GPU: the reference's 64-sample batch at its own 32x32 LR
            # crops and a smaller batch at the 128x128 LR crops the config names (tools/gan_bench.py)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import gan_bench
            tdt = args.dtype if args.dtype in ("fp32", "bf16x3") else "bf16x3"
            out["tecogan_step"] = [gan_bench.run(64, 32, tdt, 3, local_rank), gan_bench.run(8, 128, tdt, 2, local_rank)]
            out["cpu_baseline"], cpu_frames = cpu_baseline(weights, lr[:, :1])
            err = float(np.abs(gpu_frames - np.stack([f[0] for f in cpu_frames])).max())
            out["verified"] = bool(err <= VERIFY_TOL[args.dtype])
            out["verify"] = {"max_abs_err": round(err, 7), "tol": VERIFY_TOL[args.dtype], "clip": 0, "frames": 4,
                             "against": "oracle/ref_torch.py frames of the cpu_baseline leg (same inputs, same weights)"}
            if not out["verified"]:
                print(json.dumps(out), flush=True)
                raise SystemExit(f"bench.py: GPU output differs from the CPU oracle by {err:.3e} (> {VERIFY_TOL[args.dtype]})")
    else:
        ctx.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
