"""Times the native TecoGAN adversarial iteration (reference TecoGAN.py:434-479) on synthetic data: two
discriminator.train_on_batch, one discriminator.predict, two tecogan.train_on_batch (generator forward + discriminator
forward / backward + generator backward, Adam on both).  Reference config 5 (TecoGAN_example.yaml): batch_size 4 clips x
c_frames 10, ping-pong doubled, first / last frame dropped = 64 samples; BASELINE.json names 128x128 LR crops.

    python tools/gan_bench.py [--batch 64] [--lr 32] [--dtype bf16x3|fp32] [--iters 5]
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
from layers import TECOGAN_LOSS_WEIGHTS, DiscriminatorModel, glorot_init  # noqa: E402


def run(B, h, dtype, iters, device=0):
    H = 4 * h
    rng = np.random.default_rng(0)
    lr = [rng.random((B, h, h, 3), dtype=np.float32) for _ in range(2)]
    hr = [rng.random((B, H, H, 3), dtype=np.float32) for _ in range(9)]
    ctx = sr_hip.Context(h, h, max_batch=B, dtype=dtype, model="tecogan", device=device)
    ctx.load_weights(glorot_init(7))
    disc = sr_hip.Discriminator(H, H, max_batch=B, dtype=dtype, device=device)
    disc.load_weights(DiscriminatorModel((H, H, 3), seed=3).tensors)
    ones, zeros = np.ones(B, np.float32), np.zeros(B, np.float32)
    t = 0
    parts = {"disc_step": 0.0, "disc_predict": 0.0, "tecogan_step": 0.0}
    losses = None
    for it in range(iters + 1):
        if it == 1:
            parts = {k: 0.0 for k in parts}
            t_all = time.perf_counter()
        t0 = time.perf_counter()
        for k, lab in enumerate((ones, zeros)):
            disc.train_grads(hr[:9], lab)
            t += 1
            disc.apply(t, reset_moments=(k == 0))
        disc.sync()
        t1 = time.perf_counter()
        feats = disc.predict(hr[:9])[1:]
        t2 = time.perf_counter()
        for _ in range(2):
            losses = sr_hip.tecogan_grads(ctx, disc, lr[0], lr[1], hr[0], hr[1], hr[2], hr[3], hr[4], hr[5], hr[6], feats, hr[6][::-1].copy(),
                                          TECOGAN_LOSS_WEIGHTS)
            t += 1
            sr_hip.tecogan_apply(ctx, disc, t)
        ctx.sync()
        t3 = time.perf_counter()
        parts["disc_step"] += (t1 - t0) / 2
        parts["disc_predict"] += t2 - t1
        parts["tecogan_step"] += (t3 - t2) / 2
    total = (time.perf_counter() - t_all) / iters
    # the same pass as ONE native call (sr_tecogan_iteration): every array uploaded once, feature-map targets stay on the device
    batch = {"lr": lr[0], "prev_lr": lr[1], "hr": hr[6], "prev_hr": hr[0], "next_hr": hr[1], "prev_hat_hr": hr[2], "next_hat_hr": hr[3],
             "est": hr[4], "prev_est": hr[5], "next_est": hr[7], "prev_hat_est": hr[8], "next_hat_est": hr[0], "bic": hr[1], "prev_bic": hr[2],
             "next_bic": hr[3]}
    one = []
    for it in range(iters + 1):
        t0 = time.perf_counter()
        sr_hip.tecogan_iteration(ctx, disc, batch, TECOGAN_LOSS_WEIGHTS, t)
        t += 4
        one.append(time.perf_counter() - t0)
    one_call = sum(one[1:]) / iters
    ctx.close()
    disc.close()
    return {"workload": f"TecoGAN adversarial iteration, {B} samples of {h}x{h} LR -> {H}x{H} HR, host arrays in (2 D steps, predict, 2 combined steps)",
            "dtype": dtype, "ms_per_iteration": round(total * 1e3, 2), "samples_per_s": round(B / total, 1),
            "ms_disc_step": round(parts["disc_step"] / iters * 1e3, 2), "ms_disc_predict": round(parts["disc_predict"] / iters * 1e3, 2),
            "ms_tecogan_step": round(parts["tecogan_step"] / iters * 1e3, 2), "total_loss": round(float(losses[0]), 5),
            "one_native_call": {"what": "sr_tecogan_iteration: the same pass, every array uploaded once", "ms_per_iteration": round(one_call * 1e3, 2),
                                "samples_per_s": round(B / one_call, 1)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--lr", type=int, default=32)
    ap.add_argument("--dtype", default="bf16x3")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    print(run(a.batch, a.lr, a.dtype, a.iters))


if __name__ == "__main__":
    main()
