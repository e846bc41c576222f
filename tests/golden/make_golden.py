"""Generate the committed golden vectors under tests/golden/.

Run from the repo root:  python tests/golden/make_golden.py

The reference cannot run here (TensorFlow 1.13 / Keras 2.2.4 absent, SURVEY.md
section 8c), so these vectors come from the fp64 numpy restatement
(oracle/ref_numpy.py) and are rounded to fp32 for storage.  They pin the
oracle against accidental drift and give the GPU tests a fixture that does not
need the oracle's speed.  Inputs are regenerated from seeds by the tests; a few
small inputs are stored too so a fixture is self-describing.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_numpy as rn  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
F64 = np.float64


def f32(x):
    return np.asarray(x, dtype=np.float32)


def step_inputs(seed, B, h, w):
    """Seeded single-step inputs with a non-trivial previous estimate."""
    lr_t = rn.uniform01(seed, B * h * w * 3).reshape(B, h, w, 3).astype(np.float32)
    lr_tm1 = rn.uniform01(seed + 1, B * h * w * 3).reshape(B, h, w, 3).astype(np.float32)
    est = rn.uniform01(seed + 2, B * 16 * h * w * 3).reshape(B, 4 * h, 4 * w, 3).astype(np.float32)
    return lr_t, lr_tm1, est


def main():
    w = rn.make_weights(7)

    # 1. one step, 16x16, batch 2, every intermediate
    lr_t, lr_tm1, est = step_inputs(101, 2, 16, 16)
    taps = {}
    out, lw = rn.generator_step(w, lr_t, lr_tm1, est, dtype=F64, taps=taps)
    out_teco, _ = rn.generator_step(w, lr_t, lr_tm1, est, model="tecogan", dtype=F64)
    np.savez_compressed(os.path.join(OUT, "step16.npz"),
                        seed=101, est_t=f32(out), lr_warp=f32(lw), est_t_tecogan=f32(out_teco),
                        **{"tap_" + k: f32(v) for k, v in taps.items()})

    # 2. 3-frame recurrence at 24x40 (odd pyramid 24->12->6->3, 40->20->10->5)
    clip = rn.synth_clip(3, 3, 24, 40)[:, None]
    outs = rn.run_clip(w, clip, dtype=F64)
    np.savez_compressed(os.path.join(OUT, "clip24x40.npz"), clip_id=3,
                        last=f32(outs[-1]), means=outs.mean(axis=(1, 2, 3, 4)),
                        sums=outs.sum(axis=(1, 2, 3, 4)))

    # 3. config-1 stand-in: 1 clip x 10 frames at 64x64 (demo.h5 absent -> seeded weights)
    clip = rn.synth_clip(0, 10, 64, 64)[:, None]
    hr = rn.synth_clip(0, 10, 256, 256, scale=4)[:, None]
    outs = rn.run_clip(w, clip, dtype=F64)
    psnr = np.array([rn.psnr(outs[t], hr[t].astype(F64)) for t in range(10)])
    np.savez_compressed(os.path.join(OUT, "clip64.npz"), clip_id=0, psnr=psnr,
                        means=outs.mean(axis=(1, 2, 3, 4)),
                        crop_last=f32(outs[-1][0, 112:144, 112:144]),
                        crop_first=f32(outs[0][0, 112:144, 112:144]))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
