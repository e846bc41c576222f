"""Seeded synthetic clips and weights (SURVEY.md section 8 d): what the benchmark and the demo
configs feed the generator in place of the reference's video files (dataset.txt: 60 Vimeo URLs,
no network here) and its absent demo.h5.

Clip = smooth random field (8 random-phase 2-D sinusoids per channel + 5 % uniform noise, seeded per
clip with 1000 + clip_id), translated by (1.5, -0.75) px / frame so that FNet sees motion.  Weights =
Glorot-uniform kernels (Keras' default initialiser), biases U(-0.05, 0.05), FNet's flow head scaled by 8
and SRNet's last bias shifted by 0.45 so that a random-init network produces flows of a few tenths of a
pixel and outputs that spread over [0, 1].  Everything is a pure function of (seed, index) through a
counter-based splitmix64, so every rank, the tests and the CPU baseline see bit-identical data
(tests/test_host_logic.py checks this module against the oracle's own generator).
"""
import math

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(idx):
    with np.errstate(over="ignore"):
        z = (np.asarray(idx, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    """n float64 in [0, 1), element i a function of (seed, i) only."""
    with np.errstate(over="ignore"):
        base = _mix(np.array([seed], dtype=np.uint64))[0]
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
    return (_mix(ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synth_clip(clip_id, T, h, w, scale=1):
    """float32 [T, h, w, 3] in [0, 1]; `scale` = 4 gives the matching HR clip of an LR clip."""
    seed = 1000 + clip_id
    u = uniform01(seed, 8 * 3 * 4).reshape(3, 8, 4)
    fy = (u[:, :, 0] - 0.5) * 0.6 / scale
    fx = (u[:, :, 1] - 0.5) * 0.6 / scale
    ph = u[:, :, 2] * 2 * math.pi
    amp = 0.5 + u[:, :, 3]
    yy = np.arange(h, dtype=np.float64)[:, None]
    xx = np.arange(w, dtype=np.float64)[None, :]
    frames = np.zeros((T, h, w, 3), dtype=np.float64)
    for t in range(T):
        sy, sx = 1.5 * t * scale, -0.75 * t * scale
        for c in range(3):
            acc = np.zeros((h, w))
            for k in range(8):
                acc += amp[c, k] * np.sin(fy[c, k] * (yy - sy) + fx[c, k] * (xx - sx) + ph[c, k])
            frames[t, :, :, c] = acc
    frames = 0.5 + frames / (2.0 * 8 * 1.5) * 2.0
    noise = uniform01(seed + 77, T * h * w * 3).reshape(T, h, w, 3)
    frames = 0.95 * np.clip(frames, 0, 1) + 0.05 * noise
    return np.clip(frames, 0, 1).astype(np.float32)


def make_weights(seed=7, bias_range=0.05, flow_gain=8.0, out_bias=0.45):
    """Ordered {keras_name: float32 array} of the 76 tensors (layers.weight_specs order)."""
    import layers
    out = {}
    for i, (name, shape) in enumerate(layers.weight_specs()):
        u = uniform01(seed * 1000 + i, int(np.prod(shape)))
        if len(shape) == 4:
            w = (2.0 * u - 1.0) * math.sqrt(6.0 / (9 * shape[2] + 9 * shape[3]))
        else:
            w = (2.0 * u - 1.0) * bias_range
        if i == 26:
            w = w * flow_gain
        if i == 75:
            w = w + out_bias
        out[name] = w.reshape(shape).astype(np.float32)
    return out
