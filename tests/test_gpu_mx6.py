"""dtype "f16mx6" (SR_F16MX6): SRNet's residual blocks on conv_trunk6.hip -- f16 x f16 plus two MX-scaled FP6 cross terms per
product, activations stored as f16 + FP6 residue -- against the fp64/fp32 oracle at the fp32 bar of BASELINE.json's north_star
(max-abs 1e-3 on [0, 1] pixels, PSNR delta 0.01 dB), and against the bf16x3 kernels it replaces."""
import os
import sys

import numpy as np
import pytest

from oracle import ref_numpy as rn
from oracle import ref_torch as rt

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
from make_golden import step_inputs  # noqa: E402

FP32_TOL = 1e-3


@pytest.fixture(scope="module")
def sh():
    import sr_hip
    assert sr_hip.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return sr_hip


@pytest.fixture()
def split_always():
    os.environ["SR_SPLIT_TRUNK"] = "2"      # the split-planar trunk (whose residual blocks these are) at any size
    yield
    os.environ.pop("SR_SPLIT_TRUNK", None)


def make_ctx(sh, weights, h, w, B, dtype):
    ctx = sh.Context(h, w, max_batch=B, dtype=dtype)
    ctx.load_weights(weights)
    return ctx


def conv_ref(x, w, b):
    """fp64 3x3 'same' conv, NHWC x HWIO"""
    B, H, W, _ = x.shape
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    y = np.zeros((B, H, W, w.shape[3]))
    for ky in range(3):
        for kx in range(3):
            y += xp[:, ky:ky + H, kx:kx + W, :] @ w[ky, kx].astype(np.float64)
    return y + b


def test_format_round_trip(sh):
    """x -> mx6 -> x: the stored value hi + lo6 keeps ~15 bits (f16's 11 + what 4 FP6 bits under a per-32-channel scale add);
    the FP6 image alone keeps 4 bits of each block's maximum."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, 8, 40, 64)) * np.exp2(rng.integers(-6, 4, (2, 8, 40, 1)))).astype(np.float32)
    x[0, 0, 0] = 0.0
    x[0, 0, 1, :32] = 0.0
    back = sh.conv3x3_mx6(x, mode=2)
    blk = np.abs(x).reshape(2, 8, 40, 2, 32).max(-1, keepdims=True).repeat(32, -1).reshape(x.shape)
    assert np.abs(back - x).max() <= 2.0 ** -11 * 2.0 ** -3 * np.abs(x).max()
    assert (np.abs(back - x) <= blk * 2.0 ** -14 + 1e-30).all()
    x6 = sh.conv3x3_mx6(x, mode=3)
    assert (np.abs(x6 - x) <= blk * (0.25 / 3.75) + 1e-30).all()       # half a step of the top binade, scale up to 2x the tight one
    assert (x6[0, 0, 0] == 0).all() and (back[0, 0, 1, :32] == 0).all()


@pytest.mark.parametrize("B,h,w", [(1, 8, 64), (1, 8, 8), (2, 24, 40), (1, 16, 136), (3, 8, 72)])
@pytest.mark.parametrize("kind", ["relu", "res", "res_split"])
def test_one_layer(sh, B, h, w, kind):
    """One conv of a residual block against fp64: ReLU layer, residual layer (in place, mx6 out), residual layer writing
    split-planar records.  Error budget: the two FP6 cross terms (2^-11 * 2^-4 each, per-block scales) and the output's storage."""
    rng = np.random.default_rng(h * w + B)
    x = rng.standard_normal((B, h, w, 64)).astype(np.float32)
    x = np.maximum(x, 0) if kind != "relu" else x
    wt = (rng.standard_normal((3, 3, 64, 64)) * 0.05).astype(np.float32)
    b = (rng.standard_normal(64) * 0.1).astype(np.float32)
    res = rng.standard_normal((B, h, w, 64)).astype(np.float32)
    want = conv_ref(x, wt, b)
    if kind == "relu":
        got = sh.conv3x3_mx6(x, wt, b, relu=True)
        want = np.maximum(want, 0)
    else:
        got = sh.conv3x3_mx6(x, wt, b, res=res, mode=1 if kind == "res_split" else 0)
        want = want + res
    err = np.abs(got - want).max()
    print(f"\n[{kind} {B}x{h}x{w}] max-abs err {err:.2e} (output scale {np.abs(want).max():.2f})")
    assert err <= 2e-4 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("B,h,w", [(1, 8, 64), (2, 24, 40), (1, 8, 72), (3, 16, 8), (1, 96, 144), (5, 8, 8), (1, 40, 136), (4, 64, 64)])
def test_step_against_the_oracle_and_bf16x3(sh, weights7, split_always, B, h, w):
    """One generator step.  Shapes: one full tile, ragged right / bottom edges on the 8 x 64 tiles, several images (odd counts:
    uneven lanes and sub-batches), more tiles than one workgroup pair takes in a round."""
    lr_t, lr_tm1, est = step_inputs(610 + h + w, B, h, w)
    want, _ = rt.generator_step(weights7, lr_t, lr_tm1, est)
    outs = {}
    for dtype in ("bf16x3", "f16mx6"):
        ctx = make_ctx(sh, weights7, h, w, B, dtype)
        outs[dtype] = ctx.step(lr_t, lr_tm1, est)[0]
        ctx.close()
    e6, e3 = float(np.abs(outs["f16mx6"] - want).max()), float(np.abs(outs["bf16x3"] - want).max())
    print(f"\n[{B}x{h}x{w}] max-abs vs oracle: f16mx6 {e6:.2e}, bf16x3 {e3:.2e}; between them {np.abs(outs['f16mx6'] - outs['bf16x3']).max():.2e}")
    assert e6 <= FP32_TOL
    assert e6 <= 3e-4       # measured 1e-4 class; a layout mistake is 1e-2 and up


def test_config1_clip64_psnr(sh, weights7, split_always):
    """BASELINE.json configs[0] stand-in (1 clip x 10 frames, 64x64 LR) through the recurrence: every frame within 1e-3 of the
    oracle, PSNR against a common synthetic HR target within 0.01 dB of the oracle's."""
    clip = rn.synth_clip(0, 10, 64, 64)[:, None]
    hr = rn.synth_clip(0, 10, 256, 256, scale=4)[:, None]
    want = rt.run_clip(weights7, clip)
    ctx = make_ctx(sh, weights7, 64, 64, 1, "f16mx6")
    ctx.clip_begin(1)
    errs, dps = [], []
    for t in range(10):
        out = ctx.clip_push(clip[t])
        errs.append(float(np.abs(out - want[t]).max()))
        dps.append(abs(rn.psnr(out, hr[t]) - rn.psnr(want[t], hr[t])))
    ctx.close()
    print(f"\n[f16mx6] per-frame max-abs err: {['%.2e' % e for e in errs]}  max PSNR delta {max(dps):.4f} dB")
    assert max(errs) <= FP32_TOL
    assert max(dps) <= 0.01


def test_large_activations_do_not_overflow_f16(sh, weights7, split_always):
    """Trunk weights scaled so that activations pass f16's 65504: the hi parts saturate and the FP6 terms carry the rest (coarsely)
    instead of producing inf / NaN; the output stays finite and inside [0, 1]."""
    w = dict(weights7)
    k = [n for n in w if n.startswith("SRNet/conv2d_15/") and n.endswith("kernel:0")]
    assert k, "head conv's kernel not found by name"
    w[k[0]] = w[k[0]] * 3.0e5
    lr_t, lr_tm1, est = step_inputs(77, 1, 16, 64)
    ctx = make_ctx(sh, w, 16, 64, 1, "f16mx6")
    out = ctx.step(lr_t, lr_tm1, est)[0]
    ctx.close()
    assert np.isfinite(out).all() and out.min() >= 0.0 and out.max() <= 1.0


def test_training_updates_fall_back_to_bf16x3_until_the_next_commit(sh, weights7, split_always):
    """sr_train_apply re-packs the bf16x3 buffers on the device, not the host-packed f16mx6 ones: after an update the ctx runs the
    bf16x3 kernels (same numbers as a bf16x3 ctx that trained the same way); loading weights again brings conv_trunk6.hip back."""
    h, w, B = 16, 24, 2
    lr_t, lr_tm1, est = step_inputs(78, B, h, w)
    hr = np.random.default_rng(6).random((B, 4 * h, 4 * w, 3), dtype=np.float32)
    outs = {}
    for dtype in ("bf16x3", "f16mx6"):
        ctx = make_ctx(sh, weights7, h, w, B, dtype)
        before = ctx.step(lr_t, lr_tm1, est)[0]
        ctx.train_on_batch(lr_t, lr_tm1, est, hr, lr=2e-4)       # small: the trained net must stay well conditioned
        after = ctx.step(lr_t, lr_tm1, est)[0]
        trained = ctx.get_weights()
        ctx.load_weights(trained)
        again = ctx.step(lr_t, lr_tm1, est)[0]
        ctx.close()
        outs[dtype] = (before, after, again)
    b3, b6 = outs["bf16x3"], outs["f16mx6"]
    assert 0 < np.abs(b6[0] - b3[0]).max() <= 3e-4              # different kernels before training
    np.testing.assert_array_equal(b6[1], b3[1])                   # the same kernels after it
    assert np.abs(b3[1] - b3[0]).max() > 1e-4                     # and the update did change the weights
    assert 0 < np.abs(b6[2] - b3[2]).max() <= 3e-4              # conv_trunk6.hip again, on the trained weights


def test_error_does_not_grow_along_a_long_clip(sh, weights7, split_always):
    """30 recurrent frames (three times the benched clip length): the estimate feeds back through the warp every frame, so a
    format that drifted would show it here.  Per-frame max-abs against the oracle stays in the 1e-4 class."""
    T = 30
    clip = rn.synth_clip(5, T, 64, 64)[:, None]
    want = rt.run_clip(weights7, clip)
    ctx = make_ctx(sh, weights7, 64, 64, 1, "f16mx6")
    ctx.clip_begin(1)
    errs = [float(np.abs(ctx.clip_push(clip[t]) - want[t]).max()) for t in range(T)]
    ctx.close()
    print(f"\n[f16mx6, 30 frames] max-abs err: first 10 {max(errs[:10]):.2e}, middle {max(errs[10:20]):.2e}, last 10 {max(errs[20:]):.2e}")
    assert max(errs) <= 3e-4
    assert max(errs[20:]) <= 2.0 * max(errs[5:15]) + 1e-5
