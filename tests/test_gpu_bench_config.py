"""Parity of the exact path bench.py times (BASELINE.json configs[1]) and of the batch-splitting code it
relies on: 184x320 (180p zero-padded), 16 distinct clips in lock-step, 10 recurrent frames, default lanes
and automatic SRNet-trunk sub-batches, every dtype bench.py reports.

The reference keeps the state of all tiles in one batch and calls predict once per frame
(reference FRVSR.py:231-234,293,309); the CPU oracle runs the same recurrence on the clips it is asked for.
Bars: fp32 and bf16x3 max-abs <= 1e-3 on [0,1] pixels and PSNR delta <= 0.01 dB on every frame; plain
bf16 is reported with its own (looser) bound."""
import os
import sys

import numpy as np
import pytest

from oracle import ref_numpy as rn
from oracle import ref_torch as rt

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_golden import step_inputs  # noqa: E402

H, W, HP, CLIPS, FRAMES = 180, 320, 184, 16, 10
CHECKED = (0, 3, 9, 15)      # first / last clip of either lane and of different trunk sub-batches


@pytest.fixture(scope="module")
def sh():
    import sr_hip
    assert sr_hip.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return sr_hip


@pytest.fixture(scope="module")
def bench_clips():
    """[FRAMES, CLIPS, 184, 320, 3]: bench.py's inputs (synth.synth_clip == oracle synth_clip, rows 180..183 zero)."""
    lr = np.zeros((FRAMES, CLIPS, HP, W, 3), np.float32)
    for c in range(CLIPS):
        lr[:, c, :H] = rn.synth_clip(c, FRAMES, H, W)
    return lr


@pytest.fixture(scope="module")
def oracle_frames(weights7, bench_clips):
    """Oracle estimates of the checked clips, [FRAMES, 4, 736, 1280, 3] (about 3 s of CPU per clip)."""
    return rt.run_clip(weights7, bench_clips[:, list(CHECKED)])


@pytest.mark.parametrize("dtype,tol,psnr_tol,env", [("fp32", 1e-3, 0.01, {}), ("bf16x3", 1e-3, 0.01, {}), ("bf16", 0.15, 0.5, {}),
                                                    ("bf16x3", 1e-3, 0.01, {"SR_TAIL": "0"}), ("f16mx6", 1e-3, 0.01, {})])
def test_bench_configuration_every_frame(sh, weights7, bench_clips, oracle_frames, dtype, tol, psnr_tol, env):
    """env SR_TAIL=0: the x4 tail on the tap-major kernels (conv_mfma.hip) instead of conv_tail.hip's plane-major one."""
    hr = np.stack([rn.synth_clip(c, FRAMES, 4 * H, 4 * W, scale=4) for c in CHECKED], axis=1)   # common PSNR target
    os.environ.update(env)
    try:
        ctx = sh.Context(HP, W, max_batch=CLIPS, dtype=dtype)
        ctx.load_weights(weights7)
    finally:
        for k in env:
            os.environ.pop(k, None)
    ctx.clip_begin(CLIPS)
    errs, dps = [], []
    for t in range(FRAMES):
        out = ctx.clip_push(bench_clips[t])[list(CHECKED)]
        errs.append(float(np.abs(out - oracle_frames[t]).max()))
        for i in range(len(CHECKED)):
            a, b = out[i, :4 * H], oracle_frames[t, i, :4 * H]
            dps.append(abs(rn.psnr(a, hr[t, i]) - rn.psnr(b, hr[t, i])))
    ctx.close()
    print(f"\n[{dtype} {env}] 184x320 B=16 per-frame max-abs err: {['%.2e' % e for e in errs]}  max PSNR delta {max(dps):.4f} dB")
    assert max(errs) <= tol
    assert max(dps) <= psnr_tol


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
@pytest.mark.parametrize("planar", [True, False])
@pytest.mark.parametrize("lanes,sub", [(2, 1), (2, 2), (1, 1), (1, 2), (3, 1)])
def test_forced_lane_and_trunk_splits(sh, weights7, dtype, planar, lanes, sub):
    """B = 5 images of 96x144 (above the 65,536-pixel single-lane cut) with the trunk forced into sub-batches of
    1 or 2 images (SR_TRUNK_SUB) on 1, 2 or 3 lanes (SR_LANES): every image starts at a non-zero offset of the
    ping-pong tensors in some launch.  One stateless step against the oracle."""
    B, h, w = 5, 96, 144
    lr_t, lr_tm1, est = step_inputs(900 + lanes * 10 + sub, B, h, w)
    want, want_lw = rt.generator_step(weights7, lr_t, lr_tm1, est)
    env = {"SR_TRUNK_SUB": str(sub), "SR_LANES": str(lanes)}
    if not planar:
        env["SR_NO_PLANAR"] = "1"
    os.environ.update(env)
    try:
        ctx = sh.Context(h, w, max_batch=B, dtype=dtype)
        ctx.load_weights(weights7)
        got, lw = ctx.step(lr_t, lr_tm1, est)
        ctx.close()
    finally:
        for k in env:
            os.environ.pop(k, None)
    assert np.abs(got - want).max() <= 1e-3
    assert np.abs(lw - want_lw).max() <= 1e-3


# ---- split-planar trunk (conv_trunk.hip): forced on for shapes that are too small to select it by themselves ------
def _with_env(env, fn):
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k in env:
            os.environ.pop(k, None)


@pytest.mark.parametrize("model", ["frvsr", "tecogan"])
@pytest.mark.parametrize("h,w,B", [(8, 8, 1), (16, 40, 3), (24, 72, 2), (72, 24, 2), (8, 136, 1), (40, 64, 5)])
def test_split_planar_trunk_on_ragged_shapes(sh, weights7, h, w, B, model):
    """Tiles of 8 x 64 pixels: widths below / above / not a multiple of 64, a single tile row, several images
    (a pair of workgroups walks tiles of different images), more XCD bands than tiles."""
    lr_t, lr_tm1, est = step_inputs(700 + h + w, B, h, w)
    want, want_lw = rt.generator_step(weights7, lr_t, lr_tm1, est, model=model)

    def run():
        ctx = sh.Context(h, w, max_batch=B, dtype="bf16x3", model=model)
        ctx.load_weights(weights7)
        out = ctx.step(lr_t, lr_tm1, est)
        ctx.close()
        return out

    got, lw = _with_env({"SR_SPLIT_TRUNK": "2"}, run)
    assert np.abs(got - want).max() <= 1e-3
    assert np.abs(lw - want_lw).max() <= 1e-3
    old, _ = _with_env({"SR_SPLIT_TRUNK": "0"}, run)
    assert np.abs(got - old).max() <= 2e-4          # the two bf16x3 paths: same products, 16-bit trunk storage vs fp32


def test_split_planar_trunk_follows_training(sh, weights7):
    """An optimizer step re-packs every weight buffer on the device, the split-planar trunk's second packing
    included: after two train_on_batch steps the forced-split step equals the classic path on the same ctx state."""
    h, w, B = 16, 24, 2
    lr_t, lr_tm1, est = step_inputs(77, B, h, w)
    hr = np.random.default_rng(5).random((B, 4 * h, 4 * w, 3), dtype=np.float32)

    def run():
        ctx = sh.Context(h, w, max_batch=B, dtype="bf16x3")
        ctx.load_weights(weights7)
        before = ctx.step(lr_t, lr_tm1, est)[0]
        for _ in range(2):
            ctx.train_on_batch(lr_t, lr_tm1, est, hr, lr=1e-2)
        after = ctx.step(lr_t, lr_tm1, est)[0]
        ctx.close()
        return before, after

    b2, a2 = _with_env({"SR_SPLIT_TRUNK": "2"}, run)
    b0, a0 = _with_env({"SR_SPLIT_TRUNK": "0"}, run)
    assert np.abs(a0 - b0).max() > 1e-3             # the steps moved the weights visibly
    assert np.abs(b2 - b0).max() <= 2e-4
    assert np.abs(a2 - a0).max() <= 2e-4


def test_context_too_large_for_the_split_kernels_takes_the_generic_path(sh, weights7):
    """the split-planar kernels address the zero / dump tail behind image max_batch - 1 with 32-bit
    offsets, so a ctx with max_batch x h x w x 1 KiB >= 4 GiB (72 clips at 184x320) failed every bf16x3 step with 'tensor
    too large for 32-bit offsets', whatever B.  Such a ctx now runs the generic 64-bit kernels; a small B on it equals the
    same step on a small ctx (split path) to the two paths' usual distance."""
    B = 2
    lr_t, lr_tm1, est = step_inputs(4242, B, HP, W)

    def run(max_batch):
        ctx = sh.Context(HP, W, max_batch=max_batch, dtype="bf16x3")
        ctx.load_weights(weights7)
        out = ctx.step(lr_t, lr_tm1, est)
        ctx.close()
        return out

    big, big_lw = run(72)          # 72 x 58,880 px x 1 KiB + 4 KiB > 4 GiB
    small, small_lw = run(B)
    assert np.abs(big - small).max() <= 2e-4
    assert np.abs(big_lw - small_lw).max() <= 2e-4
    want, _ = rt.generator_step(weights7, lr_t[:1], lr_tm1[:1], est[:1])
    assert np.abs(big[:1] - want).max() <= 1e-3
