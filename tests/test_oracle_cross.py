"""The two CPU restatements against each other and against the golden vectors."""
import os
import sys

import numpy as np
import pytest

from oracle import ref_numpy as rn
from oracle import ref_torch as rt

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
from make_golden import step_inputs  # noqa: E402


@pytest.mark.parametrize("h,w,B", [(16, 16, 2), (24, 40, 1), (64, 64, 1)])
def test_numpy_vs_torch_step(weights7, h, w, B):
    lr_t, lr_tm1, est = step_inputs(5 + h, B, h, w)
    ta, tb = {}, {}
    a, la = rn.generator_step(weights7, lr_t, lr_tm1, est, taps=ta)
    b, lb = rt.generator_step(weights7, lr_t, lr_tm1, est, taps=tb)
    assert np.abs(a - b).max() <= 5e-5
    assert np.abs(la - lb).max() <= 1e-5
    for k in ("f_lr", "f_hr", "hr_warp", "s_s", "lr_warp"):
        # white-noise est + query coordinates up to 255 (fp32 ulp 1.5e-5): the warp
        # amplifies a 1-ulp flow difference to ~1e-5, hence the looser bound there
        tol = 5e-5 if k in ("hr_warp", "s_s") else 1e-5
        assert np.abs(ta[k] - tb[k]).max() <= tol, k


def test_numpy_vs_torch_tecogan(weights7):
    lr_t, lr_tm1, est = step_inputs(9, 1, 16, 24)
    a, _ = rn.generator_step(weights7, lr_t, lr_tm1, est, model="tecogan")
    b, _ = rt.generator_step(weights7, lr_t, lr_tm1, est, model="tecogan")
    assert np.abs(a - b).max() <= 1e-5
    f, _ = rn.generator_step(weights7, lr_t, lr_tm1, est)
    want = np.tanh(f + rn.resize_bilinear_legacy(lr_t, 4))      # TecoGAN.py:74-76
    assert np.abs(a - want).max() <= 1e-6


def test_golden_step16(weights7):
    g = np.load(os.path.join(GOLD, "step16.npz"))
    lr_t, lr_tm1, est = step_inputs(int(g["seed"]), 2, 16, 16)
    for mod in (rn, rt):
        taps = {}
        out, lw = mod.generator_step(weights7, lr_t, lr_tm1, est, taps=taps)
        assert np.abs(out - g["est_t"]).max() <= 2e-5
        assert np.abs(lw - g["lr_warp"]).max() <= 2e-5
        for k in ("f_lr", "f_hr", "hr_warp", "s_s"):
            assert np.abs(taps[k] - g["tap_" + k]).max() <= 2e-5, k
        out, _ = mod.generator_step(weights7, lr_t, lr_tm1, est, model="tecogan")
        assert np.abs(out - g["est_t_tecogan"]).max() <= 2e-5


def test_golden_clip24x40_recurrence(weights7):
    g = np.load(os.path.join(GOLD, "clip24x40.npz"))
    clip = rn.synth_clip(int(g["clip_id"]), 3, 24, 40)[:, None]
    for mod in (rn, rt):
        outs = mod.run_clip(weights7, clip)
        assert np.abs(outs[-1] - g["last"]).max() <= 5e-5
        np.testing.assert_allclose(outs.mean(axis=(1, 2, 3, 4)), g["means"], atol=1e-5)


def test_golden_clip64_psnr(weights7):
    """config-1 stand-in (BASELINE.json configs[0]); demo.h5 is absent."""
    g = np.load(os.path.join(GOLD, "clip64.npz"))
    clip = rn.synth_clip(0, 10, 64, 64)[:, None]
    hr = rn.synth_clip(0, 10, 256, 256, scale=4)[:, None]
    outs = rt.run_clip(weights7, clip)
    psnr = np.array([rn.psnr(outs[t], hr[t]) for t in range(10)])
    assert np.abs(psnr - g["psnr"]).max() <= 0.01                 # north_star PSNR gate
    assert np.abs(outs[-1][0, 112:144, 112:144] - g["crop_last"]).max() <= 1e-4
    assert np.abs(outs[0][0, 112:144, 112:144] - g["crop_first"]).max() <= 1e-4


def test_recurrent_driver_semantics(weights7):
    """state0 = zeros; est state := output; prev LR := current (FRVSR.py:231-234,293,309)."""
    clip = rn.synth_clip(1, 2, 16, 16)[:, None]
    outs = rn.run_clip(weights7, clip)
    z_lr = np.zeros((1, 16, 16, 3), np.float32)
    z_hr = np.zeros((1, 64, 64, 3), np.float32)
    o0, _ = rn.generator_step(weights7, clip[0], z_lr, z_hr)
    o1, _ = rn.generator_step(weights7, clip[1], clip[0], o0)
    np.testing.assert_array_equal(outs[0], o0)
    np.testing.assert_array_equal(outs[1], o1)


def test_batch_entries_are_independent(weights7):
    """tiles / clips share no state (FRVSR.py:280-301): batching == per-entry runs."""
    lr_t, lr_tm1, est = step_inputs(33, 3, 16, 16)
    full, _ = rt.generator_step(weights7, lr_t, lr_tm1, est)
    for b in range(3):
        one, _ = rt.generator_step(weights7, lr_t[b:b + 1], lr_tm1[b:b + 1], est[b:b + 1])
        assert np.abs(one[0] - full[b]).max() <= 5e-5
