"""Pad mode (sr_create_padded: LR sizes that are not multiples of 8, e.g. the 180 rows of 180p) and the uint8 output
path (sr_clip_push_u8 = predict + utils.denormalize, reference FRVSR.py:293-351, utils.py:61-63), through the C ABI.

Oracle for pad mode = the reference graph at the padded size on zero-padded inputs, cropped (SURVEY.md section 7
(ii)); the reference itself cannot build a 180-row model (layers.py:31-41 gives 184 rows back)."""
import numpy as np
import pytest

from oracle import ref_numpy as rn
from oracle import ref_torch as rt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sh():
    import sr_hip
    assert sr_hip.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return sr_hip


def _pad(a, rows, cols):
    out = np.zeros((a.shape[0], rows, cols, a.shape[3]), np.float32)
    out[:, :a.shape[1], :a.shape[2]] = a
    return out


def test_unpadded_create_still_refuses_odd_sizes(sh):
    with pytest.raises(ValueError, match="multiples of 8"):
        sh.Context(20, 28)
    with pytest.raises(ValueError):
        sh.Context(1, 28, pad=True)


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
@pytest.mark.parametrize("h,w,B", [(20, 28, 2), (17, 40, 1), (24, 33, 3)])
def test_pad_mode_stateless_ops(sh, weights7, h, w, B, dtype):
    hp, wp = -(-h // 8) * 8, -(-w // 8) * 8
    rng = np.random.default_rng(h * 100 + w)
    lr_t, lr_tm1 = rng.random((B, h, w, 3), dtype=np.float32), rng.random((B, h, w, 3), dtype=np.float32)
    est = rng.random((B, 4 * h, 4 * w, 3), dtype=np.float32)
    taps = {}
    want, want_lw = rt.generator_step(weights7, _pad(lr_t, hp, wp), _pad(lr_tm1, hp, wp), _pad(est, 4 * hp, 4 * wp), taps=taps)
    ctx = sh.Context(h, w, max_batch=B, dtype=dtype, pad=True)
    ctx.load_weights(weights7)
    got, lw = ctx.step(lr_t, lr_tm1, est)
    assert got.shape == (B, 4 * h, 4 * w, 3) and lw.shape == (B, h, w, 3)
    assert np.abs(got - want[:, :4 * h, :4 * w]).max() <= 1e-3
    assert np.abs(lw - want_lw[:, :h, :w]).max() <= 1e-3
    assert np.abs(ctx.fnet(lr_t, lr_tm1) - taps["f_lr"][:, :h, :w]).max() <= 1e-3
    assert np.abs(ctx.hr_warp(lr_t, lr_tm1, est) - taps["hr_warp"][:, :4 * h, :4 * w]).max() <= 1e-3
    ctx.close()


def test_pad_mode_180p_clip_recurrence(sh, weights7):
    """1 clip x 3 frames of 180x320 through a padded ctx == the oracle's recurrence at 184x320, cropped to 720 rows
    (the state keeps its 16 padded rows between frames on both sides)."""
    clip = rn.synth_clip(2, 3, 180, 320)[:, None]
    padded = np.zeros((3, 1, 184, 320, 3), np.float32)
    padded[:, :, :180] = clip
    want = rt.run_clip(weights7, padded)[:, :, :720]
    ctx = sh.Context(180, 320, max_batch=1, dtype="bf16x3", pad=True)
    ctx.load_weights(weights7)
    ctx.clip_begin(1)
    for t in range(3):
        out = ctx.clip_push(clip[t])
        assert out.shape == (1, 720, 1280, 3)
        assert np.abs(out - want[t]).max() <= 1e-3
    with pytest.raises(RuntimeError, match="multiple of 8"):
        ctx.train_on_batch(clip[0], clip[0], want[0], want[0])
    ctx.close()


@pytest.mark.parametrize("h,w,pad", [(24, 40, False), (20, 28, True), (20, 40, True)])
def test_u8_output_equals_denormalized_float_output(sh, weights7, h, w, pad):
    """Three frames of two clips: the uint8 frames (asynchronous copy stream, two buffers in rotation) equal
    utils.denormalize of the float frames bit for bit; the recurrent state is the same either way."""
    import utils
    B, T = 2, 3
    clip = np.stack([rn.synth_clip(10 + i, T, h, w) for i in range(B)], axis=1)
    ctx = sh.Context(h, w, max_batch=B, pad=pad)
    ctx.load_weights(weights7)
    ctx.clip_begin(B)
    want = [utils.denormalize(ctx.clip_push(clip[t])) for t in range(T)]
    bufs = [np.full((B, 4 * h, 4 * w, 3), 7, np.uint8) for _ in range(2)]
    ctx.clip_begin(B)
    frames = [np.ascontiguousarray(clip[t]) for t in range(T)]   # stay alive while their copies may be in flight
    for t in range(T):
        ctx.clip_push_u8(frames[t].ctypes.data, bufs[t % 2].ctypes.data)
        ctx.u8_wait(1)
        if t >= 1:
            np.testing.assert_array_equal(bufs[(t - 1) % 2], want[t - 1])
    ctx.u8_wait(0)
    np.testing.assert_array_equal(bufs[(T - 1) % 2], want[T - 1])
    ctx.close()


@pytest.mark.parametrize("h,w,pad", [(24, 40, False), (20, 28, True)])
def test_float_read_back_beside_the_following_steps(sh, weights7, h, w, pad):
    """sr_clip_read_est_begin / _end: the float estimates of frame t, copied out while frames t + 1 and t + 2 are pushed, are
    frame t's (the push after next waits on the device for the copy: it overwrites the buffer being read); the call-order
    errors are SR_ERR_STATE."""
    B, T = 3, 5
    clip = np.stack([rn.synth_clip(20 + i, T, h, w) for i in range(B)], axis=1)
    ctx = sh.Context(h, w, max_batch=B, pad=pad)
    ctx.load_weights(weights7)
    ctx.clip_begin(B)
    want = [ctx.clip_push(clip[t]) for t in range(T)]
    pin = sh.PinnedArray((2, 4 * h, 4 * w, 3), np.float32)
    bufs = [np.zeros((B, 4 * h, 4 * w, 3), np.uint8) for _ in range(2)]
    frames = [np.ascontiguousarray(clip[t]) for t in range(T)]
    try:
        with pytest.raises(RuntimeError, match="without sr_clip_read_est_begin"):
            ctx.clip_read_est_end()
        ctx.clip_begin(B)
        for t in range(T):
            ctx.clip_push_u8(frames[t].ctypes.data, bufs[t % 2].ctypes.data)
            if t == 1:
                pin.a[...] = -1.0
                ctx.clip_read_est_begin(pin.a, first=1, n=2)       # clips 1, 2 of frame 1
                with pytest.raises(RuntimeError, match="has not been ended"):
                    ctx.clip_read_est_begin(pin.a, first=0, n=2)
        ctx.clip_read_est_end()                                        # three more frames were pushed meanwhile
        np.testing.assert_array_equal(pin.a, want[1][1:3])
        ctx.u8_wait(0)
        np.testing.assert_array_equal(ctx.clip_read_est(0, B), want[T - 1])
    finally:
        ctx.sync()
        pin.close()
        ctx.close()


@pytest.mark.parametrize("finish", ["sync", "clip_begin", "read_end_first"])
def test_pending_copies_are_submitted_by_whatever_follows(sh, weights7, finish):
    """The copy-out of the newest uint8 frame (and a float read-back) is only SUBMITTED by the next push, so that the next push's
    copy-in stands in front of it in the device's copy queue (sr_hip.h).  When no push follows, sr_sync / sr_clip_begin /
    sr_clip_read_est_end submit it: the last frame of a clip arrives without a sr_u8_wait(ctx, 0), from page-locked and from
    pageable frames alike."""
    import utils
    h, w, B, T = 24, 40, 2, 2
    clip = np.stack([rn.synth_clip(40 + i, T, h, w) for i in range(B)], axis=1)
    ctx = sh.Context(h, w, max_batch=B)
    ctx.load_weights(weights7)
    ctx.clip_begin(B)
    want = [ctx.clip_push(clip[t]) for t in range(T)]
    lr_pin = sh.PinnedArray((T, B, h, w, 3), np.float32)
    out_pin = sh.PinnedArray((T, B, 4 * h, 4 * w, 3), np.uint8)
    est_pin = sh.PinnedArray((B, 4 * h, 4 * w, 3), np.float32)
    try:
        lr_pin.a[...] = clip
        out_pin.a[...] = 7
        est_pin.a[...] = -1.0
        ctx.clip_begin(B)
        for t in range(T):
            ctx.clip_push_u8(lr_pin.ptr + t * lr_pin.a[0].nbytes, out_pin.ptr + t * out_pin.a[0].nbytes)
        ctx.clip_read_est_begin(est_pin.a, first=0, n=B)      # frame T - 1, nothing follows it
        if finish == "sync":
            ctx.sync()
            np.testing.assert_array_equal(out_pin.a[T - 1], utils.denormalize(want[T - 1]))
            ctx.clip_read_est_end()
        elif finish == "clip_begin":
            ctx.clip_read_est_end()
            ctx.clip_begin(B)
            ctx.sync()
            np.testing.assert_array_equal(out_pin.a[T - 1], utils.denormalize(want[T - 1]))
        else:
            ctx.clip_read_est_end()
            np.testing.assert_array_equal(est_pin.a, want[T - 1])
            ctx.u8_wait(0)
            np.testing.assert_array_equal(out_pin.a[T - 1], utils.denormalize(want[T - 1]))
        np.testing.assert_array_equal(est_pin.a, want[T - 1])
        np.testing.assert_array_equal(out_pin.a[0], utils.denormalize(want[0]))
    finally:
        ctx.sync()
        for p in (lr_pin, out_pin, est_pin):
            p.close()
        ctx.close()
