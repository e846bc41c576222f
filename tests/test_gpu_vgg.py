"""GPU parity of TecoGAN's VGG19 perceptual terms (reference TecoGAN.py:125,129,867-901; SURVEY.md section 8 f rank 3, "VGG loss
weights' tensors supplied by file") through the C ABI (sr_vgg_*, sr_tecogan_set_vgg) against oracle/ref_vgg.py (torch float64).

The ImageNet weights are a download and are NOT used: SEEDED weights in the same 32-tensor layout exercise the same
arithmetic (PARITY UNPINNED like the rest: no Keras here).  Bars: losses 1e-5 (fp32) / 2e-4 (bf16x3) relative; the gradient
with respect to the estimate 2e-4 (fp32) of its max-abs on inputs the oracle finds well conditioned (16 ReLU layers and four
max poolings: a kink within fp32 reach of a 2x2 feature map is worth percents; measured 5e-6 .. 3e-5), bf16x3 3e-2 (its
forward sits on the other side of some kinks: measured 3e-3 .. 1.2e-2 of max-abs).  In the combined step the estimate itself
comes out of the fp32 generator a few 1e-6 away from the oracle's before it enters VGG19's 16 ReLU layers and four poolings on
64x64 .. 4x4 maps, and the conditioning search only guarantees that the ORACLE's gradients move by <= 2e-4 per +-1e-6 of
input: there the fp32 bar is median 3e-3 / worst 2e-2 over the 76 + 18 tensors (measured 8e-4 / 1.7e-3), the losses -- all
twelve, the four perceptual ones included -- agree to 1e-5, and the stand-alone test above pins the VGG19 gradient itself."""
import numpy as np
import pytest

from oracle import ref_disc as rd
from oracle import ref_vgg as rv
from test_gpu_disc import check_named, conditioned, gan_data, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sh():
    import sr_hip
    assert sr_hip.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return sr_hip


@pytest.fixture(scope="module")
def vggw():
    return rv.make_vgg_weights(3)


def image_pair(seed, B, H, W):
    rng = np.random.default_rng(seed)
    y_true = rng.random((B, H, W, 3), dtype=np.float32)
    y_pred = np.clip(y_true + 0.1 * rng.standard_normal((B, H, W, 3)).astype(np.float32), 0, 1)
    return [y_true, y_pred]


@pytest.mark.parametrize("dtype,ltol,gtol", [("fp32", 1e-5, 2e-4), ("bf16x3", 2e-4, 3e-2)])
@pytest.mark.parametrize("B,H,W", [(2, 32, 48), (1, 64, 64), (3, 16, 16)])
def test_vgg_losses_and_input_gradient(sh, vggw, B, H, W, dtype, ltol, gtol):
    weights = [0.3, 0.2, 0.4, 1.0]          # all four terms visible in the gradient (the reference's 3e-5 .. 2e-3 hide three of them)
    grads = lambda a: ({"g": rv.vgg_loss_and_grad(vggw, a[0], a[1], weights)[1]},)
    _, (y_true, y_pred) = conditioned(lambda s: image_pair(s, B, H, W), grads, 40 + H, amp=1e-6)
    want_l, want_g = rv.vgg_loss_and_grad(vggw, y_true, y_pred, weights)
    v = sh.Vgg19(H, W, max_batch=B, dtype=dtype)
    v.load_weights(vggw)
    losses, g = v.loss_grads(y_true, y_pred, weights)
    only_l, none = v.loss_grads(y_true, y_pred, weights, want_grad=False)
    again_l, again_g = v.loss_grads(y_true, y_pred, weights)
    v.close()
    print(f"\n[vgg {dtype} {B}x{H}x{W}] losses {losses} vs {want_l}; gradient rel err {rel_err(g, want_g):.2e}")
    assert np.allclose(losses, want_l, rtol=ltol)
    assert none is None and only_l == losses and again_l == losses and np.array_equal(g, again_g)
    assert rel_err(g, want_g) <= gtol


def test_vgg_rejections(sh, vggw):
    with pytest.raises(ValueError):
        sh.Vgg19(40, 64)                                   # four poolings: multiples of 16
    v = sh.Vgg19(32, 32, max_batch=1)
    with pytest.raises(ValueError):
        v.loss_grads(np.zeros((1, 32, 32, 3)), np.zeros((1, 32, 32, 3)))     # weights not committed
    with pytest.raises(ValueError):
        v.load_weights(list(vggw.values())[:-1])           # 31 tensors
    bad = list(vggw.values())
    bad[2] = bad[2][..., :32]
    with pytest.raises(ValueError):
        v.load_weights(bad)                                # block1_conv2 kernel of the wrong shape
    v.load_weights(vggw)
    with pytest.raises(ValueError):
        v.loss_grads(np.zeros((2, 32, 32, 3)), np.zeros((2, 32, 32, 3)))     # batch above max_batch
    v.close()


@pytest.mark.parametrize("dtype,tol", [("fp32", (3e-3, 2e-2)), ("bf16x3", (1e-2, 0.2))])
def test_tecogan_step_with_the_vgg_terms(sh, weights7, vggw, dtype, tol):
    """tecogan.train_on_batch with the VGG19 weights attached: all twelve losses, and the generator / discriminator gradients
    with the perceptual terms' gradient flowing through tanh and the clip into SRNet and FNet.  The four VGG weights are raised
    (x 1e3 .. 1e5) so that an error in their gradient cannot hide behind the other eight terms."""
    B, h, w = 2, 16, 16
    H, W = 4 * h, 4 * w
    lw = list(rd.LOSS_WEIGHTS)
    lw[6:10] = [0.3, 0.2, 0.4, 1.0]
    dw = rd.make_disc_weights(6, H, W)
    extra = {}

    def make(seed):
        ins, extra["feats"], extra["pp"] = gan_data(seed, B, h, w)
        return ins

    oracle = lambda a: rd.tecogan_loss_and_grads(weights7, dw, *a, extra["feats"], extra["pp"], loss_weights=lw, vgg_weights=vggw)
    _, ins = conditioned(make, lambda a: oracle(a)[1:], 903, amp=1e-6)
    feats, pp = extra["feats"], extra["pp"]
    want_l, want_gg, want_dg = oracle(ins)
    assert min(want_l[7:11]) > 0
    ctx = sh.Context(h, w, max_batch=B, dtype=dtype, model="tecogan")
    ctx.load_weights(weights7)
    d = sh.Discriminator(H, W, max_batch=B, dtype=dtype)
    d.load_weights(dw)
    v = sh.Vgg19(H, W, max_batch=B, dtype=dtype)
    v.load_weights(vggw)
    ctx.set_vgg(v)
    losses = sh.tecogan_grads(ctx, d, *ins, feats, pp, lw)
    import ctypes as C
    flat = np.empty(ctx.param_count(), np.float32)
    assert sh.lib().sr_train_read_grads(ctx._h, flat.ctypes.data_as(C.c_void_p)) == 0
    got_g = ctx.split_grads(flat)
    got_d = d.split_grads(d.read_grads())
    print(f"\n[tecogan + vgg, {dtype}] losses {np.round(losses, 6).tolist()} vs {np.round(want_l, 6).tolist()}")
    assert np.allclose(losses, want_l, rtol=1e-5 if dtype == "fp32" else 2e-4, atol=1e-7)
    check_named(got_g, want_gg, tol)
    check_named(got_d, want_dg, tol)
    # detached: the four terms are 0 again and the total drops by their weighted sum
    ctx.set_vgg(None)
    l0 = sh.tecogan_grads(ctx, d, *ins, feats, pp, lw)
    assert l0[7:11] == [0.0] * 4 and abs((losses[0] - l0[0]) - sum(wk * q for wk, q in zip(lw[6:10], losses[7:11]))) <= 1e-4 * abs(losses[0])
    for x in (ctx, d, v):
        x.close()
    with pytest.raises((ValueError, RuntimeError)):
        sh.Context(h, w, dtype=dtype).set_vgg(sh.Vgg19(H, W))       # a FRVSR ctx has no combined step; weights not committed either
