"""Known-answer tests of SURVEY.md Appendix A on BOTH CPU restatements.

The reference has no tests; these hand-derivable KATs are
what pins the oracle's op semantics (parity otherwise unpinned)."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as rn
from oracle import ref_torch as rt


def _nhwc(a):
    return np.asarray(a, dtype=np.float32)


def _both_conv(x, w, b):
    a = rn.conv3x3_same(x, w, b)
    t = rt.to_nhwc(rt.conv3x3_same(rt.to_nchw(x), torch.tensor(w).permute(3, 2, 0, 1).contiguous(),
                                   torch.tensor(b)))
    return a, t


def test_conv_kat_ones():
    x = _nhwc([[1, 2, 3], [4, 5, 6], [7, 8, 9]]).reshape(1, 3, 3, 1)
    w = np.ones((3, 3, 1, 1), np.float32)
    b = np.zeros(1, np.float32)
    want = _nhwc([[12, 21, 16], [27, 45, 33], [24, 39, 28]]).reshape(1, 3, 3, 1)
    for got in _both_conv(x, w, b):
        np.testing.assert_array_equal(got, want)


def test_conv_kat_asymmetric_shift():
    x = _nhwc([[1, 2, 3], [4, 5, 6], [7, 8, 9]]).reshape(1, 3, 3, 1)
    w = np.zeros((3, 3, 1, 1), np.float32)
    w[0, 0] = 1
    want = _nhwc([[0, 0, 0], [0, 1, 2], [0, 4, 5]]).reshape(1, 3, 3, 1)   # out[y,x] = in[y-1,x-1]
    for got in _both_conv(x, w, np.zeros(1, np.float32)):
        np.testing.assert_array_equal(got, want)


def test_conv_channel_order_and_bias():
    # W[ky,kx,i,o]: pick centre tap only, mix channels asymmetrically
    x = np.zeros((1, 2, 2, 2), np.float32)
    x[0, 0, 0] = [1, 10]
    w = np.zeros((3, 3, 2, 3), np.float32)
    w[1, 1, 0] = [1, 2, 3]
    w[1, 1, 1] = [0.5, 0.25, 0.125]
    b = _nhwc([100, 200, 300])
    for got in _both_conv(x, w, b):
        np.testing.assert_allclose(got[0, 0, 0], [106, 204.5, 304.25])
        np.testing.assert_allclose(got[0, 1, 1], [100, 200, 300])


@pytest.mark.parametrize("inp,want", [([1, 2, 3], [2, 3]), ([5, 1, 4, 9, 2], [5, 9, 2])])
def test_maxpool_kat_1d(inp, want):
    x = _nhwc(inp).reshape(1, len(inp), 1, 1)
    a = rn.maxpool2_same(x)
    t = rt.to_nhwc(rt.maxpool2_same(rt.to_nchw(x)))
    np.testing.assert_array_equal(a.ravel(), want)
    np.testing.assert_array_equal(t.ravel(), want)
    # and along W
    x = _nhwc(inp).reshape(1, 1, len(inp), 1)
    np.testing.assert_array_equal(rn.maxpool2_same(x).ravel(), want)
    np.testing.assert_array_equal(rt.to_nhwc(rt.maxpool2_same(rt.to_nchw(x))).ravel(), want)


@pytest.mark.parametrize("inp,s,want", [
    ([0, 1, 2, 3], 2, [0, .5, 1, 1.5, 2, 2.5, 3, 3]),
    ([0, 4], 4, [0, 1, 2, 3, 4, 4, 4, 4]),
])
def test_resize_legacy_kat(inp, s, want):
    # column vector: x dimension must be >= 1; use W=1 (hi index clamps to 0)
    x = _nhwc(inp).reshape(1, len(inp), 1, 1)
    a = rn.resize_bilinear_legacy(x, s)[0, :, 0, 0]
    t = rt.to_nhwc(rt.resize_bilinear_legacy(rt.to_nchw(x), s))[0, :, 0, 0]
    np.testing.assert_allclose(a, want)
    np.testing.assert_allclose(t, want)
    x = _nhwc(inp).reshape(1, 1, len(inp), 1)
    np.testing.assert_allclose(rn.resize_bilinear_legacy(x, s)[0, 0, :, 0], want)
    np.testing.assert_allclose(rt.to_nhwc(rt.resize_bilinear_legacy(rt.to_nchw(x), s))[0, 0, :, 0], want)


def test_resize_is_not_half_pixel():
    x = _nhwc([0, 1, 2, 3]).reshape(1, 1, 4, 1)
    half = torch.nn.functional.interpolate(rt.to_nchw(np.repeat(x, 2, axis=1)), scale_factor=2,
                                           mode="bilinear", align_corners=False)
    leg = rt.resize_bilinear_legacy(rt.to_nchw(np.repeat(x, 2, axis=1)), 2)
    assert not torch.allclose(half, leg)


def test_conv_transpose_kat_1d():
    a_, b_ = 2.0, 3.0
    w0, w1, w2 = 5.0, 7.0, 11.0
    # along H: kernel only in column kx=0 so the W axis sees a single tap at ox=0
    x = _nhwc([a_, b_]).reshape(1, 2, 1, 1)
    w = np.zeros((3, 3, 1, 1), np.float32)
    w[:, 0, 0, 0] = [w0, w1, w2]
    want = [a_ * w0, a_ * w1, a_ * w2 + b_ * w0, b_ * w1]
    got = rn.conv_transpose3x3_s2_same(x, w, np.zeros(1, np.float32))
    assert got.shape == (1, 4, 2, 1)
    np.testing.assert_allclose(got[0, :, 0, 0], want)
    np.testing.assert_allclose(got[0, :, 1, 0], 0)
    tw = torch.tensor(w).permute(3, 2, 0, 1).contiguous()
    gt = rt.to_nhwc(rt.conv_transpose3x3_s2_same(rt.to_nchw(x), tw, torch.zeros(1)))
    np.testing.assert_allclose(gt[0, :, 0, 0], want)
    # the padding=1/output_padding=1 variant is shifted by one pixel: must differ
    alt = torch.nn.functional.conv_transpose2d(rt.to_nchw(x), tw, stride=2, padding=1, output_padding=1)
    assert not np.allclose(rt.to_nhwc(alt)[0, :, 0, 0], want)


def test_conv_transpose_kernel_axis_order():
    # kernel is (kh,kw,OUT,IN): in ch 1 -> out ch 0 with weight 2 at tap (1,1)
    x = np.zeros((1, 1, 1, 2), np.float32)
    x[0, 0, 0] = [3, 5]
    w = np.zeros((3, 3, 3, 2), np.float32)
    w[1, 1, 0, 1] = 2
    b = _nhwc([0.5, 0, 0])
    got = rn.conv_transpose3x3_s2_same(x, w, b)
    assert got.shape == (1, 2, 2, 3)
    np.testing.assert_allclose(got[0, 1, 1], [10.5, 0, 0])
    np.testing.assert_allclose(got[0, 0, 0], [0.5, 0, 0])
    tw = torch.tensor(w).permute(3, 2, 0, 1).contiguous()
    gt = rt.to_nhwc(rt.conv_transpose3x3_s2_same(rt.to_nchw(x), tw, torch.tensor(b)))
    np.testing.assert_allclose(gt, got)


@pytest.mark.parametrize("fy,want", [(0, [0, 1, 2, 3]), (1, [0, 0, 1, 2]), (-1, [1, 2, 3, 3]),
                                     (0.5, [0, 0.5, 1.5, 2.5]), (-5, [3, 3, 3, 3])])
def test_warp_kat(fy, want):
    img = np.arange(4, dtype=np.float32).reshape(1, 4, 1, 1).repeat(2, axis=2)   # W=2 (needs >=2)
    flow = np.zeros((1, 4, 2, 2), np.float32)
    flow[..., 0] = fy
    a = rn.dense_image_warp(img, flow)[0, :, 0, 0]
    t = rt.to_nhwc(rt.dense_image_warp(rt.to_nchw(img), rt.to_nchw(flow)))[0, :, 0, 0]
    np.testing.assert_allclose(a, want)
    np.testing.assert_allclose(t, want)


def test_warp_flow_channel_convention():
    # channel 1 is the column displacement
    img = np.arange(4, dtype=np.float32).reshape(1, 1, 4, 1).repeat(2, axis=1)
    flow = np.zeros((1, 2, 4, 2), np.float32)
    flow[..., 1] = 1
    np.testing.assert_allclose(rn.dense_image_warp(img, flow)[0, 0, :, 0], [0, 0, 1, 2])
    t = rt.to_nhwc(rt.dense_image_warp(rt.to_nchw(img), rt.to_nchw(flow)))[0, 0, :, 0]
    np.testing.assert_allclose(t, [0, 0, 1, 2])


def test_space_to_depth_kat():
    x = _nhwc([[1, 2], [3, 4]]).reshape(1, 2, 2, 1)
    np.testing.assert_array_equal(rn.space_to_depth(x, 2).ravel(), [1, 2, 3, 4])
    np.testing.assert_array_equal(rt.to_nhwc(rt.space_to_depth(rt.to_nchw(x), 2)).ravel(), [1, 2, 3, 4])
    # multi-channel ordering: (dy*bs+dx)*C + c
    x = np.arange(2 * 4 * 4 * 3, dtype=np.float32).reshape(2, 4, 4, 3)
    a = rn.space_to_depth(x, 4)
    assert a.shape == (2, 1, 1, 48)
    for dy in range(4):
        for dx in range(4):
            for c in range(3):
                assert a[1, 0, 0, (dy * 4 + dx) * 3 + c] == x[1, dy, dx, c]
    np.testing.assert_array_equal(rt.to_nhwc(rt.space_to_depth(rt.to_nchw(x), 4)), a)


def test_activations():
    x = _nhwc([-2, -0.5, 0, 0.5, 2])
    np.testing.assert_allclose(rn.leaky_relu(x), [-0.4, -0.1, 0, 0.5, 2], rtol=1e-6)
    np.testing.assert_allclose(rn.relu(x), [0, 0, 0, 0.5, 2])
    np.testing.assert_allclose(rn.clip01(x), [0, 0, 0, 0.5, 1])


def test_weight_specs_count_and_params():
    specs = rn.weight_specs()
    assert len(specs) == 76
    assert sum(1 for n, _ in specs if n.startswith("FNet/")) == 28
    assert sum(1 for n, _ in specs if n.startswith("SRNet/")) == 48
    fn = sum(int(np.prod(s)) for n, s in specs if n.startswith("FNet/"))
    sn = sum(int(np.prod(s)) for n, s in specs if n.startswith("SRNet/"))
    assert (fn, sn) == (1745506, 843587)
    names = [n for n, _ in specs]
    assert names[0] == "FNet/conv2d_1/kernel:0" and names[27] == "FNet/conv2d_14/bias:0"
    assert names[28] == "SRNet/conv2d_15/kernel:0"
    assert names[70] == "SRNet/conv2d_transpose_1/kernel:0"
    assert names[74] == "SRNet/conv2d_36/kernel:0"


def test_psnr_and_denormalize():
    a = np.full((4, 4, 3), 0.5, np.float32)
    assert rn.psnr(a, a) == 100
    assert abs(rn.psnr(a, a + 0.1) - 20.0) < 1e-4
    assert rn.denormalize(np.array([0.999, 0.5, 0.0])).tolist() == [254, 127, 0]   # truncation
    assert rn.normalize(np.array([255, 0], np.uint8)).tolist() == [1.0, 0.0]


def test_rng_is_stable():
    u = rn.uniform01(7, 4)
    assert u.min() >= 0 and u.max() < 1
    np.testing.assert_array_equal(u, rn.uniform01(7, 4))
    assert not np.array_equal(u, rn.uniform01(8, 4))


# ---- definitional checks, on random tensors ----------------------------------------------
@pytest.mark.parametrize("H,W,Ci,Co", [(5, 7, 3, 4), (4, 4, 2, 2), (1, 6, 1, 3)])
def test_conv_transpose_is_the_gradient_of_the_stride2_same_conv(H, W, Ci, Co):
    """A.5 says what Conv2DTranspose(3x3, stride 2, 'same') IS in TF: conv2d_backprop_input of a stride-2 SAME
    conv over a 2H x 2W tensor, whose SAME padding at even size and stride 2 is 0 before / 1 after.  Autograd
    of exactly that forward conv must reproduce both restatements (a shifted 'padding=1, output_padding=1'
    variant would not)."""
    rng = np.random.default_rng(H * 10 + W)
    x = rng.standard_normal((2, H, W, Ci)).astype(np.float32)            # the transposed conv's input = dY of the conv
    w = rng.standard_normal((3, 3, Co, Ci)).astype(np.float32)           # Keras layout (kh, kw, filters, input_dim)
    b = rng.standard_normal(Co).astype(np.float32)
    # forward conv: u (2H, 2W, Co channels) -> (H, W, Ci), kernel HWIO (3, 3, Co, Ci) = the same array
    u = torch.zeros(2, Co, 2 * H, 2 * W, dtype=torch.float64, requires_grad=True)
    wf = torch.tensor(w, dtype=torch.float64).permute(3, 2, 0, 1)        # (out = Ci, in = Co, kh, kw)
    y = torch.nn.functional.conv2d(torch.nn.functional.pad(u, (0, 1, 0, 1)), wf, stride=2)
    assert y.shape == (2, Ci, H, W)
    (g,) = torch.autograd.grad(y, u, torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2))
    want = g.permute(0, 2, 3, 1).numpy() + b
    got_np = rn.conv_transpose3x3_s2_same(x, w, b)
    got_t = rt.to_nhwc(rt.conv_transpose3x3_s2_same(rt.to_nchw(x), torch.tensor(w).permute(3, 2, 0, 1).contiguous(),
                                                    torch.tensor(b)))
    assert got_np.shape == (2, 2 * H, 2 * W, Co)
    np.testing.assert_allclose(got_np, want, atol=2e-5)
    np.testing.assert_allclose(got_t, want, atol=2e-5)


@pytest.mark.parametrize("H,W", [(5, 7), (4, 6), (1, 3), (23, 40)])
def test_maxpool_same_is_minus_inf_padding_bottom_right(H, W):
    """A.2: SAME pooling at stride 2 pads only after the data (bottom / right) and the pad never wins."""
    rng = np.random.default_rng(H + W)
    x = -np.abs(rng.standard_normal((2, H, W, 3))).astype(np.float32) - 1.0     # all negative: a zero pad would win
    xp = torch.nn.functional.pad(rt.to_nchw(x), (0, W % 2, 0, H % 2), value=float("-inf"))
    want = rt.to_nhwc(torch.nn.functional.max_pool2d(xp, 2))
    np.testing.assert_array_equal(rn.maxpool2_same(x), want)
    np.testing.assert_array_equal(rt.to_nhwc(rt.maxpool2_same(rt.to_nchw(x))), want)


def test_conv_same_is_zero_padding_one_cross_correlation():
    """A.1 against torch's own conv2d(padding=1) on random data (no kernel flip, HWIO -> OIHW)."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 6, 5, 4)).astype(np.float32)
    w = rng.standard_normal((3, 3, 4, 3)).astype(np.float32)
    b = rng.standard_normal(3).astype(np.float32)
    want = rt.to_nhwc(torch.nn.functional.conv2d(rt.to_nchw(x, torch.float64), torch.tensor(w, dtype=torch.float64).permute(3, 2, 0, 1),
                                                 torch.tensor(b, dtype=torch.float64), padding=1))
    np.testing.assert_allclose(rn.conv3x3_same(x, w, b), want, atol=2e-5)


# ------------------------------------------------------------------ discriminator oracle (oracle/ref_disc.py)
def test_disc_conv4x4_stride2_same_by_definition():
    """TF SAME at k=4, stride 2 on an even size: out = in / 2, total padding 2 split 1 before / 1 after, cross-correlation:
    y[i, j, o] = sum_{ky, kx, c} x[2i + ky - 1, 2j + kx - 1, c] * w[ky, kx, c, o] -- spelled out in loops."""
    from oracle import ref_disc as rd
    H, W = 16, 16
    w = rd.make_disc_weights(1, H, W)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, H, W, 27)).astype(np.float32)
    dw = rd.DiscWeights(w)
    k0, b0 = w["conv2d_37/kernel:0"].astype(np.float64), w["conv2d_37/bias:0"].astype(np.float64)
    a0 = rn.conv3x3_same(x.astype(np.float64), k0, b0)
    a0 = np.where(a0 > 0, a0, 0.2 * a0)
    k1, b1 = w["conv2d_38/kernel:0"].astype(np.float64), w["conv2d_38/bias:0"].astype(np.float64)
    z = np.zeros((H // 2, W // 2, 64))
    xp = np.pad(a0[0], ((1, 1), (1, 1), (0, 0)))
    for i in range(H // 2):
        for j in range(W // 2):
            z[i, j] = np.tensordot(xp[2 * i:2 * i + 4, 2 * j:2 * j + 4], k1, axes=([0, 1, 2], [0, 1, 2])) + b1
    # inference-mode BatchNormalization with the moving statistics, then LeakyReLU(0.2)
    g, be, mm, mv = (w[f"batch_normalization_1/{n}:0"].astype(np.float64) for n in ("gamma", "beta", "moving_mean", "moving_variance"))
    y = (z - mm) / np.sqrt(mv + 1e-3) * g + be
    y = np.where(y > 0, y, 0.2 * y)
    _, feats, _, _ = rd.disc_forward(dw, rt.to_nchw(x, torch.float64), training=False)
    np.testing.assert_allclose(rt.to_nhwc(feats[0])[0], y, atol=1e-10)


def test_disc_keras_bce_clips_probabilities():
    from oracle import ref_disc as rd
    p = torch.tensor([0.0, 1.0, 0.5, 0.25], dtype=torch.float64)
    t = torch.tensor([1.0, 0.0, 1.0, 0.0], dtype=torch.float64)
    want = np.mean([-np.log(1e-7), -np.log(1e-7), -np.log(0.5), -np.log(0.75)])     # 1 - (1 - 1e-7) = 1e-7
    assert abs(float(rd.keras_bce(p, t)) - want) <= 1e-6


def test_disc_training_mode_statistics():
    """Training mode normalises with the biased batch variance and feeds the moving variance var * n / (n - (1 + eps))."""
    from oracle import ref_disc as rd
    B, H, W = 2, 16, 16
    w = rd.make_disc_weights(2, H, W)
    rng = np.random.default_rng(1)
    imgs = [rng.random((B, H, W, 3)).astype(np.float32) for _ in range(9)]
    dw = rd.DiscWeights(w)
    _, feats, _, stats = rd.disc_forward(dw, rd.concat9(imgs), training=True)
    g, be = (w[f"batch_normalization_4/{n}:0"].astype(np.float64) for n in ("gamma", "beta"))
    y = rt.to_nhwc(feats[3]).astype(np.float64)
    zn = (np.where(y > 0, y, y / 0.2) - be) / g                 # undo LeakyReLU and the affine part
    assert np.abs(zn.mean(axis=(0, 1, 2))).max() <= 1e-9
    n = B * (H // 16) * (W // 16)
    mean, var_fed = stats[3]
    z_var = (zn.var(axis=(0, 1, 2)))                            # = var / (var + eps)
    var = z_var * 1e-3 / (1 - z_var)
    np.testing.assert_allclose(var_fed.numpy(), var * n / (n - (1 + 1e-3)), rtol=1e-6)
    _, _, new = rd.disc_loss_and_grads(w, imgs, np.ones((B, 1)))
    np.testing.assert_allclose(new["batch_normalization_4/moving_mean:0"],
                               0.99 * w["batch_normalization_4/moving_mean:0"] + 0.01 * mean.numpy(), rtol=1e-6, atol=1e-9)


def test_disc_weight_specs_count():
    from oracle import ref_disc as rd
    specs = rd.disc_weight_specs(128, 128)
    assert len(specs) == 28
    assert sum(int(np.prod(s)) for _, s in specs) == 27 * 9 * 64 + 64 + 16 * (64 * 64 * 2 + 64 * 128 + 128 * 256) + (64 + 64 + 128 + 256) * 5 + 8 * 8 * 256 + 1
