"""GPU parity of every single op through the C ABI against the numpy oracle.
fp32 tolerance: max-abs <= 1e-3 on [0,1]-scale data is the north_star bar; single
ops sit far inside it, so the per-op bounds here are tighter (stated per test)."""
import numpy as np
import pytest

from oracle import ref_numpy as rn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sh():
    import sr_hip
    assert sr_hip.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return sr_hip


def rnd(seed, *shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return (rn.uniform01(seed, n).reshape(shape) * (hi - lo) + lo).astype(np.float32)


CONV_CASES = [  # B, H, W, Cin, Cout
    (1, 8, 32, 16, 32), (2, 16, 16, 6, 32), (1, 24, 40, 32, 64), (1, 9, 35, 64, 64),
    (1, 8, 8, 128, 256), (1, 3, 5, 256, 128), (2, 16, 64, 51, 64), (1, 16, 48, 64, 3),
    (1, 8, 40, 32, 2), (1, 5, 7, 5, 7), (1, 1, 1, 3, 4), (1, 40, 33, 48, 96),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout", CONV_CASES)
@pytest.mark.parametrize("act", ["none", "relu", "lrelu", "tanh", "clip01"])
def test_conv3x3_fp32(sh, B, H, W, Cin, Cout, act):
    if act not in ("none", "lrelu") and (Cin, Cout) not in ((64, 64), (64, 3), (32, 2)):
        pytest.skip("activation variants only on the hot-path shapes")
    x = rnd(1, B, H, W, Cin)
    w = rnd(2, 3, 3, Cin, Cout, lo=-0.2, hi=0.2)
    b = rnd(3, Cout, lo=-0.5, hi=0.5)
    want = rn.conv3x3_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    want = {"none": lambda v: v, "relu": rn.relu, "lrelu": rn.leaky_relu, "tanh": np.tanh, "clip01": rn.clip01}[act](want)
    got = sh.conv3x3(x, w, b, act=act)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


def test_conv3x3_kats(sh):
    x = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.float32).reshape(1, 3, 3, 1)
    got = sh.conv3x3(x, np.ones((3, 3, 1, 1), np.float32), np.zeros(1, np.float32))
    np.testing.assert_allclose(got[0, :, :, 0], [[12, 21, 16], [27, 45, 33], [24, 39, 28]], atol=1e-5)
    w = np.zeros((3, 3, 1, 1), np.float32)
    w[0, 0] = 1
    got = sh.conv3x3(x, w, np.zeros(1, np.float32))
    np.testing.assert_allclose(got[0, :, :, 0], [[0, 0, 0], [0, 1, 2], [0, 4, 5]], atol=1e-6)


def test_conv3x3_residual(sh):
    x = rnd(4, 1, 16, 32, 64)
    res = rnd(5, 1, 16, 32, 64)
    w = rnd(6, 3, 3, 64, 64, lo=-0.1, hi=0.1)
    b = rnd(7, 64)
    want = res + rn.conv3x3_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    got = sh.conv3x3(x, w, b, res=res)
    assert np.abs(got - want).max() <= 2e-5


@pytest.mark.parametrize("dtype,tol", [("bf16x3", 2e-4), ("bf16", 6e-2)])
def test_conv3x3_reduced_precision(sh, dtype, tol):
    x = rnd(8, 1, 16, 40, 64)
    w = rnd(9, 3, 3, 64, 64, lo=-0.1, hi=0.1)
    b = rnd(10, 64)
    want = rn.conv3x3_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    got = sh.conv3x3(x, w, b, dtype=dtype)
    assert np.abs(got - want).max() <= tol


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16x3", 2e-4), ("bf16", 6e-2)])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 19, 70, 64, 3), (1, 8, 40, 32, 2), (1, 33, 100, 32, 1),
                                            (3, 8, 32, 64, 3), (1, 50, 31, 128, 3)])
def test_conv3x3_few_outputs(sh, dtype, tol, B, H, W, Cin, Cout):
    """<= 3 output channels run the taps-in-N kernel (conv_tapsn.hip) where the channel count fits
    its 128-byte chunks, ragged tiles, several images and a residual included."""
    x = rnd(20, B, H, W, Cin)
    w = rnd(21, 3, 3, Cin, Cout, lo=-0.1, hi=0.1)
    b = rnd(22, Cout, lo=-0.5, hi=0.5)
    res = rnd(23, B, H, W, Cout)
    want = rn.conv3x3_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    got = sh.conv3x3(x, w, b, dtype=dtype)
    assert np.abs(got - want).max() <= tol
    got = sh.conv3x3(x, w, b, act="clip01", res=res, dtype=dtype)
    assert np.abs(got - (rn.clip01(want) + res)).max() <= tol


def test_conv_random_shape_sweep(sh):
    """Seeded sweep over ragged shapes, channel counts around the 16 / 32 / 64 boundaries, activations,
    residuals and the three precisions: conv and transposed conv against the fp64 numpy oracle."""
    rng = np.random.default_rng(1234)
    tol = {"fp32": 3e-5, "bf16x3": 3e-4, "bf16": 8e-2}
    for case in range(24):
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(1, 41)), int(rng.integers(1, 71))
        Cin = int(rng.choice([1, 3, 15, 16, 17, 31, 32, 33, 48, 64, 65, 96, 128]))
        Cout = int(rng.choice([1, 2, 3, 4, 16, 31, 32, 33, 64, 65, 96]))
        dtype = ["fp32", "bf16x3", "bf16"][case % 3]
        x = rnd(100 + case, B, H, W, Cin)
        b = rnd(300 + case, Cout, lo=-0.5, hi=0.5)
        if case % 4 == 3:      # transposed conv: Keras kernel layout (kh, kw, out, in)
            w = rnd(200 + case, 3, 3, Cout, Cin, lo=-0.1, hi=0.1)
            want = rn.relu(rn.conv_transpose3x3_s2_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64)))
            got = sh.conv_transpose3x3_s2(x, w, b, act="relu", dtype=dtype)
        else:
            w = rnd(200 + case, 3, 3, Cin, Cout, lo=-0.1, hi=0.1)
            act = ["none", "relu", "lrelu"][case % 3]
            res = rnd(400 + case, B, H, W, Cout) if case % 2 else None
            want = rn.conv3x3_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
            want = {"none": lambda v: v, "relu": rn.relu, "lrelu": rn.leaky_relu}[act](want)
            if res is not None:
                want = want + res
            got = sh.conv3x3(x, w, b, act=act, res=res, dtype=dtype)
        assert got.shape == want.shape, (case, got.shape, want.shape)
        err = np.abs(got - want).max()
        assert err <= tol[dtype] * max(1.0, np.abs(want).max()), (case, dtype, B, H, W, Cin, Cout, err)


CONVT_CASES = [(1, 8, 32, 64, 64), (2, 4, 8, 16, 64), (1, 5, 7, 3, 5), (1, 16, 40, 64, 64), (1, 1, 1, 2, 3)]


@pytest.mark.parametrize("B,H,W,Cin,Cout", CONVT_CASES)
def test_conv_transpose_fp32(sh, B, H, W, Cin, Cout):
    x = rnd(11, B, H, W, Cin)
    w = rnd(12, 3, 3, Cout, Cin, lo=-0.2, hi=0.2)
    b = rnd(13, Cout)
    want = rn.relu(rn.conv_transpose3x3_s2_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64)))
    got = sh.conv_transpose3x3_s2(x, w, b, act="relu")
    assert got.shape == (B, 2 * H, 2 * W, Cout)
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


def test_conv_transpose_kat(sh):
    x = np.array([2.0, 3.0], np.float32).reshape(1, 2, 1, 1)
    w = np.zeros((3, 3, 1, 1), np.float32)
    w[:, 0, 0, 0] = [5, 7, 11]
    got = sh.conv_transpose3x3_s2(x, w, np.zeros(1, np.float32))
    np.testing.assert_allclose(got[0, :, 0, 0], [10, 14, 22 + 15, 21], atol=1e-5)
    np.testing.assert_allclose(got[0, :, 1, 0], 0, atol=1e-6)


@pytest.mark.parametrize("dtype,tol", [("bf16x3", 2e-4), ("bf16", 6e-2)])
def test_conv_transpose_reduced_precision(sh, dtype, tol):
    x = rnd(14, 1, 8, 40, 64)
    w = rnd(15, 3, 3, 64, 64, lo=-0.1, hi=0.1)
    b = rnd(16, 64)
    want = rn.conv_transpose3x3_s2_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    got = sh.conv_transpose3x3_s2(x, w, b, dtype=dtype)
    assert np.abs(got - want).max() <= tol


@pytest.mark.parametrize("shape", [(1, 3, 1, 1), (1, 5, 1, 1), (2, 23, 40, 128), (1, 7, 9, 5), (1, 8, 8, 32)])
def test_maxpool(sh, shape):
    x = rnd(17, *shape)
    np.testing.assert_array_equal(sh.maxpool2(x), rn.maxpool2_same(x))


def test_maxpool_kat(sh):
    x = np.array([5, 1, 4, 9, 2], np.float32).reshape(1, 5, 1, 1)
    np.testing.assert_array_equal(sh.maxpool2(x).ravel(), [5, 9, 2])


@pytest.mark.parametrize("shape,s", [((1, 4, 1, 1), 2), ((1, 2, 1, 1), 4), ((2, 23, 40, 256), 2),
                                     ((1, 16, 24, 2), 4), ((1, 5, 7, 3), 4), ((1, 6, 5, 7), 2), ((1, 3, 3, 1), 3)])
def test_resize_bilinear(sh, shape, s):
    x = rnd(18, *shape)
    want = rn.resize_bilinear_legacy(x, s)
    got = sh.resize_bilinear(x, s)
    assert np.abs(got - want).max() <= 1e-6


def test_resize_kat(sh):
    x = np.array([0, 1, 2, 3], np.float32).reshape(1, 4, 1, 1)
    np.testing.assert_allclose(sh.resize_bilinear(x, 2)[0, :, 0, 0], [0, .5, 1, 1.5, 2, 2.5, 3, 3])
    x = np.array([0, 4], np.float32).reshape(1, 1, 2, 1)
    np.testing.assert_allclose(sh.resize_bilinear(x, 4)[0, 0, :, 0], [0, 1, 2, 3, 4, 4, 4, 4])


@pytest.mark.parametrize("fy,want", [(0, [0, 1, 2, 3]), (1, [0, 0, 1, 2]), (-1, [1, 2, 3, 3]),
                                     (0.5, [0, 0.5, 1.5, 2.5]), (-5, [3, 3, 3, 3])])
def test_warp_kat(sh, fy, want):
    img = np.arange(4, dtype=np.float32).reshape(1, 4, 1, 1).repeat(2, axis=2)
    flow = np.zeros((1, 4, 2, 2), np.float32)
    flow[..., 0] = fy
    np.testing.assert_allclose(sh.warp(img, flow)[0, :, 0, 0], want)


@pytest.mark.parametrize("shape", [(2, 16, 24, 3), (1, 64, 64, 3), (1, 2, 2, 1), (1, 9, 5, 4)])
def test_warp_random(sh, shape):
    img = rnd(19, *shape, lo=0, hi=1)
    flow = rnd(20, shape[0], shape[1], shape[2], 2, lo=-3, hi=3)   # includes out-of-range queries
    want = rn.dense_image_warp(img, flow)
    assert np.abs(sh.warp(img, flow) - want).max() <= 2e-5


@pytest.mark.parametrize("shape,bs", [((1, 2, 2, 1), 2), ((2, 8, 12, 3), 4), ((1, 64, 64, 3), 4), ((1, 6, 9, 2), 3)])
def test_space_to_depth(sh, shape, bs):
    x = rnd(21, *shape)
    np.testing.assert_array_equal(sh.space_to_depth(x, bs), rn.space_to_depth(x, bs))


def test_error_paths(sh):
    with pytest.raises(ValueError):
        sh.space_to_depth(np.zeros((1, 5, 4, 1), np.float32), 4)
    with pytest.raises(ValueError):
        sh.warp(np.zeros((1, 1, 4, 1), np.float32), np.zeros((1, 1, 4, 2), np.float32))
    with pytest.raises(ValueError):
        sh.Context(20, 16)            # 20 is not a multiple of 8 (FNet pyramid)
    with pytest.raises(ValueError):
        sh.Context(16, 16, scale=2)
    ctx = sh.Context(16, 16)
    with pytest.raises(ValueError):   # weights not loaded
        ctx.step(np.zeros((1, 16, 16, 3)), np.zeros((1, 16, 16, 3)), np.zeros((1, 64, 64, 3)))
    ctx.close()


@pytest.mark.parametrize("N,h,w,C", [(2, 4, 6, 3), (1, 1, 1, 3), (3, 32, 32, 3), (1, 2, 9, 1), (1, 45, 80, 3)])
def test_cubic_upscale4(sh, N, h, w, C):
    """The bicubic x4 image of the adversarial batch / the metrics baseline (TecoGAN.py:313, FRVSR.py:304) against the numpy
    restatement: same weights, same order of products and sums, no fused multiply-adds."""
    import utils
    lr = rnd(22, N, h, w, C, lo=0.0, hi=1.0)
    got = sh.cubic_upscale4(lr)
    want = utils.cubic_upscale4(lr)
    assert got.shape == (N, 4 * h, 4 * w, C)
    assert np.abs(got - want).max() <= 1e-6
    const = np.full((1, 3, 5, 3), 0.37, np.float32)              # weights sum to one, borders clamp
    assert np.abs(sh.cubic_upscale4(const) - 0.37).max() <= 1e-6


@pytest.mark.parametrize("N,H,W,C", [(2, 16, 24, 3), (1, 4, 4, 3), (3, 64, 64, 3), (1, 8, 12, 1), (1, 256, 256, 3)])
def test_blur_cubic_downscale(sh, N, H, W, C):
    """LR synthesis of the reference's drivers (FRVSR.py:117-119, 274-277) as one fused gather against the numpy
    restatement (two blur passes, two cubic passes): same taps, different summation order -> 2e-6."""
    import utils
    hr = rnd(21, N, H, W, C, lo=0.0, hi=1.0)
    got = sh.blur_cubic_downscale(hr)
    want = utils.blur_cubic_downscale(hr)
    assert got.shape == (N, H // 4, W // 4, C)
    assert np.abs(got - want).max() <= 2e-6
    const = np.full((1, 8, 8, 3), 0.37, np.float32)              # weights sum to one, borders reflect
    assert np.abs(sh.blur_cubic_downscale(const) - 0.37).max() <= 1e-6
    with pytest.raises(ValueError):
        sh.blur_cubic_downscale(np.zeros((1, 6, 8, 3), np.float32))
