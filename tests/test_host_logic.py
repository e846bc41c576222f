"""Host-side logic of the drop-in facade (no GPU): names, weight layout, files, tiling, config."""
import os

import numpy as np
import pytest
import yaml

import layers
import main as sr_main
import utils
import weights_io
from FRVSR import FrameRecurrentVideoSR
from TecoGAN import TecoGAN
from oracle import ref_numpy as rn

CFG = {'root_dir': './', 'data': {'rows': 2, 'cols': 2}, 'cnn': {'model': 'frvsr', 'pretrained_model': None}}


def test_weight_specs_match_oracle_layout():
    assert layers.weight_specs() == rn.weight_specs()
    assert len(layers.weight_specs()) == 76


def test_model_attributes_and_names():
    m = FrameRecurrentVideoSR((64, 64, 3), (256, 256, 3), '/tmp/o', CFG)
    m.build()
    # attribute and model names of reference FRVSR.py:30-50, layers.py:48,79,92,116,139
    assert [m.fnet.name, m.srnet.name, m.upscaling.name, m.hr_warp.name, m.lr_warp.name, m.space2depth.name] == \
        ['FNet', 'SRNet', 'Upscaling_0', 'Warp_1', 'Warp_2', 'Space_to_Depth']
    assert m.frvsr.count_params() == 2589093
    assert m.fnet.count_params() == 1745506 and m.srnet.count_params() == 843587
    t = TecoGAN((64, 64, 3), (256, 256, 3), '/tmp/o', dict(CFG, cnn={'model': 'tecogan'}))
    t.build()
    assert t.generator.name == 'Generator' and t.hr_warping.name == 'HR_Warping'
    assert t.bicubic_upscale.name == 'Upscaling_1' and not hasattr(t, 'frvsr')
    assert t.core.model == 'tecogan'


def test_fresh_model_uses_keras_default_initialisers():
    core = layers.GeneratorCore((16, 16, 3), seed=1)
    for name, a in core.tensors:
        if a.ndim == 4:
            lim = np.sqrt(6.0 / (9 * a.shape[2] + 9 * a.shape[3]))
            assert np.abs(a).max() <= lim and np.abs(a).max() > 0.5 * lim
        else:
            assert not a.any()


def test_shape_checks():
    with pytest.raises(ValueError):
        FrameRecurrentVideoSR((60, 64, 3), (240, 256, 3), '/tmp/o', CFG).build()      # 60 % 8
    with pytest.raises(ValueError):
        FrameRecurrentVideoSR((64, 64, 3), (128, 128, 3), '/tmp/o', CFG).build()      # not x4
    with pytest.raises(ValueError):
        layers.GeneratorCore((16, 16, 1))


def test_weights_npz_roundtrip_and_position_matching(tmp_path, weights7):
    m = FrameRecurrentVideoSR((16, 16, 3), (64, 64, 3), str(tmp_path), CFG)
    m.build()
    m.frvsr.set_weights(list(weights7.values()))
    p = str(tmp_path / 'w.npz')
    m.frvsr.save_weights(p)
    tensors = weights_io.read_tensors(p)
    assert [n for n, _ in tensors] == list(weights7.keys())
    m2 = FrameRecurrentVideoSR((16, 16, 3), (64, 64, 3), str(tmp_path), CFG)
    m2.build()
    m2.frvsr.load_weights(p)
    for a, b in zip(m2.frvsr.get_weights(), weights7.values()):
        np.testing.assert_array_equal(a, b)
    # sub-models see the shared variables (reference: fnet/srnet share weights with frvsr)
    np.testing.assert_array_equal(m2.fnet.get_weights()[0], weights7['FNet/conv2d_1/kernel:0'])
    np.testing.assert_array_equal(m2.srnet.get_weights()[-1], weights7['SRNet/conv2d_36/bias:0'])
    # FRVSR and TecoGAN generator files are interchangeable (same two weighted sub-models)
    t = TecoGAN((16, 16, 3), (64, 64, 3), str(tmp_path), CFG)
    t.build()
    t.generator.load_weights(p)
    np.testing.assert_array_equal(t.generator.get_weights()[5], list(weights7.values())[5])
    # a sub-model file holds only its group
    pf = str(tmp_path / 'fnet.npz')
    m2.fnet.save_weights(pf)
    assert len(weights_io.read_tensors(pf)) == 28
    # names with other numeric suffixes still load (Keras matches by position)
    renamed = [(n.replace('conv2d_', 'conv2d_1'), a) for n, a in tensors]
    weights_io.write_tensors(str(tmp_path / 'r.npz'), renamed)
    m2.frvsr.load_weights(str(tmp_path / 'r.npz'))
    # wrong shapes / counts are rejected
    bad = list(tensors)
    bad[0] = (bad[0][0], np.zeros((3, 3, 6, 31), np.float32))
    weights_io.write_tensors(str(tmp_path / 'bad.npz'), bad)
    with pytest.raises(ValueError):
        m2.frvsr.load_weights(str(tmp_path / 'bad.npz'))
    weights_io.write_tensors(str(tmp_path / 'short.npz'), tensors[:20])
    with pytest.raises(ValueError):
        m2.frvsr.load_weights(str(tmp_path / 'short.npz'))
    with pytest.raises(OSError):
        m2.frvsr.load_weights(str(tmp_path / 'missing.npz'))


def test_h5_weight_files_without_h5py(tmp_path, weights7):
    """.h5 files go through h5py when present, else through the bundled h5lite (the HDF5 subset h5py's
    default format uses for Keras save_weights).  Round trip in the Keras layout, position matching,
    and the reader against a libhdf5-written file (scipy's MATLAB v7.3 test data)."""
    import h5lite
    m = FrameRecurrentVideoSR((16, 16, 3), (64, 64, 3), str(tmp_path), {"cnn": {}, "data": {}})
    m.build()
    m.frvsr.set_weights(list(weights7.values()))
    path = str(tmp_path / "demo.h5")
    m.frvsr.save_weights(path)
    assert open(path, "rb").read(8) == h5lite.SIGNATURE
    f = h5lite.File(path)
    assert [n.decode() for n in f.attrs["layer_names"]] == ["FNet", "SRNet"]
    assert len(f["FNet"].attrs["weight_names"]) == 28 and len(f["SRNet"].attrs["weight_names"]) == 48
    # Keras 2.2.4 layout: weight_names are the variable names (no group prefix), the dataset sits under its group
    assert f["SRNet"].attrs["weight_names"][0] == b"conv2d_15/kernel:0"
    k = f["SRNet/conv2d_transpose_1/kernel:0"]
    assert k.shape == (3, 3, 64, 64) and k.dtype == np.float32
    assert [n for n, _ in weights_io.read_tensors(path)] == list(weights7.keys())
    m2 = FrameRecurrentVideoSR((16, 16, 3), (64, 64, 3), str(tmp_path), {"cnn": {}, "data": {}})
    m2.build()
    m2.frvsr.load_weights(path)
    for a, b in zip(m2.frvsr.get_weights(), weights7.values()):
        np.testing.assert_array_equal(a, b)
    # sub-model loads only its own group, by position
    m2.fnet.load_weights(path)
    # a file as Keras writes it for the whole model: every layer listed, the weightless ones with empty groups
    full = str(tmp_path / "full.h5")
    weights_io.write_tensors(full, list(weights7.items()), weightless_layers=["input_1", "input_2", "Upscaling_0", "Warp_1"])
    assert [n.decode() for n in h5lite.File(full).attrs["layer_names"]][:4] == ["input_1", "input_2", "Upscaling_0", "Warp_1"]
    m3 = FrameRecurrentVideoSR((16, 16, 3), (64, 64, 3), str(tmp_path), {"cnn": {}, "data": {}})
    m3.build()
    m3.frvsr.load_weights(full)
    for a, b in zip(m3.frvsr.get_weights(), weights7.values()):
        np.testing.assert_array_equal(a, b)
    # the older layout of this repo (weight_names carrying the group prefix) still loads
    old = {"attrs": {"layer_names": np.array([b"FNet", b"SRNet"])}, "children": {}}
    for g in ("FNet", "SRNet"):
        names = [n for n in weights7 if n.startswith(g + "/")]
        node = {"attrs": {"weight_names": np.array([n.encode() for n in names])}, "children": {}}
        for n in names:
            cur = node
            parts = n.split("/")
            for part in parts[:-1]:
                cur = cur["children"].setdefault(part, {"attrs": {}, "children": {}})
            cur["children"][parts[-1]] = weights7[n]
        old["children"][g] = node
    h5lite.write_file(str(tmp_path / "old.h5"), old)
    assert [n for n, _ in weights_io.read_tensors(str(tmp_path / "old.h5"))] == list(weights7.keys())
    # a real libhdf5 file: superblock at offset 512, old-style group, version-2 layout message
    import scipy.io
    real = os.path.join(os.path.dirname(scipy.io.__file__), "matlab", "tests", "data", "testhdf5_7.4_GLNX86.mat")
    if os.path.exists(real):
        d = h5lite.File(real)["testdouble"]
        assert d.shape == (9, 1) and d.attrs["MATLAB_class"] == b"double"
        np.testing.assert_allclose(d[()].ravel(), np.linspace(0, 2 * np.pi, 9), atol=1e-12)
    bad = tmp_path / "bad.h5"
    bad.write_bytes(b"not an hdf5 file" * 8)
    with pytest.raises(ValueError):
        weights_io.read_tensors(str(bad))


def test_utils_semantics():
    x = np.array([0, 127, 255], np.uint8)
    assert utils.normalize(x).dtype == np.float32 and utils.normalize(x)[2] == 1.0
    assert utils.denormalize(np.array([0.999, 0.5])).tolist() == [254, 127]      # truncation (utils.py:61-63)
    a = np.full((2, 2, 3), 0.25, np.float32)
    assert utils.psnr(a, a) == 100
    assert abs(utils.psnr(a, a + 0.1) - rn.psnr(a, a + 0.1)) < 1e-12


def test_ssim_restatement():
    """utils.ssim restates cv2 gray + skimage-0.14 compare_ssim defaults (reference utils.py:74-77);
    neither package is installed, so this is a definition-level check (PARITY UNPINNED): identical
    images give 1, the value equals a direct windowed evaluation of the SSIM formula with
    reflect-padded 7x7 windows, sample covariance, data_range 2, borders cropped by 3."""
    import utils
    rng = np.random.default_rng(3)
    a = rng.random((12, 15, 3)).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape).astype(np.float32), 0, 1)
    assert abs(utils.ssim(a, a) - 1.0) < 1e-12
    coef = np.array([0.114, 0.587, 0.299])
    x, y = a.astype(np.float64) @ coef, b.astype(np.float64) @ coef
    c1, c2 = (0.01 * 2) ** 2, (0.03 * 2) ** 2
    vals = []
    for i in range(3, 12 - 3):          # windows fully inside: the reflect padding never matters after the crop
        for j in range(3, 15 - 3):
            wx, wy = x[i - 3:i + 4, j - 3:j + 4].ravel(), y[i - 3:i + 4, j - 3:j + 4].ravel()
            mx, my = wx.mean(), wy.mean()
            vx, vy = wx.var(ddof=1), wy.var(ddof=1)
            vxy = ((wx - mx) * (wy - my)).sum() / 48
            vals.append((2 * mx * my + c1) * (2 * vxy + c2) / ((mx * mx + my * my + c1) * (vx + vy + c2)))
    got = utils.ssim(a, b)
    assert abs(got - np.mean(vals)) < 1e-6
    assert 0 < got < 1
    with pytest.raises(ValueError):
        utils.ssim(a[:5], b[:5])


def test_blur_cubic_downscale_kats():
    """Restatement of cv2.GaussianBlur(5x5, sigma 0) + cv2.resize(INTER_CUBIC) x1/4 (reference
    FRVSR.py:274-277; OpenCV absent: parity unpinned).  Constants are preserved, an impulse spreads
    with the separable product of the [1,4,6,4,1]/16 and [-3,19,19,-3]/32 kernels."""
    import utils
    c = np.full((1, 16, 24, 3), 0.37, np.float32)
    np.testing.assert_allclose(utils.blur_cubic_downscale(c), 0.37, atol=1e-6)
    imp = np.zeros((1, 16, 16, 3), np.float32)
    imp[0, 9, 6, 1] = 1.0
    out = utils.blur_cubic_downscale(imp)
    assert out.shape == (1, 4, 4, 3)
    g, k = np.array([1, 4, 6, 4, 1]) / 16, np.array([-3, 19, 19, -3]) / 32
    def resp(pos):          # 1-D response of output d to an impulse at pos
        b = np.zeros(16 + 4); b[pos:pos + 5] = g; b = b[2:-2]            # blurred impulse (interior: no border)
        return np.array([b[4 * d:4 * d + 4] @ k for d in range(4)])
    np.testing.assert_allclose(out[0, :, :, 1], np.outer(resp(9), resp(6)), atol=1e-7)
    assert out[..., 0].max() == 0 and out[..., 2].max() == 0          # channels stay separate
    with pytest.raises(ValueError):
        utils.blur_cubic_downscale(np.zeros((1, 6, 8, 3), np.float32))


def test_cubic_upscale4_kats():
    """Restatement of cv2.resize(x4, INTER_CUBIC) (reference FRVSR.py:304; OpenCV absent: parity
    unpinned): constants are reproduced, a ramp and an impulse come out with the hand-evaluated weights
    of the a = -0.75 kernel (which, unlike a = -0.5, does not reproduce linear ramps exactly)."""
    import utils
    c = np.full((1, 5, 6, 3), 0.25, np.float32)
    np.testing.assert_allclose(utils.cubic_upscale4(c), 0.25, atol=1e-6)
    ramp = np.tile(np.arange(8, dtype=np.float32)[None, None, :, None], (1, 4, 1, 3))
    up = utils.cubic_upscale4(ramp)
    assert up.shape == (1, 16, 32, 3)
    # destination 8 samples source 1.625: taps on pixels 0..3 with phase t = 0.625
    a, t = -0.75, 0.625
    c0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    c1 = ((a + 2) * t - (a + 3)) * t * t + 1
    c2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    c3 = 1 - c0 - c1 - c2
    assert abs(up[0, 5, 8, 0] - (0 * c0 + 1 * c1 + 2 * c2 + 3 * c3)) < 1e-5
    assert abs(up[0, 5, 8, 0] - 1.595703) < 1e-5
    # left border: destination 0 samples source -0.375 -> taps -2..1 clamped to pixel 0, 0, 0, 1
    t = 0.625
    assert abs(up[0, 0, 0, 0] - (1 * c3)) < 1e-5
    imp = np.zeros((1, 1, 9, 3), np.float32)
    imp[0, 0, 4, 0] = 1
    # destination 18 samples source 4.125: tap weights of phase 0.125 around pixel 4
    t = 0.125
    w1 = ((-0.75 + 2) * t - (-0.75 + 3)) * t * t + 1
    assert abs(utils.cubic_upscale4(imp)[0, 0, 18, 0] - w1) < 1e-6


def test_tiling_roundtrip_and_order():
    frame = np.arange(2 * 4 * 3 * 6 * 3, dtype=np.float32).reshape(8, 18, 3)
    tiles = utils.split_tiles(frame, 2, 3, 4, 6)
    assert tiles.shape == (6, 4, 6, 3)
    np.testing.assert_array_equal(tiles[1 * 3 + 2], frame[4:8, 12:18])          # index i*cols + j
    np.testing.assert_array_equal(utils.merge_tiles(tiles, 2, 3), frame)


def test_shard_range_contiguous_blocks():
    assert [utils.shard_range(128, 8, r) for r in range(8)] == [(16 * r, 16 * r + 16) for r in range(8)]
    assert [utils.shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert utils.shard_range(2, 4, 3) == (2, 2)


def test_yaml_surface(tmp_path):
    for name in ('FRVSR_example.yaml', 'TecoGAN_example.yaml', '180p.yaml'):
        cfg = yaml.safe_load(open(os.path.join(os.path.dirname(layers.__file__), 'options', name)))
        lr, hr = sr_main.shapes_from_config(cfg)
        assert lr[0] * 4 == hr[0] and lr[1] * 4 == hr[1]
        model = sr_main.make_model(cfg, str(tmp_path))
        assert type(model).__name__ == {'frvsr': 'FrameRecurrentVideoSR', 'tecogan': 'TecoGAN'}[cfg['cnn']['model']]
    cfg['data']['high_res'] = 100
    with pytest.raises(ValueError):
        sr_main.shapes_from_config(cfg)
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(layers.__file__), 'options', 'FRVSR_example.yaml')))
    cfg['cnn']['model'] = 'edvr'
    with pytest.raises(ValueError):
        sr_main.make_model(cfg, str(tmp_path))


def test_out_of_scope_stages_say_so():
    from TecoGAN import TecoGAN
    m = FrameRecurrentVideoSR((16, 16, 3), (64, 64, 3), '/tmp/o', CFG)
    with pytest.raises(NotImplementedError):
        m.run()
    t = TecoGAN((16, 16, 3), (64, 64, 3), '/tmp/o', dict(CFG))
    with pytest.raises(NotImplementedError, match='ImageNet'):
        t.vgg_54(None, None)                                         # perceptual losses: the weights are a download ...
    import layers
    tm = layers.TecoGANModel(None, None)
    with pytest.raises(ValueError, match='32 weight tensors'):
        tm.set_vgg_weights([np.zeros((3, 3, 3, 64))])               # ... unless train.vgg19_weights supplies all 32 tensors
    tm.set_vgg_weights(None)
    assert tm.vgg_weights is None
    assert [TecoGAN.gen_learning_rate(i) for i in (0, 49999, 50000, 100000, 10 ** 6)] == [1e-4, 1e-4, 5e-5, 2.5e-5, 2.5e-5]


def test_discriminator_model_weights_and_files(tmp_path):
    """layers.Discriminator(HR_shape).build() (reference layers.py:142-186): 28 tensors in Keras order, fresh-model
    initialisation, save_weights in the Keras layout of a model made of plain layers (dataset conv2d_37/conv2d_37/kernel:0),
    load_weights by ORDER whatever the saving process numbered its layers."""
    import layers
    import weights_io
    d = layers.Discriminator((64, 48, 3)).build()
    assert d.name == 'Discriminator' and d.trainable is True
    named = d.named_weights()
    assert len(named) == 28 and d.count_params() == sum(int(np.prod(s)) for _, s in layers.disc_weight_specs((64, 48, 3)))
    shapes = dict((n, a.shape) for n, a in named)
    assert shapes['conv2d_37/kernel:0'] == (3, 3, 27, 64) and shapes['conv2d_41/kernel:0'] == (4, 4, 128, 256)
    assert shapes['dense_1/kernel:0'] == (4 * 3 * 256, 1) and shapes['batch_normalization_4/moving_variance:0'] == (256,)
    w = dict(named)
    assert np.all(w['batch_normalization_2/gamma:0'] == 1) and np.all(w['batch_normalization_2/moving_variance:0'] == 1)
    assert not w['batch_normalization_2/beta:0'].any() and not w['conv2d_38/bias:0'].any() and w['conv2d_38/kernel:0'].std() > 0
    d.save_weights(str(tmp_path / 'd.h5'))
    back = weights_io.read_tensors(str(tmp_path / 'd.h5'))
    assert [n for n, _ in back] == [n for n, _ in named]
    import h5lite
    with h5lite.File(str(tmp_path / 'd.h5'), 'r') as f:
        assert [x.decode() for x in f['conv2d_37'].attrs['weight_names']] == ['conv2d_37/kernel:0', 'conv2d_37/bias:0']
        assert f['conv2d_37']['conv2d_37/kernel:0'][()].shape == (3, 3, 27, 64)
        # Keras lists EVERY layer of the model, in model.layers order, weightless ones with an empty weight_names
        # attribute (h5py stores the empty list Keras hands it as an empty float64 array) -- ADVICE r2
        names = [x.decode() for x in f.attrs['layer_names']]
        assert names == layers.disc_layer_names() and len(names) == 27
        assert names[:10] == [f'input_{i}' for i in range(12, 21)] + ['concatenate_3'] and names[-3:] == ['flatten_1', 'dense_1', 'activation_2']
        for weightless in ('input_12', 'concatenate_3', 'leaky_re_lu_14', 'flatten_1', 'activation_2'):
            wn = f[weightless].attrs['weight_names']
            assert wn.shape == (0,) and wn.dtype == np.float64 and list(f[weightless].keys()) == []
    # a file written by a process that numbered its layers differently loads by position
    other = [(n.replace('conv2d_3', 'conv2d_').replace('conv2d_4', 'conv2d_1'), a + 1) for n, a in named]
    weights_io.write_tensors(str(tmp_path / 'e.npz'), other)
    d.load_weights(str(tmp_path / 'e.npz'))
    for a, (_, b) in zip(d.get_weights(), named):
        np.testing.assert_array_equal(a, b + 1)
    with pytest.raises(ValueError):
        d.set_weights(d.get_weights()[:-1])
    with pytest.raises(ValueError):
        layers.Discriminator((40, 64, 3)).build()                      # four stride-2 convolutions: multiples of 16
    opt = layers.Adam(lr=5e-5)
    with pytest.raises(NotImplementedError):
        d.compile(loss=['binary_crossentropy'] + ['mean_squared_error'] * 4, loss_weights=[1., 1., 0., 0., 0.], optimizer=opt)
    with pytest.raises(RuntimeError):
        d.train_on_batch([None] * 9, [None] * 5)                       # not compiled
    d.compile(loss=['binary_crossentropy'] + ['mean_squared_error'] * 4, loss_weights=[1., 0., 0., 0., 0.], optimizer=opt)
    assert d.optimizer is opt and opt.iterations == 0 and d._fresh_moments


def test_tecogan_model_compile_contract():
    import layers
    core = layers.GeneratorCore((16, 16, 3), model='tecogan')
    gen, disc = layers.GeneratorModel(core, 'Generator'), layers.Discriminator((64, 64, 3)).build()
    m = layers.TecoGANModel(gen, disc)
    assert len(m.get_weights()) == 76 + 28 and m.train_discriminator
    losses = ['mean_squared_error', 'binary_crossentropy'] + ['mean_squared_error'] * 4 + [print] * 4 + ['mean_squared_error'] * 2
    with pytest.raises(RuntimeError):
        m.train_on_batch([None] * 8, [None] * 12)
    with pytest.raises(NotImplementedError):
        m.compile(loss=losses, loss_weights=[1.] * 11 + [0.5])
    with pytest.raises(NotImplementedError):
        m.compile(loss=['mean_squared_error'] * 12)
    m.compile(loss=losses, loss_weights=layers.TECOGAN_LOSS_WEIGHTS, optimizer=layers.Adam(lr=5e-5))
    assert m.loss_weights[1] == -1e-3 and m.loss_weights[10] == 0.25


def test_clip_loader_rejects_videos(tmp_path):
    (tmp_path / 'a.mp4').write_bytes(b'0')
    with pytest.raises(ValueError):
        utils.load_clip(str(tmp_path / 'a.mp4'))
    np.save(tmp_path / 'c.npy', np.zeros((2, 8, 8, 3), np.uint8))
    assert utils.load_clip(str(tmp_path / 'c.npy')).shape == (2, 8, 8, 3)
    assert utils.get_videos({'data': {'videos': str(tmp_path)}}) == [str(tmp_path / 'a.mp4'), str(tmp_path / 'c.npy')]


# ------------------------------------------------------------------ training, host side
def test_keras_adam_restatement_first_steps():
    """Keras 2.2.4 Adam: the first update is lr * g / (|g| + eps * sqrt(1 - beta_2)) ~ lr * sign(g); with a
    constant gradient every later update is lr as well (bias corrections cancel)."""
    from oracle import ref_train
    opt = ref_train.KerasAdam(lr=1e-3)
    p = {"w": np.array([1.0, -2.0, 0.5], np.float32)}
    g = {"w": np.array([0.3, -4.0, 1e-3], np.float64)}
    p1 = opt.step(p, g)
    assert np.allclose(p["w"] - p1["w"], 1e-3 * np.sign(g["w"]), rtol=5e-3)     # epsilon shows at |g| = 1e-3: 0.3 %
    assert np.allclose((p["w"] - p1["w"])[:2], 1e-3 * np.sign(g["w"])[:2], rtol=1e-4)
    p2 = opt.step(p1, g)
    assert np.allclose(p1["w"] - p2["w"], 1e-3 * np.sign(g["w"]), rtol=5e-3)
    # hand-evaluated second step with a changed gradient: m = .9*.1*g1 + .1*g2, v = .999*.001*g1^2 + .001*g2^2
    opt = ref_train.KerasAdam(lr=1e-3)
    q = opt.step({"w": np.zeros(1, np.float32)}, {"w": np.array([1.0])})
    q = opt.step(q, {"w": np.array([-1.0])})
    m, v = 0.9 * 0.1 - 0.1, 0.999 * 0.001 + 0.001
    lr_t = 1e-3 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    assert np.allclose(q["w"], -1e-3 - lr_t * m / (np.sqrt(v) + 1e-7), rtol=1e-5)


def test_oracle_loss_is_the_keras_mean_over_all_elements():
    from oracle import ref_train
    from oracle import ref_torch as rt
    w = rn.make_weights(seed=7)
    rng = np.random.default_rng(3)
    a, b = rng.random((2, 8, 8, 3), dtype=np.float32), rng.random((2, 8, 8, 3), dtype=np.float32)
    c, hr = rng.random((2, 32, 32, 3), dtype=np.float32), rng.random((2, 32, 32, 3), dtype=np.float32)
    losses, grads, (est, lw) = ref_train.loss_and_grads(w, a, b, c, hr)
    assert abs(losses[1] - np.mean((est - hr.astype(np.float64)) ** 2)) < 1e-12
    assert abs(losses[2] - np.mean((lw - a.astype(np.float64)) ** 2)) < 1e-12
    assert abs(losses[0] - losses[1] - losses[2]) < 1e-15
    assert list(grads) == list(w) and all(grads[n].shape == w[n].shape for n in w)
    # finite-difference check of one weight through both losses (FNet's first bias feeds both warps)
    n = "FNet/conv2d_1/bias:0"
    eps = 1e-6
    wp, wm = dict(w), dict(w)
    wp[n] = w[n].astype(np.float64); wp[n][3] += eps
    wm[n] = w[n].astype(np.float64); wm[n][3] -= eps
    lp = ref_train.loss_and_grads(wp, a, b, c, hr)[0][0]
    lm = ref_train.loss_and_grads(wm, a, b, c, hr)[0][0]
    assert abs((lp - lm) / (2 * eps) - grads[n][3]) <= 1e-6 * max(1.0, abs(grads[n][3]) * 1e3)


def test_get_frames_and_compile_surface(tmp_path):
    import utils
    from layers import Adam, GeneratorCore, GeneratorModel
    clip = (np.arange(9 * 40 * 48 * 3) % 251).astype(np.uint8).reshape(9, 40, 48, 3)
    cfg = {"train": {"c_frames": 4}, "data": {"high_res": 32}}
    fr = utils.get_frames(clip, cfg, np.random.RandomState(0))
    assert fr.shape == (4, 32, 32, 3)
    found = [(t, y, x) for t in range(6) for y in range(9) for x in range(17)
             if np.array_equal(fr, clip[t:t + 4, y:y + 32, x:x + 32])]
    assert len(found) >= 1                                   # 4 consecutive frames, one crop position for all
    with pytest.raises(ValueError):
        utils.get_frames(clip[:3], cfg)
    m = GeneratorModel(GeneratorCore((16, 16, 3)), "FRVSR")
    with pytest.raises(RuntimeError):
        m.train_on_batch([np.zeros((1, 16, 16, 3))] * 2 + [np.zeros((1, 64, 64, 3))], [np.zeros((1, 64, 64, 3)), np.zeros((1, 16, 16, 3))])
    with pytest.raises(NotImplementedError):
        m.compile(loss=["mean_squared_error", "mean_absolute_error"], optimizer=Adam(1e-4))
    with pytest.raises(NotImplementedError):
        m.compile(loss=["mean_squared_error"] * 2, loss_weights=[1., 0.5], optimizer=Adam(1e-4))
    m.compile(loss=["mean_squared_error"] * 2, loss_weights=[1., 1.], optimizer=Adam(lr=1e-4))
    assert m.optimizer.lr == 1e-4 and m.optimizer.epsilon == 1e-7
    with pytest.raises(NotImplementedError):                 # second target must be I_LR_t
        m.train_on_batch([np.zeros((1, 16, 16, 3))] * 2 + [np.zeros((1, 64, 64, 3))], [np.zeros((1, 64, 64, 3)), np.ones((1, 16, 16, 3))])


def test_product_synth_data_is_the_oracles_generator():
    """super-resolution_amd/synth.py (what bench.py and the demo configs feed the generator) and the oracle's own
    generator are separate code on purpose (the product imports nothing from oracle/); they must stay bit-identical
    so that bench.py, the parity tests and the CPU baseline see the same clips and weights."""
    import synth
    from oracle import ref_numpy as rn
    np.testing.assert_array_equal(synth.synth_clip(4, 3, 20, 36), rn.synth_clip(4, 3, 20, 36))
    np.testing.assert_array_equal(synth.synth_clip(1, 2, 16, 24, scale=4), rn.synth_clip(1, 2, 16, 24, scale=4))
    a, b = synth.make_weights(7), rn.make_weights(7)
    assert list(a) == list(b)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run starts 2 ranks itself (--gpus was
    ignored); --launch-check makes them rendezvous over gloo without a GPU.  A launcher-provided WORLD_SIZE that
    disagrees with --gpus is an error, not a silent 1-GPU run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["world_size_env"] == 2
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--launch-check"],
                       env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_traffic_comes_from_the_newest_profile_and_never_from_another_kernel(tmp_path):
    """bench.py's roofline.traffic silently fell back to round 1's file (another kernel).  The lookup
    takes the newest profiles/rN_hbm_traffic.json only, matches the timed kernels by name prefix, and reports None with a
    note when they are not in it."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    # the committed files: the newest one must resolve for the default dtype, and be the highest round on file
    rounds = sorted(int(f[1:f.index("_")]) for f in os.listdir(os.path.join(root, "profiles")) if f.endswith("_hbm_traffic.json") and f[1:f.index("_")].isdigit())
    for dtype in ("f16mx6", "bf16x3"):                  # the bench's default and the one it reports beside it
        val, src, note = bench.pmc_traffic(dtype)
        assert src == f"r{rounds[-1]}_hbm_traffic.json", (src, rounds)
        assert val and 4e8 < val < 9e8, (dtype, val, note)          # 482 / 723 MB algorithmic on the two instantiations
        d = json.load(open(os.path.join(root, "profiles", src)))
        assert "trunk_kernel" in d and d["trunk_kernel"][dtype] == bench.TRUNK_KERNELS[dtype]
    # synthetic directory: r9 (without the kernels) shadows r3 (with them) -> None + note, not r3's number
    k = {"launches": 5, "fetch_bytes": 1, "write_bytes": 2, "hbm_bytes": 700}
    json.dump({"kernels": {"trunk_conv_kernel<false, true, 0, false> @grid1": k, "trunk_conv_kernel<true, false, 0, false> @grid1": dict(k, hbm_bytes=500)}},
              open(tmp_path / "r3_hbm_traffic.json", "w"))
    assert bench.pmc_traffic("bf16x3", str(tmp_path))[:2] == (600, "r3_hbm_traffic.json")
    json.dump({"kernels": {"some_other_kernel @grid1": k}}, open(tmp_path / "r9_hbm_traffic.json", "w"))
    val, src, note = bench.pmc_traffic("bf16x3", str(tmp_path))
    assert val is None and src == "r9_hbm_traffic.json" and "trunk_conv_kernel" in note
    assert bench.pmc_traffic("bf16x3", str(tmp_path / "nothing"))[0] is None
