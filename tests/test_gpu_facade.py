"""The drop-in Python API on the GPU: reference-style predict calls, sub-models, eval driver."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as rn
from oracle import ref_torch as rt

pytestmark = pytest.mark.gpu


def build(kind, lr, tmp, weights, rows=1, cols=1, dtype='fp32'):
    from FRVSR import FrameRecurrentVideoSR
    from TecoGAN import TecoGAN
    cfg = {'root_dir': str(tmp), 'dtype': dtype, 'data': {'rows': rows, 'cols': cols, 'videos': str(tmp / 'clips')},
           'cnn': {'model': kind, 'pretrained_model': 'w.npz'}, 'eval': {'output_dir': str(tmp / 'out')}}
    import weights_io
    weights_io.write_tensors(str(tmp / 'w.npz'), list(weights.items()))
    cls = FrameRecurrentVideoSR if kind == 'frvsr' else TecoGAN
    m = cls(lr, (4 * lr[0], 4 * lr[1], 3), str(tmp), cfg)
    m.build()
    return m


def test_reference_style_predict_calls(tmp_path, weights7):
    m = build('frvsr', (16, 24, 3), tmp_path, weights7)
    m.frvsr.load_weights(str(tmp_path / 'w.npz'))
    lr_t = rn.synth_clip(1, 1, 16, 24)
    lr_tm1 = rn.synth_clip(2, 1, 16, 24)
    est_prev = np.zeros((1, 64, 96, 3))                      # float64 zeros like FRVSR.py:233-234
    est, lr_warp = m.frvsr.predict([lr_t, lr_tm1, est_prev])
    want, want_lw = rt.generator_step(weights7, lr_t, lr_tm1, est_prev)
    assert est.dtype == np.float32 and np.abs(est - want).max() <= 1e-3
    assert np.abs(lr_warp - want_lw).max() <= 1e-3
    # sub-models reproduce the graph: fnet -> upscaling -> warp -> space2depth -> srnet (FRVSR.py:58-63)
    flow = m.fnet.predict([lr_t, lr_tm1])
    f_hr = m.upscaling.predict(flow)
    warped = m.hr_warp.predict([est.astype(np.float32), f_hr])
    s_s = m.space2depth.predict(warped)
    est2 = m.srnet.predict([lr_t, s_s])
    est2_ref, _ = m.frvsr.predict([lr_t, lr_tm1, est])
    assert np.abs(est2 - est2_ref).max() <= 1e-4
    lw = m.lr_warp.predict([lr_tm1, flow])
    assert np.abs(lw - lr_warp).max() <= 1e-5


def test_tecogan_generator_and_hr_warping(tmp_path, weights7):
    t = build('tecogan', (16, 16, 3), tmp_path, weights7)
    t.generator.load_weights(str(tmp_path / 'w.npz'))
    lr_t, lr_tm1 = rn.synth_clip(3, 1, 16, 16), rn.synth_clip(4, 1, 16, 16)
    est_prev = rn.uniform01(5, 64 * 64 * 3).reshape(1, 64, 64, 3).astype(np.float32)
    est, _ = t.generator.predict([lr_t, lr_tm1, est_prev])
    want, _ = rt.generator_step(weights7, lr_t, lr_tm1, est_prev, model='tecogan')
    assert np.abs(est - want).max() <= 1e-3
    taps = {}
    rt.generator_step(weights7, lr_t, lr_tm1, est_prev, taps=taps)
    hw = t.hr_warping.predict([lr_t, lr_tm1, est_prev])      # single array, not a list (TecoGAN.py:83-84)
    assert isinstance(hw, np.ndarray) and np.abs(hw - taps['hr_warp']).max() <= 2e-4


def test_eval_driver_tiles_state_and_metrics(tmp_path, weights7):
    """2x2 tiles of 16x16 LR, 11 frames: tiling (no halo), per-tile recurrence, metrics.txt columns."""
    m = build('frvsr', (16, 16, 3), tmp_path, weights7, rows=2, cols=2)
    os.makedirs(tmp_path / 'clips')
    hr = rn.synth_clip(7, 11, 128, 128, scale=4)
    np.save(tmp_path / 'clips' / 'a.npy', (hr * 255).astype(np.uint8))
    mean_psnr = m.eval()
    text = open(tmp_path / 'out' / 'metrics.txt').read()
    assert text.count('Video:\ta.npy\tFrame:\t') == 2 and 'Inference time:' in text and 'Total PSNR:' in text
    est = np.load(tmp_path / 'out' / 'a.npy')
    assert est.shape == (11, 128, 128, 3) and est.dtype == np.uint8
    # oracle for one tile (row 1, col 0), full recurrence
    import utils
    hr_n = utils.normalize((hr * 255).astype(np.uint8))
    lr = utils.box_downscale(hr_n, 4)
    tile = lr[:, 16:32, 0:16][:, None]
    want = rt.run_clip(weights7, tile)
    got = est[:, 64:128, 0:64].astype(np.float32)
    assert np.abs(got - np.floor(want[:, 0] * 255.0)).max() <= 1.0      # uint8 truncation of values within 1e-3
    frames_psnr = [utils.psnr(hr_n[t], est[t].astype(np.float32) / 255.0) for t in (0, 10)]
    assert abs(mean_psnr - np.mean(frames_psnr)) < 0.05


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_eval_lock_step_equals_one_clip_at_a_time(tmp_path, weights7, dtype):
    """stage.eval advances eval.clips_in_flight clips in lock-step through the uint8 output path
    (sr_clip_push_u8 + page-locked double buffers, float read-back on the metric frames).  Three clips of unequal length
    (12, 7, 21 frames; 1x2 tiles of 16x24) in one group give, bit for bit, the frames and the metrics of the same clips
    evaluated one at a time, and of the float path (super_resolve_clip + denormalize)."""
    import re
    import utils
    m = build('frvsr', (16, 24, 3), tmp_path, weights7, rows=1, cols=2, dtype=dtype)
    os.makedirs(tmp_path / 'clips')
    lengths = {'a.npy': 12, 'b.npy': 7, 'c.npy': 21}
    for k, (name, T) in enumerate(lengths.items()):
        np.save(tmp_path / 'clips' / name, (rn.synth_clip(30 + k, T, 64, 192, scale=4) * 255).astype(np.uint8))

    def run(n_fly, out, **extra):
        m.configs['eval'] = dict({'output_dir': str(tmp_path / out), 'clips_in_flight': n_fly}, **extra)
        mean = m.eval()
        text = open(tmp_path / out / 'metrics.txt').read()
        rows = sorted(re.sub(r'Inference time:\t\d+', 'Inference time:\tX', l) for l in text.splitlines() if l.startswith('Video:'))
        return mean, rows, text.splitlines()[-1], {n: np.load(tmp_path / out / n) for n in lengths}

    mean3, rows3, total3, est3 = run(3, 'lock')
    mean1, rows1, total1, est1 = run(1, 'single')
    assert len(rows3) == 2 + 1 + 3 and rows3 == rows1 and total3 == total1 and mean3 == mean1
    # two page-locked slots instead of one per frame (what a group too large for eval.pinned_limit_mb gets)
    mean_r, rows_r, total_r, est_r = run(3, 'ring', pinned_limit_mb=0)
    assert rows_r == rows3 and total_r == total3
    for n in lengths:
        np.testing.assert_array_equal(est_r[n], est3[n])
    assert m.eval_fps > 0
    for n, T in lengths.items():
        assert est3[n].shape == (T, 64, 192, 3) and est3[n].dtype == np.uint8
        np.testing.assert_array_equal(est3[n], est1[n])
    # the float path of the same driver on clip c: same frames after the reference's denormalize, same metric values
    hr = utils.normalize(np.load(tmp_path / 'clips' / 'c.npy'))
    import sr_hip
    lr = sr_hip.blur_cubic_downscale(hr)
    est_f, stats = m.super_resolve_clip(lr, hr, 'c.npy')
    np.testing.assert_array_equal(utils.denormalize(est_f), est3['c.npy'])
    got = [float(re.search(r'PSNR:\t([0-9.]+)', l).group(1)) for l in rows3 if 'c.npy' in l]
    assert np.allclose(sorted(got), sorted(stats['psnr']), atol=1e-6)


def test_train_stage_runs_the_reference_loop(tmp_path, weights7):
    """stage.train through the facade (reference FRVSR.py:76-202): batches of batch_size clips x c_frames
    frames, previous estimates from the generator itself, train_on_batch, checkpoints that load back.  The
    first batch is checked against the oracle's train_on_batch; training then lowers the loss on a fixed clip."""
    from oracle import ref_train
    import utils
    import weights_io
    m = build('frvsr', (16, 16, 3), tmp_path, weights7)
    os.makedirs(tmp_path / 'clips')
    for k in range(2):
        np.save(tmp_path / 'clips' / f'{k}.npy', (rn.synth_clip(20 + k, 8, 96, 96, scale=4) * 255).astype(np.uint8))
    m.configs['train'] = {'iterations': 0, 'batch_size': 2, 'c_frames': 3, 'sample_freq': 1, 'checkpoint_freq': 1,
                          'info_freq': 1, 'seed': 1}
    m.configs['data'].update(high_res=64)
    # one batch by hand: the facade's inputs are what the oracle's generator produces for the same clips
    hr = np.stack([utils.normalize(utils.get_frames(str(tmp_path / 'clips' / f'{k}.npy'), m.configs, np.random.RandomState(k)))
                   for k in range(2)])
    m._load_pretrained()
    x, y = m.make_train_batch(hr)
    assert [a.shape for a in x] == [(6, 16, 16, 3), (6, 16, 16, 3), (6, 64, 64, 3)] and y[1] is x[0]
    lr = utils.blur_cubic_downscale(hr[1])
    want = rt.run_clip(weights7, lr[:, None])                                 # clip 1 through the oracle
    assert not x[1][3].any() and not x[2][3].any()                           # frame 0: zero state (FRVSR.py:124-125)
    assert np.abs(x[1][4] - lr[0]).max() <= 2e-6 and np.abs(x[2][5] - want[1, 0]).max() <= 1e-3   # LR synthesis runs on the GPU
    new_w, want_l = ref_train.train_on_batch(dict(weights7), ref_train.KerasAdam(1e-4), x[0], x[1], x[2], y[0])
    loss = m.frvsr.train_on_batch(x, y)
    assert np.allclose(loss, want_l, rtol=1e-4)
    got = dict(zip(weights7.keys(), m.frvsr.get_weights()))                  # get_weights sees the trained values
    d = np.concatenate([np.abs(got[n] - new_w[n]).ravel() for n in new_w])
    assert d.mean() <= 0.02 * 1e-4 and np.abs(got['SRNet/conv2d_36/bias:0'] - weights7['SRNet/conv2d_36/bias:0']).max() > 5e-5
    first = loss[0]
    # the stage itself: 1 iteration (i = 0), checkpoint + sample + yaml written (rank 0)
    m.train()
    ck = tmp_path / 'Checkpoints' / 'frvsr_model_weights_0.h5'
    assert ck.exists() and (tmp_path / 'Samples' / 'Sample_0.npy').exists() and (tmp_path / 'FRVSR.yaml').exists()
    assert np.load(tmp_path / 'Samples' / 'Sample_0.npy').shape == (192, 64, 3)
    saved = weights_io.read_tensors(str(ck))
    for (n, a), b in zip(saved, m.frvsr.get_weights()):
        np.testing.assert_array_equal(a, b)
    # 40 more steps on the same batch: the loss goes down (Adam at 1e-4 on 6 samples)
    for _ in range(40):
        loss = m.frvsr.train_on_batch(x, y)
    assert loss[0] < 0.9 * first


def test_tecogan_generator_pretraining_stage(tmp_path, weights7):
    """TecoGAN.train(): the generator pre-training phase (reference TecoGAN.py:164-275) runs natively with the halving
    learning-rate schedule and writes generator checkpoints; the adversarial phase then starts at iteration 0 and 5e-5."""
    from oracle import ref_train
    from TecoGAN import TecoGAN
    assert [TecoGAN.gen_learning_rate(i) for i in (0, 49999, 50000, 100000, 10 ** 7)] == [1e-4, 1e-4, 5e-5, 2.5e-5, 2.5e-5]
    t = build('tecogan', (16, 16, 3), tmp_path, weights7)
    os.makedirs(tmp_path / 'clips')
    np.save(tmp_path / 'clips' / '0.npy', (rn.synth_clip(31, 6, 80, 80, scale=4) * 255).astype(np.uint8))
    t.configs['train'] = {'iterations': 4, 'batch_size': 1, 'c_frames': 3, 'sample_freq': 1, 'checkpoint_freq': 1,
                          'info_freq': 1, 'seed': 3, 'pretrained_disc': None}
    t.configs['data'].update(high_res=64)
    assert t.generator.optimizer.lr == 5e-5                                   # TecoGAN.py:110
    assert t.generator.optimizer is t.discriminator.optimizer is t.tecogan.optimizer   # ONE Adam, TecoGAN.py:111-131
    before = [a.copy() for a in weights7.values()]                            # train() loads cnn.pretrained_model first

    class EndOfPretraining(Exception):
        pass

    seen = {}

    def stop(hr):
        seen['lr'], seen['shape'] = t.generator.optimizer.lr, hr.shape
        raise EndOfPretraining()

    t.gan_iteration = stop
    with pytest.raises(EndOfPretraining):
        t.train()                                                            # iterations 0, 1 (< iterations / 2), then the GAN loop
    assert seen == {'lr': 5e-5, 'shape': (1, 3, 64, 64, 3)}                   # TecoGAN.py:278
    assert t.opt.iterations == 2                                              # the shared counter saw the two generator steps
    assert (tmp_path / 'Checkpoints' / 'generator_model_weights_1.h5').exists() and (tmp_path / 'TecoGAN.yaml').exists()
    after = t.generator.get_weights()
    moved = max(np.abs(a - b).max() for a, b in zip(after, before))
    assert 1e-4 <= moved <= 2.1e-4                                            # two Adam steps at 1e-4
    # one more step on a hand-made batch against the oracle's tecogan step from the same weights
    import utils
    hr = utils.normalize(np.load(tmp_path / 'clips' / '0.npy')[None, :3, :64, :64])
    x, y = t.make_train_batch(hr)
    w_now = dict(zip(weights7.keys(), t.generator.get_weights()))
    want_l, _, _ = ref_train.loss_and_grads(w_now, x[0], x[1], x[2], y[0], model='tecogan')
    loss = t.generator.train_on_batch(x, y)
    assert np.allclose(loss, want_l, rtol=1e-4)


def test_tecogan_adversarial_stage(tmp_path, weights7):
    """The adversarial phase (reference TecoGAN.py:277-527): the ping-pong batch against the oracle's recurrence and warps,
    one whole iteration (2 discriminator steps with fresh moments, predict, 2 combined steps, one shared Adam counter)
    against the oracle's restatement, then train() itself with a pretrained discriminator: checkpoints of both models."""
    from oracle import ref_disc as rd
    import utils
    import weights_io
    t = build('tecogan', (16, 16, 3), tmp_path, weights7)
    t._load_pretrained()
    dw = rd.make_disc_weights(9, 64, 64)
    t.discriminator.set_weights(list(dw.values()))
    assert [n for n, _ in t.discriminator.named_weights()] == list(dw.keys())
    hr = rn.synth_clip(41, 4, 64, 64, scale=4)[None]                          # one clip of c = 4 frames
    b = t.make_gan_batch(hr)
    assert all(a.shape[0] == 4 for a in b.values()) and len(b) == 16          # 2 groups x (c - 2) frames
    assert np.array_equal(b['hr'], b['hr'][::-1])                             # ping-pong: the batch is its own mirror image
    assert np.array_equal(b['hr'][0], hr[0, 1]) and np.array_equal(b['next_hr'][1], hr[0, 3]) and np.array_equal(b['prev_hr'][2], hr[0, 3])
    lr = utils.blur_cubic_downscale(hr[0])
    assert np.abs(b['lr'][1] - lr[2]).max() <= 2e-6 and np.abs(b['next_lr'][3] - lr[0]).max() <= 2e-6
    assert np.abs(b['bic'][0] - utils.cubic_upscale4(b['lr'][0:1])[0]).max() <= 1e-6
    fwd = rt.run_clip(weights7, lr[:, None], model='tecogan')                # forward group through the oracle
    bwd = rt.run_clip(weights7, lr[::-1][:, None], model='tecogan')          # the reversed group
    assert np.abs(b['est'][0] - fwd[1, 0]).max() <= 1e-3 and np.abs(b['prev_est'][1] - fwd[1, 0]).max() <= 1e-3
    assert np.abs(b['est'][2] - bwd[1, 0]).max() <= 1e-3 and np.abs(b['next_est'][3] - bwd[3, 0]).max() <= 1e-3
    taps = {}
    rt.generator_step(weights7, b['lr'][1:2], b['next_lr'][1:2], b['next_est'][1:2], model='tecogan', taps=taps)
    assert np.abs(b['next_hat_est'][1] - taps['hr_warp'][0]).max() <= 2e-4
    rt.generator_step(weights7, b['lr'][2:3], b['prev_lr'][2:3], b['prev_hr'][2:3], model='tecogan', taps=taps)
    assert np.abs(b['prev_hat_hr'][2] - taps['hr_warp'][0]).max() <= 2e-4
    # one iteration against the oracle, from the same batch arrays
    t.make_gan_batch = lambda _hr: b
    t.step_by_step = True                       # the five model calls of the reference; the fused native pass is compared below
    moments = ({}, {}, {}, {})
    gen_w, disc_w, its, want_l = rd.gan_iteration(dict(weights7), dict(dw), b, 0, moments)
    loss = t.gan_iteration(hr)
    assert t.opt.iterations == its == 4
    assert np.allclose(loss, want_l, rtol=2e-3, atol=1e-6), (loss, want_l)
    step = 5e-5 * 4
    got_g = dict(zip(weights7.keys(), t.generator.get_weights()))
    got_d = dict(t.discriminator.named_weights())
    for got, want, w0 in ((got_g, gen_w, weights7), (got_d, disc_w, dw)):
        for n, a in want.items():
            if n.startswith('conv2d_') and n.endswith('bias:0') and not n.startswith('conv2d_37'):
                continue                                                      # zero gradient (BatchNormalization follows): Adam steps on rounding noise
            d = np.abs(got[n] - a)
            if 'moving' in n:
                assert d.max() <= 1e-5, n
                continue
            assert np.abs(a - w0[n]).max() > 0, n
            assert d.mean() <= 0.05 * step and d.max() <= 2.01 * step, (n, d.mean(), d.max())
    # the same pass as ONE native call (sr_tecogan_iteration) from the same start: the same weights, bit for bit
    t2 = build('tecogan', (16, 16, 3), tmp_path, weights7)
    t2._load_pretrained()
    t2.discriminator.set_weights(list(dw.values()))
    t2.make_gan_batch = lambda _hr: b
    loss2 = t2.gan_iteration(hr)
    assert t2.opt.iterations == 4 and np.allclose(loss2, loss, rtol=1e-6, atol=1e-9)
    for x, y in zip(t2.generator.get_weights() + t2.discriminator.get_weights(), t.generator.get_weights() + t.discriminator.get_weights()):
        np.testing.assert_array_equal(x, y)
    # the stage: pretrained discriminator => no pre-training (TecoGAN.py:148-152), 2 iterations, both checkpoints
    t.step_by_step = False
    del t.make_gan_batch
    os.makedirs(tmp_path / 'clips')
    np.save(tmp_path / 'clips' / '0.npy', (rn.synth_clip(31, 6, 80, 80, scale=4) * 255).astype(np.uint8))
    weights_io.write_tensors(str(tmp_path / 'd.h5'), list(dw.items()), keep_layer_prefix=True)
    t.configs['train'] = {'iterations': 2, 'batch_size': 2, 'c_frames': 4, 'sample_freq': 1, 'checkpoint_freq': 1,
                          'info_freq': 1, 'seed': 3, 'pretrained_disc': 'd.h5'}
    t.configs['data'].update(high_res=64)
    loss = t.train()
    assert len(loss) == 13 and all(np.isfinite(loss)) and loss[7:11] == [0.0] * 4     # the VGG terms count as 0
    assert t.opt.iterations == 4 + 2 * 4
    ck = tmp_path / 'Checkpoints'
    assert (ck / 'generator_model_weights_1.h5').exists() and (ck / 'discriminator_model_weights_1.h5').exists()
    assert t.configs['train']['pretrained_disc'].endswith('discriminator_model_weights_1.h5')
    saved = weights_io.read_tensors(str(ck / 'discriminator_model_weights_1.h5'))
    assert [n for n, _ in saved] == list(dw.keys())
    for (n, a), (_, cur) in zip(saved, t.discriminator.named_weights()):
        np.testing.assert_array_equal(a, cur)
    assert (tmp_path / 'Samples' / 'Sample_1.npy').exists()


def test_tecogan_perceptual_terms_from_a_weight_file(tmp_path, weights7):
    """train.vgg19_weights (SURVEY.md section 8 f rank 3, first branch): the 32 VGG19 tensors come from a file -- here seeded
    stand-ins, the ImageNet weights are a download -- and the adversarial iteration, both as one native call and through the
    five Keras-shaped model calls, reports the four perceptual terms; vgg_22 .. vgg_54 are callable like the reference's."""
    from oracle import ref_disc as rd
    from oracle import ref_vgg as rv
    import weights_io
    vggw = rv.make_vgg_weights(3)
    weights_io.write_tensors(str(tmp_path / 'vgg19.npz'), list(vggw.items()))
    from TecoGAN import TecoGAN
    cfg = {'root_dir': str(tmp_path), 'dtype': 'fp32', 'data': {'rows': 1, 'cols': 1, 'videos': str(tmp_path / 'clips')},
           'cnn': {'model': 'tecogan', 'pretrained_model': 'w.npz'}, 'eval': {'output_dir': str(tmp_path / 'out')},
           'train': {'vgg19_weights': 'vgg19.npz'}}
    weights_io.write_tensors(str(tmp_path / 'w.npz'), list(weights7.items()))
    losses = {}
    for mode in ('native', 'keras'):
        t = TecoGAN((16, 16, 3), (64, 64, 3), str(tmp_path), cfg)
        t.build()
        t._load_pretrained()
        assert len(t.tecogan.vgg_weights) == 32
        t.discriminator.set_weights(list(rd.make_disc_weights(9, 64, 64).values()))     # a fresh model draws random weights
        t.step_by_step = mode == 'keras'
        hr = rn.synth_clip(41, 4, 64, 64, scale=4)[None]
        losses[mode] = t.gan_iteration(hr)
        b = t._last_gan_batch
    assert min(losses['native'][7:11]) > 0
    assert np.allclose(losses['native'], losses['keras'], rtol=1e-5, atol=1e-9)
    # the loss functions themselves, on the last batch's (hr, est): the oracle's values
    want, _ = rv.vgg_loss_and_grad(vggw, b['hr'], b['est'])
    got = [t.vgg_22(b['hr'], b['est']), t.vgg_34(b['hr'], b['est']), t.vgg_44(b['hr'], b['est']), t.vgg_54(b['hr'], b['est'])]
    assert np.allclose(got, want, rtol=1e-4)
    # without the file: 0, as before
    cfg2 = dict(cfg, train={})
    t0 = TecoGAN((16, 16, 3), (64, 64, 3), str(tmp_path), cfg2)
    t0.build()
    t0._load_pretrained()
    assert t0.gan_iteration(hr)[7:11] == [0.0] * 4
