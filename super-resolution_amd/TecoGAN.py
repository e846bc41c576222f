"""Temporally Coherent GAN on MI355X: drop-in for reference TecoGAN.py.

``generator`` = the FRVSR graph plus ``tanh(SRNet(...) + bilinear_x4(I_LR_t))`` (TecoGAN.py:65-80;
the reference's ``bicubic_upscale`` is in fact bilinear, layers.py:89), ``hr_warping`` returning the
warped previous estimate for the same three inputs (TecoGAN.py:83-85), ``discriminator`` (layers.py:142-186)
and ``tecogan``, the combined model of TecoGAN.py:87-131.  ``train()`` runs both phases of the reference:
generator pre-training (TecoGAN.py:164-275: two MSE losses, learning rate halved every 50k iterations down
to 2.5e-5) and the adversarial phase (TecoGAN.py:277-527: ping-pong batch, two discriminator steps, two
combined steps per iteration), every step on the native library.

The four VGG19 perceptual terms (TecoGAN.py:125,129,867-901) need the ImageNet weights that ``keras.applications.VGG19``
downloads; there is no network here, so they come from a file: ``train.vgg19_weights: path`` (.npz / .h5 with the 32
tensors of ``VGG19(include_top=False)`` in Keras order, e.g. written once by ``model.save_weights`` on a machine that has
them).  With the file the combined step evaluates all twelve terms (native VGG19 forward / input gradient, sr_vgg_*);
without it those four count as 0 and are reported as 0 (the other eight terms are exact either way).
"""
import os

import numpy as np
import yaml

import dp
import sr_hip
from FRVSR import FrameRecurrentVideoSR
from layers import (TECOGAN_LOSS_WEIGHTS, Adam, Discriminator, GeneratorModel, HRWarpingModel, TecoGANModel, Upscaling)
from utils import denormalize, get_frames, get_videos, load_clip, normalize, psnr


class TecoGAN(FrameRecurrentVideoSR):
    model_kind = 'tecogan'
    step_by_step = False      # True: the adversarial pass goes through the five model calls of the reference instead of one native call

    def build(self):
        super().build()
        device = self.core.device
        self.bicubic_upscale = Upscaling(self.lr_shape, 1, device=device).build()
        print('[INFO] Creating Discriminator ...')
        self.discriminator = Discriminator(self.hr_shape, dtype=self.core.dtype, device=device).build()
        print('[INFO] Discriminator ready.')
        print('[INFO] Creating Temporally Coherent GAN ...')
        self.generator = GeneratorModel(self.core, 'Generator')
        self.hr_warping = HRWarpingModel(self.core)
        train_cfg = (self.configs.get('train') or {}) if isinstance(self.configs, dict) else {}
        self.tecogan = TecoGANModel(self.generator, self.discriminator,
                                    train_discriminator=train_cfg.get('gan_updates_discriminator', True))
        self.opt = Adam(lr=5e-5)                                    # ONE instance for the three models, TecoGAN.py:110-131
        self.generator.compile(loss=['mean_squared_error', 'mean_squared_error'], loss_weights=[1., 1.], optimizer=self.opt)
        self.discriminator.compile(loss=['binary_crossentropy'] + ['mean_squared_error'] * 4,
                                   loss_weights=[1., 0., 0., 0., 0.], optimizer=self.opt)
        vgg_path = train_cfg.get('vgg19_weights')
        if vgg_path:
            import weights_io
            path = vgg_path if os.path.isabs(vgg_path) else os.path.join(self.configs.get('root_dir', './'), vgg_path)
            print('[INFO] Loading VGG19 weights for the perceptual losses ...')
            self.tecogan.set_vgg_weights([a for _, a in weights_io.read_tensors(path)])
        self.tecogan.compile(loss=['mean_squared_error', 'binary_crossentropy'] + ['mean_squared_error'] * 4
                             + [self.vgg_22, self.vgg_34, self.vgg_44, self.vgg_54] + ['mean_squared_error'] * 2,
                             loss_weights=list(TECOGAN_LOSS_WEIGHTS), optimizer=self.opt)
        del self.frvsr
        print('[INFO] Temporally Coherent GAN ready.')

    # the reference's perceptual losses (TecoGAN.py:867-901) as stand-alone functions of two HR batches (the combined step
    # evaluates them on the device together with their gradient; these are for inspection and tests)
    def _vgg(self, k, y_true, y_pred):
        if getattr(getattr(self, 'tecogan', None), 'vgg_weights', None) is None:
            raise NotImplementedError('VGG19 perceptual losses need the ImageNet weights (a download): give them as '
                                      'train.vgg19_weights (32 tensors in Keras order); without them the terms count as 0')
        y_true = np.asarray(y_true, np.float32)
        v = sr_hip.Vgg19(self.hr_shape[0], self.hr_shape[1], max_batch=len(y_true), dtype=self.core.dtype, device=self.core.device)
        try:
            v.load_weights(self.tecogan.vgg_weights)
            return v.loss_grads(y_true, y_pred, want_grad=False)[0][k]
        finally:
            v.close()

    def vgg_22(self, y_true, y_pred):
        return self._vgg(0, y_true, y_pred)

    def vgg_34(self, y_true, y_pred):
        return self._vgg(1, y_true, y_pred)

    def vgg_44(self, y_true, y_pred):
        return self._vgg(2, y_true, y_pred)

    def vgg_54(self, y_true, y_pred):
        return self._vgg(3, y_true, y_pred)

    def _generator(self):
        return self.generator

    @staticmethod
    def gen_learning_rate(i):
        """1e-4 halved every 50k iterations, floor 2.5e-5 (TecoGAN.py:173-175, 225-228)."""
        return max(1e-4 / 2 ** (i // 5e4), 2.5e-5)

    # ------------------------------------------------------------------ adversarial batch (TecoGAN.py:296-432)
    def make_gan_batch(self, hr_clips):
        """hr_clips [n, c, H, W, 3] float32 in [0,1] -> dict of the fifteen arrays of one adversarial iteration, each
        [2n(c-2), ...] in the reference's sample order (group-major: the n clips forward, then the same frames backward --
        ``np.append(hr_batch, hr_batch[::-1])``, TecoGAN.py:305 -- first and last frame of every group of c dropped,
        TecoGAN.py:416-432).  Per kept sample b: LR = blur + cubic downscale, bic = cubic x4 of it, est = the generator's
        own output along the group (zero state at the group's first frame), *_hat_* = hr_warping of the neighbour frame's
        estimate / ground truth towards frame b.  The reference runs one predict per sample; here the 2n groups advance
        in lock step (same numbers: the groups are independent)."""
        n, c = hr_clips.shape[:2]
        if c < 3:
            raise ValueError('the adversarial phase needs c_frames >= 3 (previous, present and next frame)')
        H, W, _ = self.hr_shape
        flat = hr_clips.reshape((n * c, H, W, 3)).astype(np.float32)
        hr = np.concatenate([flat, flat[::-1]], axis=0).reshape((2 * n, c, H, W, 3))
        G = 2 * n
        lr = sr_hip.blur_cubic_downscale(hr.reshape((G * c, H, W, 3)), device=self.core.device)
        bic = sr_hip.cubic_upscale4(lr, device=self.core.device).reshape((G, c, H, W, 3))   # utils.cubic_upscale4 in numpy: 30 ms
        lr = lr.reshape((G, c) + lr.shape[1:])
        gen = self.generator
        self.core.ctx(G * (c - 2))                 # size the native context for the training batch before the recurrence starts
        est = np.empty_like(hr)
        gen.clip_begin(G)
        for f in range(c):
            est[:, f] = gen.clip_push(lr[:, f])
        mid = slice(1, c - 1)
        keep = lambda a, s=mid: np.ascontiguousarray(a[:, s].reshape((G * (c - 2),) + a.shape[2:]))
        past, fut = slice(0, c - 2), slice(2, c)
        out = {'hr': keep(hr), 'prev_hr': keep(hr, past), 'next_hr': keep(hr, fut),
               'lr': keep(lr), 'prev_lr': keep(lr, past), 'next_lr': keep(lr, fut),
               'bic': keep(bic), 'prev_bic': keep(bic, past), 'next_bic': keep(bic, fut),
               'est': keep(est), 'prev_est': keep(est, past), 'next_est': keep(est, fut)}
        warp = self.hr_warping.predict
        out['prev_hat_est'] = warp([out['lr'], out['prev_lr'], out['prev_est']])
        out['prev_hat_hr'] = warp([out['lr'], out['prev_lr'], out['prev_hr']])
        out['next_hat_est'] = warp([out['lr'], out['next_lr'], out['next_est']])
        out['next_hat_hr'] = warp([out['lr'], out['next_lr'], out['next_hr']])
        return out

    def gan_iteration(self, hr_clips):
        """One iteration of the adversarial loop body (TecoGAN.py:434-479): the discriminator re-compiled, one step on the
        real triplets (label 1) and one on the generated ones (label 0), then two combined steps with the feature maps
        of the real triplets as targets.  Returns the last tecogan.train_on_batch losses."""
        b = self.make_gan_batch(hr_clips)
        k = len(b['hr'])
        self._last_gan_batch = b
        if not dp.active() and not self.step_by_step:
            # one native call: every array goes to the device once, the feature-map targets never leave it
            # (sr_tecogan_iteration; the calls below are the same pass through the Keras-shaped model objects)
            opt = self.opt
            ctx, d = self.core.ctx(k), self.discriminator.native(k)
            self.tecogan.attach_vgg(ctx, k)
            loss, _ = sr_hip.tecogan_iteration(ctx, d, b, self.tecogan.loss_weights, opt.iterations, lr=opt.lr, beta_1=opt.beta_1,
                                               beta_2=opt.beta_2, epsilon=opt.epsilon, train_disc=self.tecogan.train_discriminator)
            opt.iterations += 4
            self.core._dirty = self.discriminator._dirty = True
            self.discriminator._fresh_moments = False
            self.discriminator.trainable = False
            return loss
        ones, zeros = np.ones(k, np.float32), np.zeros(k, np.float32)
        self.discriminator.trainable = True
        self.discriminator.compile(loss=['binary_crossentropy'] + ['mean_squared_error'] * 4,
                                   loss_weights=[1., 0., 0., 0., 0.], optimizer=self.opt)
        real = [b['prev_hr'], b['hr'], b['next_hr'], b['prev_hat_hr'], b['hr'], b['next_hat_hr'], b['prev_bic'], b['bic'], b['next_bic']]
        fake = [b['prev_est'], b['est'], b['next_est'], b['prev_hat_est'], b['est'], b['next_hat_est'], b['prev_bic'], b['bic'], b['next_bic']]
        # (the reference also predicts d_1..d_4 here, TecoGAN.py:446: they only feed zero-weighted losses)
        self.discriminator.train_on_batch(real, [ones, None, None, None, None])
        self.discriminator.train_on_batch(fake, [zeros, None, None, None, None])
        self.discriminator.trainable = False
        _, d_1, d_2, d_3, d_4 = self.discriminator.predict(real)
        net_input = [b['lr'], b['prev_lr'], b['prev_est'], b['next_est'], b['next_hat_est'], b['prev_bic'], b['bic'], b['next_bic']]
        net_output = [b['hr'], ones, d_1, d_2, d_3, d_4, b['hr'], b['hr'], b['hr'], b['hr'], np.ascontiguousarray(b['hr'][::-1]), b['lr']]
        loss = None
        for _ in range(2):
            loss = self.tecogan.train_on_batch(net_input, net_output)
        return loss

    # ------------------------------------------------------------------ training (TecoGAN.py:136-527)
    def train(self):
        cfg = self.configs['train']
        rank, world, _ = dp.world()
        root = self.configs.get('root_dir', './')
        pretrain_gen = True
        if cfg.get('pretrained_disc'):
            print('[INFO] Loading pretrained discriminator ...')
            pre = cfg['pretrained_disc']
            self.discriminator.load_weights(pre if os.path.isabs(pre) else os.path.join(root, pre))
            pretrain_gen = False
            print('[INFO] Discriminator ready.')
        i = self._start_iteration('[INFO] Generator ready.')
        if pretrain_gen:
            print('[INFO] Starting generator training process ...')
            self._train_generator(i, cfg['iterations'] / 2, self.gen_learning_rate,
                                  'generator_model_weights_%d.h5', 'TecoGAN.yaml', 'Generator')
            i = 0                                                   # TecoGAN.py:291-292
        self.opt.lr = 5e-5                                          # TecoGAN.py:278 (one optimizer: all three models)
        print('\n[INFO] Generator learning rate: %f' % self.generator.optimizer.lr)
        print('[INFO] Discriminator learning rate: %f' % self.discriminator.optimizer.lr)
        print('[INFO] TecoGAN learning rate: %f' % self.tecogan.optimizer.lr)
        print('[INFO] Starting TecoGAN training process ...')
        samples_dir = os.path.join(self.output_dir, 'Samples')
        checkpoints_dir = os.path.join(self.output_dir, 'Checkpoints')
        if rank == 0:
            os.makedirs(samples_dir, exist_ok=True)
            os.makedirs(checkpoints_dir, exist_ok=True)
        videos = get_videos(self.configs)
        if not videos:
            raise ValueError('no training clips found in data.videos')
        if world > 1:                                               # every rank starts from rank 0's weights
            dp.init_if_needed()
            self.core.refresh()
            self.core.set_tensors(dp.broadcast_weights_(self.core.tensors, self.core.device))
            self.discriminator.set_weights([a for _, a in dp.broadcast_weights_(self.discriminator.named_weights(), self.core.device)])
        rng = np.random.RandomState(None if cfg.get('seed') is None else int(cfg['seed']) + 1000 + rank)
        clips = {}
        loss = None
        while i < cfg['iterations']:
            picks = rng.randint(0, len(videos), size=cfg['batch_size'])
            hr = []
            for j in picks:
                if j not in clips:
                    clips[j] = load_clip(videos[j])
                f = get_frames(clips[j], self.configs, rng)
                hr.append(normalize(f) if f.dtype == np.uint8 else f.astype(np.float32))
            loss = self.gan_iteration(np.stack(hr))
            if rank == 0 and i % cfg['info_freq'] == 0:
                print('[INFO]', '-' * 15, 'Iteration %d' % i, '-' * 15)
                print('[INFO] Total loss: %f \t Gen loss: %f \t Disc loss: %f' % (loss[0], loss[1], loss[2]))
            if rank == 0 and i % cfg['sample_freq'] == 0:
                b = self._last_gan_batch                            # sample 0 of the batch, TecoGAN.py:491-510
                est, _ = self.generator.predict([b['lr'][0:1], b['prev_lr'][0:1], b['prev_est'][0:1]])
                bic = sr_hip.cubic_upscale4(b['lr'][0:1], device=self.core.device)[0]
                print('[INFO] Generator PSNR: %f \t Bicubic PSNR: %f' % (psnr(est[0], b['hr'][0]), psnr(bic, b['hr'][0])))
                window = np.concatenate([denormalize(np.clip(bic, 0, 1)), denormalize(np.clip(est[0], 0, 1)), denormalize(b['hr'][0])], axis=0)
                np.save(os.path.join(samples_dir, 'Sample_%d.npy' % i), window)
            if rank == 0 and i % cfg['checkpoint_freq'] == 0:
                print('[INFO] Saving model ...')
                gen_weights = os.path.join(checkpoints_dir, 'generator_model_weights_%d.h5' % i)
                disc_weights = os.path.join(checkpoints_dir, 'discriminator_model_weights_%d.h5' % i)
                self.generator.save_weights(gen_weights)
                self.discriminator.save_weights(disc_weights)
                print('[INFO] Model saved.')
                self.configs['cnn']['pretrained_model'] = gen_weights
                self.configs['train']['pretrained_disc'] = disc_weights
                with open(os.path.join(self.output_dir, 'TecoGAN.yaml'), 'w') as fh:
                    yaml.dump(self.configs, fh, default_flow_style=False)
            i += 1
        print('[INFO] Training process ready.')
        return loss
