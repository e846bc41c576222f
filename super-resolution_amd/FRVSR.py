"""Frame-Recurrent Video Super-Resolution on MI355X: drop-in for reference FRVSR.py.

Same class, constructor and stage methods as the reference (FRVSR.py:19-26, build :27,
train :76 (native training step, see train / make_train_batch), eval :205, run :379); the generator graph of FRVSR.py:55-66 is one native
context (libsr_hip.so), reachable through the same attributes:
``frvsr, fnet, srnet, upscaling, hr_warp, lr_warp, space2depth``.

eval() keeps the reference's recurrent driver semantics (FRVSR.py:231-234, 280-309):
rows x cols independent tiles batched through one step, zero initial state, estimate
state := output, previous LR := current tiles, timing bracket around the step, PSNR and
SSIM every 10th frame, ``metrics.txt`` with the reference's columns.  Differences, on purpose
(SURVEY.md Appendix B): state is reset per clip, the mean PSNR divides by the number of
samples actually summed, clips are .npy arrays (video decode / YOLO are out of scope).
"""
import datetime
import os

import numpy as np

import yaml

import dp
import sr_hip
from layers import (Adam, FNet, GeneratorCore, GeneratorModel, Space2Depth, SRNet, Upscaling, Warp)
from utils import (blur_cubic_downscale, box_downscale, denormalize, get_frames, get_videos, load_clip,
                   merge_tiles, normalize, psnr, shard_range, split_tiles, ssim)


class FrameRecurrentVideoSR(object):
    model_kind = 'frvsr'

    def __init__(self, lr_shape, hr_shape, output_dir, configs):
        self.lr_shape = tuple(lr_shape)
        self.hr_shape = tuple(hr_shape)
        self.output_dir = output_dir
        self.configs = configs

    def _check_shapes(self):
        if self.lr_shape[0] * 4 != self.hr_shape[0] or self.lr_shape[1] * 4 != self.hr_shape[1]:
            raise ValueError('Incompatible resolutions with upscale x4.')
        if (self.lr_shape[0] % 8 or self.lr_shape[1] % 8) and not self._pad_to_8():
            raise ValueError('low resolution height and width must be multiples of 8 '
                             '(FNet pools 3x and upsamples 3x; reference layers.py:31-41); '
                             'data.pad_to_8: true pads inside the native context (inference only)')

    def _pad_to_8(self):
        data = self.configs.get('data') if isinstance(self.configs, dict) else None
        return bool(data and data.get('pad_to_8'))

    def build(self):
        self._check_shapes()
        dtype = (self.configs.get('dtype') or 'fp32') if isinstance(self.configs, dict) else 'fp32'
        device = int(os.environ.get('LOCAL_RANK', '0'))
        pad = self._pad_to_8() and bool(self.lr_shape[0] % 8 or self.lr_shape[1] % 8)
        self.core = GeneratorCore(self.lr_shape, model=self.model_kind, dtype=dtype, device=device, pad=pad)
        print('[INFO] Creating FNet ...')
        self.fnet = FNet(self.lr_shape, core=self.core).build()
        print('[INFO] Creating SRNet ...')
        self.srnet = SRNet(self.lr_shape, core=self.core).build()
        self.upscaling = Upscaling(self.lr_shape, device=device).build()
        self.hr_warp = Warp(self.hr_shape, 1, device=device).build()
        self.lr_warp = Warp(self.lr_shape, 2, device=device).build()
        self.space2depth = Space2Depth(self.hr_shape, device=device).build()
        self.frvsr = GeneratorModel(self.core, 'FRVSR')
        # loss functions and learning rate of the reference (FRVSR.py:69-71)
        self.opt = Adam(lr=1e-4)
        self.frvsr.compile(loss=['mean_squared_error', 'mean_squared_error'], loss_weights=[1., 1.], optimizer=self.opt)
        print('[INFO] Frame-Recurrent Video Super Resolution ready.')

    # the model object holding the generator ('frvsr' here, 'generator' in TecoGAN)
    def _generator(self):
        return self.frvsr

    def _load_pretrained(self):
        pre = self.configs['cnn'].get('pretrained_model')
        if pre:
            path = pre if os.path.isabs(pre) else os.path.join(self.configs.get('root_dir', './'), pre)
            print('[INFO] Loading pretrained model ...')
            self._generator().load_weights(path)

    # ------------------------------------------------------------------ training
    def make_train_batch(self, hr_batch):
        """hr_batch [B_clips, c, H, W, 3] float32 in [0,1] -> ([lr, prev_lr, prev_est], [hr, lr]) with
        B_clips*c samples in the reference's order b = clip*c + frame (FRVSR.py:115-155): LR = blur + cubic
        downscale, previous LR frame (zeros at frame 0), previous estimate = the generator's own output for
        the previous frame, computed WITHOUT gradient (zeros at frame 0).  The reference computes those
        estimates one predict per sample and, from frame 2 on, feeds predict the estimate of frame b-3
        (``prev_est_batch[b-2]``, FRVSR.py:149); here the clips advance in lock step through the
        device-resident recurrence, est_f = G(lr_f, lr_f-1, est_f-1) (SURVEY.md Appendix B decision)."""
        n, c = hr_batch.shape[:2]
        # LR synthesis on the GPU (fused blur + cubic gather; the numpy restatement takes 236 ms per 40 frames)
        lr = sr_hip.blur_cubic_downscale(hr_batch.reshape((n * c,) + hr_batch.shape[2:]), device=self.core.device)
        lr = lr.reshape((n, c) + lr.shape[1:])                                   # [n, c, h, w, 3]
        gen = self._generator()
        prev_est = np.empty((n, c) + self.hr_shape, np.float32)
        prev_est[:, 0] = 0.0                                                     # frame 0: zero state (FRVSR.py:124-125)
        gen.clip_begin(n)
        for f in range(c - 1):
            prev_est[:, f + 1] = gen.clip_push(lr[:, f])
        prev_lr = np.zeros_like(lr)
        prev_lr[:, 1:] = lr[:, :-1]
        flat = lambda a: np.ascontiguousarray(a.reshape((n * c,) + a.shape[2:]))
        lr_b = flat(lr)
        return [lr_b, flat(prev_lr), flat(prev_est)], [flat(hr_batch), lr_b]

    def train(self):
        """Reference FRVSR.py:76-202.  One process per GPU: every rank draws its own ``batch_size`` clips, the
        gradient is averaged over the ranks (layers.GeneratorModel.train_on_batch), rank 0 writes samples
        and checkpoints.  Samples are .npy arrays (LR-bicubic | estimate | HR side by side rows as in the
        reference's JPG, no OpenCV here)."""
        i = self._start_iteration('[INFO] Model ready.')
        print('[INFO] Starting training process...')
        loss = self._train_generator(i, self.configs['train']['iterations'] + 1, None,
                                     'frvsr_model_weights_%d.h5', 'FRVSR.yaml', 'FRVSR')
        print('[INFO] Training process ready.')
        return loss

    def _start_iteration(self, ready_msg):
        """Load cnn.pretrained_model and take the start iteration from its file name (FRVSR.py:85-96)."""
        i = 0
        pre = self.configs['cnn'].get('pretrained_model')
        if pre:
            self._load_pretrained()
            try:
                i = int(os.path.splitext(pre)[0].split('_')[-1])
            except ValueError:
                print('[Info] Not valid initial iteration. Setting initial iteration to 0.')
                i = 0
            print(ready_msg)
        return i

    def _train_generator(self, i, end, lr_schedule, ckpt_pattern, yaml_name, label):
        """The generator training loop shared by FRVSR.train (FRVSR.py:100-200) and TecoGAN's generator
        pre-training (TecoGAN.py:178-275): iterations i .. end-1; lr_schedule(i) -> learning rate or None."""
        cfg = self.configs['train']
        rank, world, _ = dp.world()
        samples_dir = os.path.join(self.output_dir, 'Samples')
        checkpoints_dir = os.path.join(self.output_dir, 'Checkpoints')
        if rank == 0:
            os.makedirs(samples_dir, exist_ok=True)
            os.makedirs(checkpoints_dir, exist_ok=True)
        videos = get_videos(self.configs)
        if not videos:
            raise ValueError('no training clips found in data.videos')
        gen = self._generator()
        if world > 1:
            dp.init_if_needed()
            self.core.refresh()
            self.core.set_tensors(dp.broadcast_weights_(self.core.tensors, self.core.device))
        rng = np.random.RandomState(None if cfg.get('seed') is None else int(cfg['seed']) + rank)   # train.seed: optional, reproducible runs
        clips = {}
        loss = None
        while i < end:
            picks = rng.randint(0, len(videos), size=cfg['batch_size'])
            hr = []
            for j in picks:
                if j not in clips:
                    clips[j] = load_clip(videos[j])
                hr.append(get_frames(clips[j], self.configs, rng))
            # decoded uint8 frames go to the device as they are (normalised there); anything else as float32
            if not all(f.dtype == np.uint8 for f in hr):
                hr = [normalize(f) if f.dtype == np.uint8 else f.astype(np.float32) for f in hr]
            hr = np.stack(hr)
            if lr_schedule is not None:
                gen.optimizer.lr = float(lr_schedule(i))          # K.set_value(optimizer.lr, ...) in the reference
            # LR synthesis, previous frames / estimates and train_on_batch in one device-resident call
            # (make_train_batch + gen.train_on_batch is the same iteration through host arrays)
            loss = gen.train_on_clips(hr)
            if rank == 0 and i % cfg['info_freq'] == 0:
                print('[INFO]', '-' * 15, 'Iteration %d' % i, '-' * 15)
                print('[INFO] Total loss: %f \t SR loss: %f \t Flow loss: %f \t Learning rate: %f'
                      % (loss[0], loss[1], loss[2], gen.optimizer.lr))
            if rank == 0 and i % cfg['sample_freq'] == 0:
                # first frame of the first clip, zero state (the reference's sample 0, FRVSR.py:165-170)
                hr0 = normalize(hr[0, 0]) if hr.dtype == np.uint8 else hr[0, 0]
                lr0 = sr_hip.blur_cubic_downscale(hr0[None], device=self.core.device)[0]
                est, _ = gen.predict([lr0[None], np.zeros_like(lr0)[None], np.zeros_like(hr0)[None]])
                bic = sr_hip.cubic_upscale4(lr0[None], device=self.core.device)[0]
                print('[INFO] %s PSNR: %f \t Bicubic PSNR: %f' % (label, psnr(est[0], hr0), psnr(bic, hr0)))
                window = np.concatenate([denormalize(np.clip(bic, 0, 1)), denormalize(np.clip(est[0], 0, 1)), denormalize(hr0)], axis=0)
                np.save(os.path.join(samples_dir, 'Sample_%d.npy' % i), window)
            if rank == 0 and i % cfg['checkpoint_freq'] == 0:
                print('[INFO] Saving model...')
                path = os.path.join(checkpoints_dir, ckpt_pattern % i)
                gen.save_weights(path)
                print('[INFO] Model saved.')
                self.configs['cnn']['pretrained_model'] = path
                with open(os.path.join(self.output_dir, yaml_name), 'w') as fh:
                    yaml.dump(self.configs, fh, default_flow_style=False)
            i += 1
        return loss

    def run(self):
        raise NotImplementedError('the webcam / video demo UI (reference FRVSR.py:379-536) is out of scope')

    # ------------------------------------------------------------------ evaluation
    def super_resolve_clip(self, lr_frames, hr_frames=None, name='clip', metrics_file=None):
        """lr_frames [T, rows*h, cols*w, 3] float32 in [0,1] -> estimated [T, rows*4h, cols*4w, 3].
        Returns (frames, stats) with stats = dict(psnr=[...], ssim=[...] (every 10th frame), ms=[...per frame...])."""
        rows = int(self.configs['data'].get('rows', 1))
        cols = int(self.configs['data'].get('cols', 1))
        h, w, _ = self.lr_shape
        T = lr_frames.shape[0]
        if lr_frames.shape[1:] != (rows * h, cols * w, 3):
            raise ValueError(f'expected LR frames of shape {(rows * h, cols * w, 3)}, got {lr_frames.shape[1:]}')
        gen = self._generator()
        gen.clip_begin(rows * cols)            # zero state for prev LR tiles and prev estimates
        out = np.empty((T, rows * 4 * h, cols * 4 * w, 3), np.float32)
        stats = {'psnr': [], 'ssim': [], 'bic_psnr': [], 'bic_ssim': [], 'ms': []}
        for t in range(T):
            tiles = split_tiles(lr_frames[t], rows, cols, h, w)
            t0 = datetime.datetime.now()
            est_tiles = gen.clip_push(tiles)
            ms = (datetime.datetime.now() - t0).total_seconds() * 1000
            out[t] = merge_tiles(est_tiles, rows, cols)
            stats['ms'].append(ms)
            if t % 10 == 0 and hr_frames is not None:
                p = psnr(hr_frames[t], out[t])
                q = ssim(hr_frames[t], out[t])
                stats['psnr'].append(p)
                stats['ssim'].append(q)
                bic = sr_hip.cubic_upscale4(lr_frames[t:t + 1], device=self.core.device)[0]    # the reference's bicubic baseline (FRVSR.py:304,322,328)
                stats['bic_psnr'].append(psnr(bic, hr_frames[t]))
                stats['bic_ssim'].append(ssim(bic, hr_frames[t]))
                if metrics_file is not None:    # the reference's row (FRVSR.py:339)
                    metrics_file.write('Video:\t%s\tFrame:\t%d\tInference time:\t%d\tPSNR:\t%f\tSSIM:\t%f\n'
                                       % (name, t, ms, p, q))
        return out, stats

    def eval(self):
        self._load_pretrained()
        out_dir = self.configs['eval']['output_dir']
        os.makedirs(out_dir, exist_ok=True)
        rank = int(os.environ.get('RANK', '0'))
        world = int(os.environ.get('WORLD_SIZE', '1'))
        clips = get_videos(self.configs)
        lo, hi = shard_range(len(clips), world, rank)       # clips are independent: shard, no collective
        rows = int(self.configs['data'].get('rows', 1))
        cols = int(self.configs['data'].get('cols', 1))
        H, W = self.hr_shape[0] * rows, self.hr_shape[1] * cols
        total, total_ssim, total_bic, total_bic_ssim, n = 0.0, 0.0, 0.0, 0.0, 0
        with open(os.path.join(out_dir, 'metrics.txt' if world == 1 else 'metrics_rank%d.txt' % rank), 'w+') as f:
            for path in clips[lo:hi]:
                clip = load_clip(path)
                hr = normalize(clip) if clip.dtype == np.uint8 else clip.astype(np.float32)
                if hr.shape[1:3] != (H, W):
                    raise ValueError(f'{path}: frames are {hr.shape[1:3]}, the configured HR canvas is {(H, W)}')
                # the reference's LR synthesis (FRVSR.py:274-277); data.lr_synthesis: box selects a plain average
                synth = str(self.configs['data'].get('lr_synthesis', 'blur_cubic'))
                if synth == 'box':
                    lr = box_downscale(hr, 4)
                elif synth == 'blur_cubic_host':
                    lr = blur_cubic_downscale(hr)                         # numpy restatement
                else:
                    lr = sr_hip.blur_cubic_downscale(hr, device=self.core.device)   # same arithmetic, one fused GPU pass
                est, stats = self.super_resolve_clip(lr, hr, os.path.basename(path), f)
                np.save(os.path.join(out_dir, os.path.basename(path)), denormalize(est))
                total += sum(stats['psnr'])
                total_ssim += sum(stats['ssim'])
                total_bic += sum(stats['bic_psnr'])
                total_bic_ssim += sum(stats['bic_ssim'])
                n += len(stats['psnr'])
                f.write('\n')
                print('[INFO] %s: %d frames, FRVSR inference %.2f ms/frame, PSNR %.3f' % (
                    os.path.basename(path), len(est), float(np.mean(stats['ms'])),
                    float(np.mean(stats['psnr'])) if stats['psnr'] else float('nan')))
            mean_psnr = total / n if n else float('nan')
            mean_ssim = total_ssim / n if n else float('nan')
            f.write('Total PSNR:\t%f\tTotal SSIM:\t%f' % (mean_psnr, mean_ssim))    # FRVSR.py:368
            if n:                                                                     # FRVSR.py:369, typo included
                f.write('Total bicubuc PSNR:\t%f\tTotal bicubic SSIM:\t%f' % (total_bic / n, total_bic_ssim / n))
                print('\n[INFO] Avg. Bicubic PSNR: %f\tAvg. Bicubic SSIM: %f' % (total_bic / n, total_bic_ssim / n))
        print('\n[INFO] Avg. FRVSR PSNR: %f\tAvg. FRVSR SSIM: %f' % (mean_psnr, mean_ssim))
        return mean_psnr
