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

    def super_resolve_clips(self, lr_clips, hr_clips=None, names=None, metrics_file=None):
        """The eval loop on SEVERAL clips in lock-step (what bench.py's `end_to_end` leg times):
        clip k's rows x cols tiles are batch entries k*rows*cols .., one native step advances frame t of every clip, the
        estimates leave the device as uint8 frames (the reference's denormalize, utils.py:60-62) through page-locked double
        buffers -- the copy of frame t runs under the kernels of frame t+1 (sr_clip_push_u8) -- and every 10th frame the
        float estimates are read back for PSNR / SSIM exactly as the reference computes them (FRVSR.py:320-339).  Clips of
        unequal length finish independently (a finished clip's batch entries keep running on stale input, ignored).
        lr_clips: list of [T_k, rows*h, cols*w, 3] float32; hr_clips: matching float32 HR clips or None.
        Returns (list of uint8 [T_k, rows*4h, cols*4w, 3], list of stats dicts, timing dict)."""
        rows = int(self.configs['data'].get('rows', 1))
        cols = int(self.configs['data'].get('cols', 1))
        h, w, _ = self.lr_shape
        K, tiles = len(lr_clips), rows * cols
        B = K * tiles
        for c in lr_clips:
            if c.shape[1:] != (rows * h, cols * w, 3):
                raise ValueError(f'expected LR frames of shape {(rows * h, cols * w, 3)}, got {c.shape[1:]}')
        names = list(names) if names is not None else ['clip%d' % k for k in range(K)]
        Ts = [int(c.shape[0]) for c in lr_clips]
        gen = self._generator()
        ctx = gen.core.ctx(B)
        # Page-locked staging.  When the whole group fits (eval.pinned_limit_mb, default 8192: T x B frames of LR float32 +
        # uint8 estimates), every frame has its own slot: the LR tiles are staged before the loop (the reference's tiling
        # is outside its inference bracket too, FRVSR.py:280-296) and the estimates land where they stay -- no host copy
        # per step.  Otherwise two slots each: stage / consume frame by frame beside the GPU's work.
        Tmax = max(Ts)
        lr_bytes, out_bytes = B * h * w * 3 * 4, B * 16 * h * w * 3
        limit = int(float(self.configs.get('eval', {}).get('pinned_limit_mb', 8192)) * (1 << 20))
        slots = Tmax if Tmax * (lr_bytes + out_bytes) <= limit else 2
        lr_pin = sr_hip.PinnedArray((slots, B, h, w, 3), np.float32)
        out_pin = sr_hip.PinnedArray((slots, B, 4 * h, 4 * w, 3), np.uint8)
        lr_pin.a[...] = 0
        est_pin = sr_hip.PinnedArray((B, 4 * h, 4 * w, 3), np.float32) if hr_clips is not None else None   # metric frames
        outs = [np.zeros((T, rows * 4 * h, cols * 4 * w, 3), np.uint8) for T in Ts]
        for o in outs:
            o.fill(0)            # touch the pages now, not inside the loop
        stats = [{'psnr': [], 'ssim': [], 'bic_psnr': [], 'bic_ssim': [], 'ms': []} for _ in range(K)]
        timing = {'frames': 0, 'loop_s': 0.0, 'metrics_s': 0.0, 'slots': slots}
        now = datetime.datetime.now

        def stage(t):            # LR tiles of frame t of every clip that has one
            dst = lr_pin.a[t % slots]
            for k in range(K):
                if t < Ts[k]:
                    dst[k * tiles:(k + 1) * tiles] = split_tiles(lr_clips[k][t], rows, cols, h, w)

        def consume(t):          # frame t of every clip that has one has landed in out_pin[t % slots]
            src = out_pin.a[t % slots]
            for k in range(K):
                if t < Ts[k]:
                    outs[k][t] = merge_tiles(src[k * tiles:(k + 1) * tiles], rows, cols)

        def score(est_f, t, active, ms):     # PSNR / SSIM rows of metric frame t (FRVSR.py:320-339); host time, not loop time
            t1 = now()
            for k in active:
                est = merge_tiles(est_f[k * tiles:(k + 1) * tiles], rows, cols)
                hr = hr_clips[k][t]
                p, q = psnr(hr, est), ssim(hr, est)
                stats[k]['psnr'].append(p)
                stats[k]['ssim'].append(q)
                bic = sr_hip.cubic_upscale4(lr_clips[k][t:t + 1], device=self.core.device)[0]   # FRVSR.py:304,322,328
                stats[k]['bic_psnr'].append(psnr(bic, hr))
                stats[k]['bic_ssim'].append(ssim(bic, hr))
                if metrics_file is not None:    # the reference's row (FRVSR.py:339)
                    metrics_file.write('Video:\t%s\tFrame:\t%d\tInference time:\t%d\tPSNR:\t%f\tSSIM:\t%f\n'
                                       % (names[k], t, ms / len(active), p, q))
            timing['metrics_s'] += (now() - t1).total_seconds()

        pending = None           # (frame, its active clips, its ms): a float read-back in flight
        try:
            if slots == Tmax:
                for t in range(Tmax):
                    stage(t)
            ctx.clip_begin(B)        # zero state for prev LR tiles and prev estimates (reference FRVSR.py:233-234)
            for t in range(Tmax):
                t0 = now()
                active = [k for k in range(K) if t < Ts[k]]
                if slots != Tmax:
                    stage(t)
                ctx.clip_push_u8(lr_pin.ptr + (t % slots) * lr_bytes, out_pin.ptr + (t % slots) * out_bytes)
                ctx.u8_wait(1)
                if slots != Tmax and t > 0:
                    consume(t - 1)
                # metric frames (every 10th, FRVSR.py:320): the float estimates of frame t leave the device beside step t + 1
                # (sr_clip_read_est_begin) and are scored two iterations later, or after the loop
                est_f = None
                if pending is not None and t >= pending[0] + 2:      # two steps later: the copy never holds the next push back
                    ctx.clip_read_est_end()
                    est_f, (tm, active_m, ms_m) = est_pin.a, pending
                    pending = None
                ms = (now() - t0).total_seconds() * 1000
                if t % 10 == 0 and hr_clips is not None:
                    ta = now()
                    ctx.clip_read_est_begin(est_pin.a, 0, B)
                    ms += (now() - ta).total_seconds() * 1000
                    pending = (t, active, ms)
                timing['loop_s'] += ms / 1000
                timing['frames'] += len(active)
                for k in active:
                    stats[k]['ms'].append(ms / len(active))
                if est_f is not None:
                    score(est_f, tm, active_m, ms_m)
            t0 = now()
            ctx.u8_wait(0)
            if pending is not None:
                ctx.clip_read_est_end()
            timing['loop_s'] += (now() - t0).total_seconds()
            if pending is not None:
                score(est_pin.a, *pending)
                pending = None
            if slots == Tmax:
                for t in range(Tmax):
                    consume(t)
            else:
                consume(Tmax - 1)
        finally:
            if pending is not None:      # an exception between begin and end: the copy must be over before the slot is freed
                try:
                    ctx.clip_read_est_end()
                except Exception:
                    pass
            ctx.sync()
            lr_pin.close()
            out_pin.close()
            if est_pin is not None:
                est_pin.close()
        return outs, stats, timing

    def eval(self):
        """Reference FRVSR.py:205-376 on the rank's shard of the clips, `eval.clips_in_flight` (default 16) clips at a time
        in lock-step (super_resolve_clips).  metrics.txt keeps the reference's rows and totals; the rows of one group are
        written frame by frame, so clips of a group interleave (each row names its video)."""
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
        n_fly = max(1, int(self.configs['eval'].get('clips_in_flight', 16)))
        synth = str(self.configs['data'].get('lr_synthesis', 'blur_cubic'))
        total, total_ssim, total_bic, total_bic_ssim, n = 0.0, 0.0, 0.0, 0.0, 0
        frames, loop_s = 0, 0.0
        with open(os.path.join(out_dir, 'metrics.txt' if world == 1 else 'metrics_rank%d.txt' % rank), 'w+') as f:
            mine = clips[lo:hi]
            for g0 in range(0, len(mine), n_fly):
                group = mine[g0:g0 + n_fly]
                hrs, lrs = [], []
                for path in group:
                    clip = load_clip(path)
                    hr = normalize(clip) if clip.dtype == np.uint8 else clip.astype(np.float32)
                    if hr.shape[1:3] != (H, W):
                        raise ValueError(f'{path}: frames are {hr.shape[1:3]}, the configured HR canvas is {(H, W)}')
                    # the reference's LR synthesis (FRVSR.py:274-277); data.lr_synthesis: box selects a plain average
                    if synth == 'box':
                        lr = box_downscale(hr, 4)
                    elif synth == 'blur_cubic_host':
                        lr = blur_cubic_downscale(hr)                         # numpy restatement
                    else:
                        lr = sr_hip.blur_cubic_downscale(hr, device=self.core.device)   # same arithmetic, one fused GPU pass
                    hrs.append(hr)
                    lrs.append(lr)
                names = [os.path.basename(p) for p in group]
                ests, stats, timing = self.super_resolve_clips(lrs, hrs, names, f)
                frames += timing['frames']
                loop_s += timing['loop_s']
                for path, est, st in zip(group, ests, stats):
                    np.save(os.path.join(out_dir, os.path.basename(path)), est)
                    total += sum(st['psnr'])
                    total_ssim += sum(st['ssim'])
                    total_bic += sum(st['bic_psnr'])
                    total_bic_ssim += sum(st['bic_ssim'])
                    n += len(st['psnr'])
                    print('[INFO] %s: %d frames, FRVSR inference %.2f ms/frame, PSNR %.3f' % (
                        os.path.basename(path), len(est), float(np.mean(st['ms'])),
                        float(np.mean(st['psnr'])) if st['psnr'] else float('nan')))
                f.write('\n')
                print('[INFO] %d clips in lock-step: %d frames in %.3f s = %.1f frames/s (LR frames in, uint8 frames and the '
                      'metric frames out; metrics %.3f s on the host beside it)'
                      % (len(group), timing['frames'], timing['loop_s'], timing['frames'] / max(timing['loop_s'], 1e-9), timing['metrics_s']))
            mean_psnr = total / n if n else float('nan')
            mean_ssim = total_ssim / n if n else float('nan')
            f.write('Total PSNR:\t%f\tTotal SSIM:\t%f' % (mean_psnr, mean_ssim))    # FRVSR.py:368
            if n:                                                                     # FRVSR.py:369, typo included
                f.write('Total bicubuc PSNR:\t%f\tTotal bicubic SSIM:\t%f' % (total_bic / n, total_bic_ssim / n))
                print('\n[INFO] Avg. Bicubic PSNR: %f\tAvg. Bicubic SSIM: %f' % (total_bic / n, total_bic_ssim / n))
        self.eval_fps = frames / loop_s if loop_s > 0 else float('nan')
        print('\n[INFO] Avg. FRVSR PSNR: %f\tAvg. FRVSR SSIM: %f\tInference: %.1f frames/s' % (mean_psnr, mean_ssim, self.eval_fps))
        return mean_psnr
