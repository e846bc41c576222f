"""Drop-in counterparts of the reference's generator blocks (reference layers.py:15-140).

Same class names, constructor arguments and ``build()`` protocol as the
reference; ``build()`` returns a model object with the slice of the Keras
``Model`` API the reference's call sites use: ``name``, ``predict``,
``load_weights``, ``save_weights``, ``get_weights``, ``set_weights``.  All
arithmetic runs in libsr_hip.so through ``sr_hip`` (ctypes); nothing here
computes on the CPU and there is no fallback.

Weighted blocks (FNet, SRNet) are views of a ``GeneratorCore``, which owns the
76 Keras-named tensors and the native context.  A block built on its own gets a
private core; blocks built by ``FRVSR.FrameRecurrentVideoSR.build`` /
``TecoGAN.TecoGAN.build`` share the model's core, exactly as the reference's
sub-models share variables with ``frvsr`` / ``generator``.
"""
from __future__ import annotations

import os

import numpy as np

import sr_hip
import weights_io

# (cin, cout) in creation order: reference layers.py:30-44 and :67-75
FNET_CONVS = [(6, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256),
              (256, 128), (128, 128), (128, 64), (64, 64), (64, 32), (32, 2)]
SRNET_LAYERS = [("conv", 51, 64)] + [("conv", 64, 64)] * 20 + [("convT", 64, 64)] * 2 + [("conv", 64, 3)]


def weight_specs():
    """[(keras_name, shape)] of the 76 tensors in Keras 2.2.4 save order:
    FNet conv2d_1..14, SRNet conv2d_15..35, conv2d_transpose_1..2, conv2d_36; kernel then bias."""
    specs, n, m = [], 1, 1
    for cin, cout in FNET_CONVS:
        specs += [(f"FNet/conv2d_{n}/kernel:0", (3, 3, cin, cout)), (f"FNet/conv2d_{n}/bias:0", (cout,))]
        n += 1
    for kind, cin, cout in SRNET_LAYERS:
        if kind == "conv":
            specs += [(f"SRNet/conv2d_{n}/kernel:0", (3, 3, cin, cout)), (f"SRNet/conv2d_{n}/bias:0", (cout,))]
            n += 1
        else:   # Conv2DTranspose kernels are (kh, kw, filters, input_dim)
            specs += [(f"SRNet/conv2d_transpose_{m}/kernel:0", (3, 3, cout, cin)),
                      (f"SRNet/conv2d_transpose_{m}/bias:0", (cout,))]
            m += 1
    return specs   # creation order: the two transposed convs sit between conv2d_35 and conv2d_36


def glorot_init(seed=None):
    """Fresh-model initialisation: glorot_uniform kernels, zero biases (Keras 2.2.4 defaults,
    nothing is overridden in reference layers.py)."""
    rng = np.random.default_rng(seed)
    out = []
    for name, shape in weight_specs():
        if len(shape) == 4:
            limit = np.sqrt(6.0 / (9 * shape[2] + 9 * shape[3]))
            out.append((name, rng.uniform(-limit, limit, size=shape).astype(np.float32)))
        else:
            out.append((name, np.zeros(shape, np.float32)))
    return out


class GeneratorCore:
    """Weights + native context of one generator (FNet + SRNet + warps) for one LR tile shape."""

    def __init__(self, lr_shape, model="frvsr", dtype="fp32", device=0, seed=None, pad=False):
        """pad: LR height / width need not be multiples of 8 (sr_create_padded: zero padding inside the native
        context, outputs cropped; inference only)."""
        h, w, c = (int(v) for v in lr_shape)
        if c != 3:
            raise ValueError(f"generator expects 3 channels, got {c}")
        self.lr_shape = (h, w, c)
        self.hr_shape = (4 * h, 4 * w, c)
        self.model, self.dtype, self.device, self.pad = model, dtype, device, bool(pad)
        self.tensors = glorot_init(seed)      # [(keras_name, array)] in file order
        self._ctx = None
        self._ctx_batch = 0
        self._dirty = False                   # the ctx holds newer weights than self.tensors (after training)

    # -- weights ---------------------------------------------------------------
    def set_tensors(self, tensors):
        """Position-matched like Keras ``load_weights`` (names are informative only)."""
        tensors = list(tensors)
        want = weight_specs()
        if len(tensors) != len(want):
            raise ValueError(f"expected {len(want)} weight tensors (28 FNet + 48 SRNet), got {len(tensors)}")
        new = []
        for (name, arr), (wname, wshape) in zip(tensors, want):
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            if arr.shape != tuple(wshape):
                raise ValueError(f"{name}: shape {arr.shape} does not match {wname} {tuple(wshape)}")
            grp = wname.split("/")[0]
            new.append((name if name.startswith(grp + "/") else wname, arr))
        self.tensors = new
        self._dirty = False
        if self._ctx is not None:
            self._ctx.load_weights(self.tensors)

    def refresh(self):
        """After train_on_batch the master weights live on the device: pull them before anyone reads."""
        if self._dirty and self._ctx is not None:
            self.tensors = self._ctx.get_weights()
        self._dirty = False

    def group(self, prefix):
        self.refresh()
        return [(n, a) for n, a in self.tensors if n.startswith(prefix + "/")]

    def set_group(self, prefix, arrays):
        cur = self.group(prefix)
        arrays = list(arrays)
        if len(arrays) != len(cur):
            raise ValueError(f"{prefix}: expected {len(cur)} arrays, got {len(arrays)}")
        repl = {n: np.asarray(a, np.float32) for (n, _), a in zip(cur, arrays)}
        self.set_tensors([(n, repl.get(n, a)) for n, a in self.tensors])

    # -- native context (lazy: building a model needs no GPU, computing does) ----
    def ctx(self, batch):
        if self._ctx is None or batch > self._ctx_batch:
            if self._ctx is not None:
                if self._ctx.train_iterations():
                    print('[WARN] the batch grew after training started: the native context is rebuilt, '
                          'weights are kept, Adam moments restart from zero')
                self.refresh()
                self._ctx.close()
            h, w, _ = self.lr_shape
            self._ctx = sr_hip.Context(h, w, max_batch=int(batch), dtype=self.dtype, model=self.model,
                                       device=self.device, pad=self.pad)
            self._ctx_batch = int(batch)
            self._ctx.load_weights(self.tensors)
        return self._ctx


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


class _Model:
    """The part of keras.models.Model the reference's call sites touch."""
    name = "model"

    def predict(self, x, batch_size=None, verbose=0):
        raise NotImplementedError

    def count_params(self):
        return sum(int(a.size) for a in self.get_weights())

    def get_weights(self):
        return []

    def set_weights(self, weights):
        if weights:
            raise ValueError(f"{self.name} has no weights")

    def load_weights(self, path):
        weights_io.load_into(self, path)

    def save_weights(self, path):
        weights_io.save_from(self, path)

    def summary(self):
        print(f"{self.name}: {self.count_params()} parameters")


class _CoreModel(_Model):
    def __init__(self, core, prefixes):
        self.core = core
        self._prefixes = prefixes

    def named_weights(self):
        return [(n, a) for p in self._prefixes for n, a in self.core.group(p)]

    def get_weights(self):
        return [a for _, a in self.named_weights()]

    def set_weights(self, weights):
        weights = list(weights)
        i = 0
        for p in self._prefixes:
            k = len(self.core.group(p))
            self.core.set_group(p, weights[i:i + k])
            i += k
        if i != len(weights):
            raise ValueError(f"{self.name}: expected {i} arrays, got {len(weights)}")


class FNetModel(_CoreModel):
    """[I_LR_t, I_LR_t_1] -> flow (B,h,w,2), channel 0 = d_row, 1 = d_col (reference layers.py:27-48)."""
    name = "FNet"

    def __init__(self, core):
        super().__init__(core, ["FNet"])

    def predict(self, x, batch_size=None, verbose=0):
        lr_t, lr_tm1 = _as_list(x)
        return self.core.ctx(len(lr_t)).fnet(lr_t, lr_tm1)


class SRNetModel(_CoreModel):
    """[I_LR_t (B,h,w,3), S_s (B,h,w,48)] -> (B,4h,4w,3) in [0,1] (reference layers.py:64-79)."""
    name = "SRNet"

    def __init__(self, core):
        super().__init__(core, ["SRNet"])

    def predict(self, x, batch_size=None, verbose=0):
        lr_t, s_s = _as_list(x)
        return self.core.ctx(len(lr_t)).srnet(lr_t, s_s)


class UpscalingModel(_Model):
    """UpSampling2D(size=4, 'bilinear') with TF1 legacy sampling; any channel count
    (the reference declares 3 input channels but feeds the 2-channel flow, layers.py:88 vs FRVSR.py:60)."""

    def __init__(self, idx, device=0):
        self.name = "Upscaling_" + str(idx)
        self.device = device

    def predict(self, x, batch_size=None, verbose=0):
        (img,) = _as_list(x) if isinstance(x, (list, tuple)) else (x,)
        return sr_hip.resize_bilinear(img, 4, device=self.device)


class Space2DepthModel(_Model):
    name = "Space_to_Depth"

    def __init__(self, device=0):
        self.device = device

    def predict(self, x, batch_size=None, verbose=0):
        (img,) = _as_list(x) if isinstance(x, (list, tuple)) else (x,)
        return sr_hip.space_to_depth(img, 4, device=self.device)


class WarpModel(_Model):
    """[image, flow] -> dense_image_warp(image, flow) (input order of reference layers.py:138)."""

    def __init__(self, idx, device=0):
        self.name = "Warp_" + str(idx)
        self.device = device

    def predict(self, x, batch_size=None, verbose=0):
        img, flow = _as_list(x)
        return sr_hip.warp(img, flow, device=self.device)


# ---- builder classes, same names and signatures as the reference --------------------------------
class FNet(object):
    def __init__(self, LR_shape, core=None):
        self.LR_shape = LR_shape
        self._core = core

    def build(self):
        return FNetModel(self._core or GeneratorCore(self.LR_shape))


class SRNet(object):
    def __init__(self, LR_shape, core=None):
        self.lr_shape = LR_shape
        self.Ss_shape = (LR_shape[0], LR_shape[1], LR_shape[2] * 16)
        self._core = core

    def build(self):
        return SRNetModel(self._core or GeneratorCore(self.lr_shape))


# the weight-free blocks take an optional device= (the GPU of this process' rank; the reference has one device)
class Upscaling(object):
    def __init__(self, LR_shape, idx=0, device=0):
        self.lr_shape = LR_shape
        self.idx = idx
        self.device = device

    def build(self):
        return UpscalingModel(self.idx, device=self.device)


class Space2Depth(object):
    def __init__(self, HR_shape, device=0):
        self.hr_shape = HR_shape
        self.device = device

    def build(self):
        if self.hr_shape[0] % 4 or self.hr_shape[1] % 4:
            raise ValueError("Space2Depth: HR height and width must be multiples of 4")
        return Space2DepthModel(device=self.device)


class Warp(object):
    def __init__(self, input_shape, idx, device=0):
        self.input_shape = input_shape
        self.idx = idx
        self.device = device

    def build(self):
        return WarpModel(self.idx, device=self.device)


class Adam(object):
    """keras.optimizers.Adam as the reference constructs it (FRVSR.py:69: ``Adam(lr=1e-4)``); Keras 2.2.4
    defaults for the rest.  The update itself runs on the device (sr_train_apply)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0., amsgrad=False):
        if decay or amsgrad:
            raise NotImplementedError('Adam decay / amsgrad are not used by the reference and not built')
        self.lr, self.beta_1, self.beta_2 = float(lr), float(beta_1), float(beta_2)
        self.epsilon = 1e-7 if epsilon is None else float(epsilon)      # K.epsilon()
        # optimizer.iterations: ONE counter per Adam instance, shared by every model compiled with it (the reference
        # compiles generator, discriminator and tecogan with the same instance, TecoGAN.py:111-131); the moments belong
        # to each model's own train function
        self.iterations = 0


class GeneratorModel(_CoreModel):
    """frvsr / generator: [I_LR_t, I_LR_t_1, I_est_t_1] -> [I_est_t, I_LR_est_t_1]
    (reference FRVSR.py:65-66, TecoGAN.py:78-79).  Inputs may be float64 (np.zeros default,
    FRVSR.py:233-234); they are cast to float32 like Keras does."""

    def __init__(self, core, name):
        super().__init__(core, ["FNet", "SRNet"])
        self.name = name

    def predict(self, x, batch_size=None, verbose=0):
        lr_t, lr_tm1, est_tm1 = _as_list(x)
        est, lr_warp = self.core.ctx(len(lr_t)).step(lr_t, lr_tm1, est_tm1, want_lr_warp=True)
        return [est, lr_warp]

    # -- training (reference FRVSR.py:69-71,156)
    optimizer = None

    def compile(self, loss, loss_weights=None, optimizer=None):
        """Only what the reference compiles (FRVSR.py:70-71): two mean-squared-error outputs weighted 1, 1."""
        losses = list(loss) if isinstance(loss, (list, tuple)) else [loss]
        weights = [1., 1.] if loss_weights is None else [float(v) for v in loss_weights]
        if losses != ['mean_squared_error'] * 2 or weights != [1., 1.]:
            raise NotImplementedError("the native training step implements loss=['mean_squared_error'] * 2 with "
                                      "loss_weights=[1., 1.] (reference FRVSR.py:70-71)")
        self.optimizer = optimizer if optimizer is not None else Adam()

    def train_on_batch(self, x, y):
        """[I_LR_t, I_LR_t_1, I_est_t_1], [I_HR_t, I_LR_t] -> [total, sr_loss, flow_loss] like Keras
        (FRVSR.py:154-156).  The flow loss compares the warped previous LR frame with I_LR_t itself, which
        is what the reference passes as the second target; any other target is refused.  Under
        torch.distributed (one process per GPU) the flat gradient is averaged over the ranks with one
        RCCL all-reduce before the Adam step, i.e. the job trains on the union of the ranks' batches."""
        if self.optimizer is None:
            raise RuntimeError('You must compile a model before training/testing. Use `model.compile(optimizer, loss)`.')
        lr_t, lr_tm1, est_tm1 = _as_list(x)
        hr_t, flow_target = _as_list(y)
        if flow_target is not lr_t and not np.array_equal(np.asarray(flow_target, np.float32), np.asarray(lr_t, np.float32)):
            raise NotImplementedError('the second target must be I_LR_t (reference FRVSR.py:155)')
        ctx = self.core.ctx(len(lr_t))
        opt = self.optimizer
        import dp
        if dp.active():
            dp.init_if_needed()
            if getattr(self, '_gbuf', None) is None:
                self._gbuf = dp.GradBuffer(ctx.param_count(), self.core.device)
            losses = ctx.train_grads(lr_t, lr_tm1, est_tm1, hr_t, d_grads=self._gbuf.ptr())
            losses = self._gbuf.reduce(losses, ctx)
            ctx.train_apply(d_grads=self._gbuf.ptr(), grad_scale=self._gbuf.grad_scale, lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon)
        else:
            losses = ctx.train_grads(lr_t, lr_tm1, est_tm1, hr_t)
            ctx.train_apply(lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon)
        opt.iterations += 1
        self.core._dirty = True
        return losses

    def train_on_clips(self, hr_clips):
        """One iteration of the reference's training loop body (FRVSR.py:103-156) from the HR clips alone:
        hr_clips [n_clips, c_frames, H, W, 3], uint8 frames or float32 in [0,1].  LR synthesis, previous
        frames / estimates (lock-step no-gradient recurrence) and the train_on_batch step all happen on the
        device (sr_train_clips_grads); returns [total, sr_loss, flow_loss].  Same data-parallel rule as
        train_on_batch."""
        if self.optimizer is None:
            raise RuntimeError('You must compile a model before training/testing. Use `model.compile(optimizer, loss)`.')
        hr_clips = np.asarray(hr_clips)
        ctx = self.core.ctx(hr_clips.shape[0] * hr_clips.shape[1])
        opt = self.optimizer
        import dp
        if dp.active():
            dp.init_if_needed()
            if getattr(self, '_gbuf', None) is None:
                self._gbuf = dp.GradBuffer(ctx.param_count(), self.core.device)
            losses = self._gbuf.reduce(ctx.train_clips_grads(hr_clips, d_grads=self._gbuf.ptr()), ctx)
            ctx.train_apply(d_grads=self._gbuf.ptr(), grad_scale=self._gbuf.grad_scale, lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon)
        else:
            losses = ctx.train_clips_grads(hr_clips)
            ctx.train_apply(lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon)
        opt.iterations += 1
        self.core._dirty = True
        return losses

    # device-resident recurrence (state never leaves the GPU); used by the eval drivers
    def clip_begin(self, batch):
        self._clip_ctx = self.core.ctx(batch)
        self._clip_ctx.clip_begin(batch)

    def clip_push(self, lr_t, want_output=True):
        ctx = getattr(self, "_clip_ctx", None)
        if ctx is None:
            raise RuntimeError("clip_push without clip_begin")
        if ctx is not self.core._ctx or not ctx._h:
            # a larger predict / train batch rebuilt the native context: the recurrent state went with the old one
            raise RuntimeError("the native context was rebuilt for a larger batch since clip_begin: "
                               "call clip_begin again (the recurrent state restarts from zero)")
        return ctx.clip_push(lr_t, want_output=want_output)


class HRWarpingModel(_CoreModel):
    """hr_warping: the same three inputs -> I_hat_est_t_1 (reference TecoGAN.py:83-85); single output array."""
    name = "HR_Warping"

    def __init__(self, core):
        super().__init__(core, ["FNet"])

    def predict(self, x, batch_size=None, verbose=0):
        lr_t, lr_tm1, est_tm1 = _as_list(x)
        return self.core.ctx(len(lr_t)).hr_warp(lr_t, lr_tm1, est_tm1)


# ---- TecoGAN's discriminator and combined model (reference layers.py:142-186, TecoGAN.py:87-131) ------------------
DISC_BLOCKS = [(64, 64), (64, 64), (64, 128), (128, 256)]           # Discriminator_Block filters, layers.py:170-173
DISC_LOSSES = ['binary_crossentropy'] + ['mean_squared_error'] * 4   # TecoGAN.py:116-119


def disc_weight_specs(hr_shape, first_conv=37):
    """[(keras_name, shape)] of the 28 tensors in Discriminator.save_weights order.  Numbering as in a process that built
    the generator first (FNet conv2d_1..14, SRNet conv2d_15..36), like reference TecoGAN.py:24-60; Keras matches by ORDER."""
    H, W, c = (int(v) for v in hr_shape)
    specs = [(f"conv2d_{first_conv}/kernel:0", (3, 3, 9 * c, 64)), (f"conv2d_{first_conv}/bias:0", (64,))]
    for k, (cin, cout) in enumerate(DISC_BLOCKS):
        specs += [(f"conv2d_{first_conv + 1 + k}/kernel:0", (4, 4, cin, cout)), (f"conv2d_{first_conv + 1 + k}/bias:0", (cout,))]
        specs += [(f"batch_normalization_{k + 1}/{n}:0", (cout,)) for n in ("gamma", "beta", "moving_mean", "moving_variance")]
    specs += [("dense_1/kernel:0", ((H // 16) * (W // 16) * 256, 1)), ("dense_1/bias:0", (1,))]
    return specs


def disc_layer_names(first_conv=37, first_input=12, concat=3, first_lrelu=14, activation=2):
    """`layer_names` of Discriminator.save_weights as Keras 2.2.4 lists them (model.layers: the nine inputs, then the layers by
    depth, reference layers.py:153-176), numbered as in a process that ran TecoGAN.build() up to the discriminator
    (TecoGAN.py:31-60: 11 inputs, 2 concatenates, 13 LeakyReLUs, 1 Activation before it).  Only the order matters to Keras."""
    names = [f"input_{first_input + i}" for i in range(9)] + [f"concatenate_{concat}", f"conv2d_{first_conv}", f"leaky_re_lu_{first_lrelu}"]
    for k in range(4):
        names += [f"conv2d_{first_conv + 1 + k}", f"batch_normalization_{k + 1}", f"leaky_re_lu_{first_lrelu + 1 + k}"]
    return names + ["flatten_1", "dense_1", f"activation_{activation}"]


class DiscriminatorModel(_Model):
    """9 HR images -> [probability (B,1), d_1..d_4] (reference layers.py:178-183).  predict runs BatchNormalization on the
    moving statistics, train_on_batch on the batch statistics (and moves the moving ones), like Keras."""
    name = "Discriminator"

    def __init__(self, hr_shape, dtype="fp32", device=0, seed=None):
        H, W, c = (int(v) for v in hr_shape)
        if c != 3:
            raise ValueError(f"the discriminator expects 3-channel images, got {c}")
        if H % 16 or W % 16:
            raise ValueError("Discriminator: HR height and width must be multiples of 16 (four stride-2 convolutions)")
        self.hr_shape, self.dtype, self.device = (H, W, c), dtype, device
        rng = np.random.default_rng(seed)
        self.tensors = []
        for name, shape in disc_weight_specs(self.hr_shape):
            if name.endswith("kernel:0"):                       # glorot_uniform, the Keras default of Conv2D and Dense
                fan_in = int(np.prod(shape[:-1]))
                fan_out = int(np.prod(shape[:2])) * shape[3] if len(shape) == 4 else shape[1]
                limit = np.sqrt(6.0 / (fan_in + fan_out))
                a = rng.uniform(-limit, limit, size=shape)
            elif name.endswith("gamma:0") or name.endswith("moving_variance:0"):
                a = np.ones(shape)
            else:
                a = np.zeros(shape)
            self.tensors.append((name, a.astype(np.float32)))
        self.trainable = True
        self.optimizer = None
        self._fresh_moments = True
        self._native = None
        self._native_batch = 0
        self._dirty = False
        self._gbuf = None

    # -- weights
    def refresh(self):
        if self._dirty and self._native is not None:
            self.tensors = self._native.get_weights()
        self._dirty = False

    def named_weights(self):
        self.refresh()
        return list(self.tensors)

    def get_weights(self):
        return [a for _, a in self.named_weights()]

    def set_weights(self, weights):
        weights = list(weights)
        if len(weights) != len(self.tensors):
            raise ValueError(f"{self.name}: expected {len(self.tensors)} arrays, got {len(weights)}")
        new = []
        for (n, cur), a in zip(self.tensors, weights):
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.shape != cur.shape:
                raise ValueError(f"{n}: shape {a.shape} does not match {cur.shape}")
            new.append((n, a))
        self.tensors, self._dirty = new, False
        if self._native is not None:
            self._native.load_weights(self.tensors)

    def load_weights(self, path):
        """Keras matches the file's weighted layers to the model's BY ORDER: the layer numbering of the saving process
        (conv2d_37 here, anything elsewhere) does not matter."""
        tensors = weights_io.read_tensors(path)
        if len(tensors) != len(self.tensors):
            raise ValueError(f"{path}: holds {len(tensors)} tensors, the model {self.name} expects {len(self.tensors)}")
        self.set_weights([a for _, a in tensors])

    def save_weights(self, path):
        weights_io.write_tensors(path, self.named_weights(), keep_layer_prefix=True, layer_order=disc_layer_names())

    # -- native object (lazy, sized by the largest batch seen)
    def native(self, batch):
        if self._native is None or batch > self._native_batch:
            if self._native is not None:
                print('[WARN] the discriminator batch grew: the native object is rebuilt, weights are kept, Adam moments restart')
                self.refresh()
                self._native.close()
            H, W, _ = self.hr_shape
            self._native = sr_hip.Discriminator(H, W, max_batch=int(batch), dtype=self.dtype, device=self.device)
            self._native_batch = int(batch)
            self._native.load_weights(self.tensors)
        return self._native

    def predict(self, x, batch_size=None, verbose=0):
        imgs = _as_list(x)
        return self.native(len(imgs[0])).predict(imgs)

    def compile(self, loss, loss_weights=None, optimizer=None):
        """What the reference compiles (TecoGAN.py:116-120, 436-440): binary cross-entropy on the probability, the four
        feature maps with weight 0.  Every compile gives the train function fresh Adam moments (the reference re-compiles
        each iteration); the iteration counter lives in the shared optimizer."""
        weights = [1., 0., 0., 0., 0.] if loss_weights is None else [float(v) for v in loss_weights]
        if list(loss) != DISC_LOSSES or weights != [1., 0., 0., 0., 0.]:
            raise NotImplementedError("the native discriminator step implements loss=['binary_crossentropy', 4 x "
                                      "'mean_squared_error'] with loss_weights=[1, 0, 0, 0, 0] (reference TecoGAN.py:116-120)")
        self.optimizer = optimizer if optimizer is not None else Adam()
        self._fresh_moments = True

    def train_on_batch(self, x, y):
        """[9 images], [labels, d_1..d_4] -> [total, bce, 0, 0, 0, 0]: the feature-map targets carry weight 0 and are not
        evaluated (Keras would report their mean squared errors; the reference discards the return value, TecoGAN.py:450)."""
        if self.optimizer is None:
            raise RuntimeError('You must compile a model before training/testing. Use `model.compile(optimizer, loss)`.')
        imgs = _as_list(x)
        labels = np.asarray(_as_list(y)[0], np.float32).reshape(-1)
        d = self.native(len(imgs[0]))
        opt = self.optimizer
        opt.iterations += 1
        kw = dict(lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon, slot=0, reset_moments=self._fresh_moments)
        import dp
        if dp.active():
            dp.init_if_needed()
            if self._gbuf is None:
                self._gbuf = dp.GradBuffer(d.param_count(), self.device)
            loss = d.train_grads(imgs, labels)
            d.read_grads(d_out=self._gbuf.ptr())
            loss = self._gbuf.reduce([loss, 0.0, 0.0])[0]
            d.apply(opt.iterations, d_grads=self._gbuf.ptr(), grad_scale=self._gbuf.grad_scale, **kw)
        else:
            loss = d.train_grads(imgs, labels)
            d.apply(opt.iterations, **kw)
        self._fresh_moments = False
        self._dirty = True
        return [loss, loss, 0.0, 0.0, 0.0, 0.0]


class Discriminator(object):
    def __init__(self, HR_shape, dtype="fp32", device=0):
        self.HR_shape = HR_shape
        self.dtype, self.device = dtype, device

    def build(self):
        return DiscriminatorModel(self.HR_shape, dtype=self.dtype, device=self.device)


# loss_weights of tecogan.compile (TecoGAN.py:127-130): GT, Dst, Ds1..Ds4, VGG22/34/44/54, ping-pong, warp
TECOGAN_LOSS_WEIGHTS = [1., -1e-3, 1.67e-3, 1.43e-3, 8.33e-4, 2e-5, 3e-5, 1.4e-6, 6e-6, 2e-3, 0.25, 1.]


class TecoGANModel(_Model):
    """tecogan: [I_LR_t, I_LR_t_1, I_est_t_1, I_hr_fut, I_hr_warp_fut, I_bic_t_1, I_bic_t, I_bic_t_fut] -> the 12 outputs of
    reference TecoGAN.py:101-108; only train_on_batch is built (the reference never predicts with it).  The four VGG19
    terms need ImageNet weights (keras.applications downloads them): their weights are taken as 0 and their losses
    reported as 0.  train_discriminator: the reference compiles tecogan while the discriminator is trainable
    (TecoGAN.py:122 before :461), so Keras 2.2.4 moves the discriminator's weights in this step too -- the default here;
    False freezes them, which is what the reference's author presumably intended."""
    name = "model_1"

    def __init__(self, gen, disc, train_discriminator=True):
        self.gen, self.disc = gen, disc
        self.train_discriminator = bool(train_discriminator)
        self.optimizer = None
        self.loss_weights = list(TECOGAN_LOSS_WEIGHTS)
        self._gbuf = None
        self.vgg_weights = None      # the 32 VGG19 tensors of the perceptual terms, or None (terms count as 0)
        self._vgg = None

    def set_vgg_weights(self, tensors):
        """The 32 arrays of keras.applications.VGG19(include_top=False) in Keras order (block1_conv1 kernel, bias, ...), or
        None: with them the combined step evaluates outputs 6..9 (reference TecoGAN.py:125,129), without them they count as 0."""
        if tensors is not None:
            tensors = [np.ascontiguousarray(a, np.float32) for a in tensors]
            if len(tensors) != 32:
                raise ValueError(f'VGG19 (include_top=False) has 32 weight tensors, got {len(tensors)}')
        if self._vgg is not None:
            self._vgg.close()
            self._vgg = None
        self.vgg_weights = tensors

    def attach_vgg(self, ctx, batch):
        """Make `ctx` evaluate the VGG19 terms (or not) according to vgg_weights; the native object is sized by the largest batch seen."""
        if self.vgg_weights is None:
            if getattr(ctx, '_vgg', None) is not None:
                ctx.set_vgg(None)
            return
        if self._vgg is None or batch > self._vgg.max_batch:
            if self._vgg is not None:
                if getattr(ctx, '_vgg', None) is self._vgg:
                    ctx.set_vgg(None)
                self._vgg.close()
            H, W, _ = self.gen.core.hr_shape
            self._vgg = sr_hip.Vgg19(H, W, max_batch=int(batch), dtype=self.gen.core.dtype, device=self.gen.core.device)
            self._vgg.load_weights(self.vgg_weights)
        if getattr(ctx, '_vgg', None) is not self._vgg:
            ctx.set_vgg(self._vgg)

    def named_weights(self):
        return self.gen.named_weights() + self.disc.named_weights()

    def get_weights(self):
        return [a for _, a in self.named_weights()]

    def compile(self, loss, loss_weights=None, optimizer=None):
        loss = list(loss)
        if len(loss) != 12 or [loss[i] for i in (0, 1, 2, 3, 4, 5, 10, 11)] != ['mean_squared_error', 'binary_crossentropy'] + ['mean_squared_error'] * 6:
            raise NotImplementedError('tecogan.compile: the native step implements the 12 losses of reference TecoGAN.py:122-126')
        lw = list(TECOGAN_LOSS_WEIGHTS) if loss_weights is None else [float(v) for v in loss_weights]
        if len(lw) != 12 or lw[11] != 1.0:
            raise NotImplementedError('tecogan.compile: 12 loss weights, the warp weight must be 1 (reference TecoGAN.py:127-130)')
        self.loss_weights = lw
        self.optimizer = optimizer if optimizer is not None else Adam()

    def train_on_batch(self, x, y):
        """-> [total, GT, Dst, Ds1..Ds4, VGG22, VGG34, VGG44, VGG54 (0 without vgg_weights), ping-pong, warp] (Keras order).
        Targets: y[1] must be ones, y[11] must be I_LR_t (reference TecoGAN.py:473-476); y[6..9] are y[0] in the reference
        (TecoGAN.py:474-478) and the native step uses y[0] for them."""
        if self.optimizer is None:
            raise RuntimeError('You must compile a model before training/testing. Use `model.compile(optimizer, loss)`.')
        lr_t, lr_tm1, est_tm1, next_est, next_hat_est, bic_past, bic_pres, bic_fut = _as_list(x)
        y = _as_list(y)
        if len(y) != 12:
            raise ValueError(f'tecogan.train_on_batch: expected 12 targets, got {len(y)}')
        if not np.all(np.asarray(y[1]) == 1):
            raise NotImplementedError('the adversarial target must be ones (reference TecoGAN.py:473)')
        if y[11] is not lr_t and not np.array_equal(np.asarray(y[11], np.float32), np.asarray(lr_t, np.float32)):
            raise NotImplementedError('the last target must be I_LR_t (reference TecoGAN.py:476)')
        B = len(lr_t)
        ctx = self.gen.core.ctx(B)
        d = self.disc.native(B)
        self.attach_vgg(ctx, B)
        opt = self.optimizer
        losses = sr_hip.tecogan_grads(ctx, d, lr_t, lr_tm1, est_tm1, next_est, next_hat_est, bic_past, bic_pres, bic_fut, y[0], y[2:6],
                                      y[10], self.loss_weights)
        opt.iterations += 1
        kw = dict(lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon, train_disc=self.train_discriminator)
        import dp
        if dp.active():
            dp.init_if_needed()
            ng, nd = ctx.param_count(), d.param_count()
            if self._gbuf is None:
                self._gbuf = dp.GradBuffer(ng + nd, self.gen.core.device, n_losses=13)
            sr_hip.tecogan_read_grads_dev(ctx, d, self._gbuf.ptr(), self._gbuf.ptr() + 4 * ng)
            losses = self._gbuf.reduce(losses)
            sr_hip.tecogan_apply(ctx, d, opt.iterations, d_gen_grads=self._gbuf.ptr(), d_disc_grads=self._gbuf.ptr() + 4 * ng,
                                 grad_scale=self._gbuf.grad_scale, **kw)
        else:
            sr_hip.tecogan_apply(ctx, d, opt.iterations, **kw)
        self.gen.core._dirty = True
        if self.train_discriminator:
            self.disc._dirty = True
        return losses
