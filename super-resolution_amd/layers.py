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
        if dp.world()[1] > 1:
            dp.init_if_needed()
            if getattr(self, '_gbuf', None) is None:
                self._gbuf = dp.GradBuffer(ctx.param_count(), self.core.device)
            losses = ctx.train_grads(lr_t, lr_tm1, est_tm1, hr_t, d_grads=self._gbuf.ptr())
            losses = self._gbuf.reduce(losses)
            ctx.train_apply(d_grads=self._gbuf.ptr(), lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon)
        else:
            losses = ctx.train_grads(lr_t, lr_tm1, est_tm1, hr_t)
            ctx.train_apply(lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon)
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
        if dp.world()[1] > 1:
            dp.init_if_needed()
            if getattr(self, '_gbuf', None) is None:
                self._gbuf = dp.GradBuffer(ctx.param_count(), self.core.device)
            losses = self._gbuf.reduce(ctx.train_clips_grads(hr_clips, d_grads=self._gbuf.ptr()))
            ctx.train_apply(d_grads=self._gbuf.ptr(), lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon)
        else:
            losses = ctx.train_clips_grads(hr_clips)
            ctx.train_apply(lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon)
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
