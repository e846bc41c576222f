"""ctypes binding of libsr_hip.so (C ABI in include/sr_hip.h).

This is the only place Python touches the native library.  There is no CPU
fallback: if the library or a HIP device is missing, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsr_hip.so")

FP32, BF16, BF16X3 = 0, 1, 2
F16MX6 = 3
DTYPES = {"fp32": FP32, "float32": FP32, "bf16": BF16, "bfloat16": BF16, "bf16x3": BF16X3, "f16mx6": F16MX6, "mx6": F16MX6}
FRVSR, TECOGAN = 0, 1
ACT = {"none": 0, None: 0, "relu": 1, "lrelu": 2, "leaky_relu": 2, "tanh": 3, "clip01": 4}

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/sr_hip.h declares
SIGNATURES = {
    "sr_last_error": (C.c_char_p, [_vp]),
    "sr_device_count": (C.c_int, []),
    "sr_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sr_create_padded": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sr_destroy": (C.c_int, [_vp]),
    "sr_set_stream": (C.c_int, [_vp, _vp]),
    "sr_sync": (C.c_int, [_vp]),
    "sr_load_tensor": (C.c_int, [_vp, C.c_char_p, _vp, _i64p, C.c_int]),
    "sr_commit_weights": (C.c_int, [_vp]),
    "sr_get_tensor": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, _vp, _i64p, C.POINTER(C.c_int)]),
    "sr_step": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "sr_step_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "sr_step_hr_warp": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "sr_clip_begin": (C.c_int, [_vp, C.c_int]),
    "sr_clip_push": (C.c_int, [_vp, _vp, _vp]),
    "sr_clip_push_dev": (C.c_int, [_vp, _vp, _vp]),
    "sr_clip_push_u8": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "sr_u8_wait": (C.c_int, [_vp, C.c_int]),
    "sr_clip_read_est": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "sr_clip_read_est_begin": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "sr_clip_read_est_end": (C.c_int, [_vp]),
    "sr_host_alloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "sr_host_free": (C.c_int, [_vp]),
    "sr_fnet": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "sr_srnet": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "sr_profile_enable": (C.c_int, [_vp, C.c_int]),
    "sr_profile_count": (C.c_int, [_vp]),
    "sr_profile_read": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), _i64p,
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sr_profile_reset": (C.c_int, [_vp]),
    "sr_op_conv3x3": (C.c_int, [C.c_int] * 7 + [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "sr_op_convT3x3s2": (C.c_int, [C.c_int] * 7 + [_vp, _vp, _vp, C.c_int, _vp]),
    "sr_op_conv3x3_mx6": (C.c_int, [C.c_int] * 4 + [_vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp]),
    "sr_op_maxpool2": (C.c_int, [C.c_int] * 5 + [_vp, _vp]),
    "sr_op_resize_bilinear": (C.c_int, [C.c_int] * 6 + [_vp, _vp]),
    "sr_op_warp": (C.c_int, [C.c_int] * 5 + [_vp, _vp, _vp]),
    "sr_op_space_to_depth": (C.c_int, [C.c_int] * 6 + [_vp, _vp]),
    "sr_op_blur_cubic_downscale": (C.c_int, [C.c_int] * 5 + [_vp, _vp]),
    "sr_op_cubic_upscale4": (C.c_int, [C.c_int] * 5 + [_vp, _vp]),
    "sr_train_param_count": (C.c_int, [_vp, _i64p]),
    "sr_train_grads": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _f32p]),
    "sr_train_grads_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _f32p]),
    "sr_train_read_grads": (C.c_int, [_vp, _vp]),
    "sr_train_apply": (C.c_int, [_vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "sr_rccl_unique_id": (C.c_int, [_vp]),
    "sr_rccl_comm_create": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, C.c_int, C.c_int]),
    "sr_rccl_comm_destroy": (C.c_int, [_vp]),
    "sr_rccl_allreduce_f32": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "sr_train_allreduce": (C.c_int, [_vp, _vp, _vp]),
    "sr_train_on_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_float, _f32p]),
    "sr_train_iterations": (C.c_int64, [_vp]),
    "sr_train_clips_grads": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _f32p]),
    "sr_train_on_clips": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_float, _f32p]),
    "sr_op_conv3x3_bwd": (C.c_int, [C.c_int] * 7 + [_vp] * 6),
    "sr_op_convT3x3s2_bwd": (C.c_int, [C.c_int] * 7 + [_vp] * 6),
    "sr_op_maxpool2_bwd": (C.c_int, [C.c_int] * 5 + [_vp] * 3),
    "sr_op_resize_bilinear_bwd": (C.c_int, [C.c_int] * 6 + [_vp] * 2),
    "sr_op_warp_bwd_flow": (C.c_int, [C.c_int] * 5 + [_vp] * 4),
    "sr_disc_last_error": (C.c_char_p, [_vp]),
    "sr_disc_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sr_disc_destroy": (C.c_int, [_vp]),
    "sr_disc_sync": (C.c_int, [_vp]),
    "sr_disc_tensor_count": (C.c_int, []),
    "sr_disc_param_count": (C.c_int, [_vp, _i64p]),
    "sr_disc_set_tensor": (C.c_int, [_vp, C.c_int, _vp, _i64p, C.c_int]),
    "sr_disc_commit": (C.c_int, [_vp]),
    "sr_disc_get_tensor": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, _vp, _i64p, C.POINTER(C.c_int)]),
    "sr_disc_predict": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _vp, C.POINTER(_vp)]),
    "sr_disc_train_grads": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _vp, _f32p]),
    "sr_disc_read_grads": (C.c_int, [_vp, _vp, C.c_int]),
    "sr_disc_apply": (C.c_int, [_vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_int, C.c_int]),
    "sr_tecogan_grads": (C.c_int, [_vp, _vp, C.c_int] + [_vp] * 9 + [C.POINTER(_vp), _vp, _f32p, _f32p]),
    "sr_tecogan_apply": (C.c_int, [_vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_int]),
    "sr_tecogan_read_grads_dev": (C.c_int, [_vp, _vp, _vp, _vp]),
    "sr_tecogan_iteration": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_vp), _f32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64,
                                       C.c_int, _f32p, _f32p]),
    "sr_vgg_last_error": (C.c_char_p, [_vp]),
    "sr_vgg_tensor_count": (C.c_int, []),
    "sr_vgg_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sr_vgg_destroy": (C.c_int, [_vp]),
    "sr_vgg_set_tensor": (C.c_int, [_vp, C.c_int, _vp, _i64p, C.c_int]),
    "sr_vgg_commit": (C.c_int, [_vp]),
    "sr_vgg_loss_grads": (C.c_int, [_vp, C.c_int, _vp, _vp, _f32p, _f32p, _vp]),
    "sr_tecogan_set_vgg": (C.c_int, [_vp, _vp]),
    "sr_op_conv3x3_wgrad_noalloc": (C.c_int, [C.c_int] * 7 + [_vp] * 5),
    "sr_bench_conv": (C.c_int, [C.c_int] * 10 + [C.POINTER(C.c_float)]),
}

_lib = None


def lib():
    """Load libsr_hip.so once (loudly: a missing library is an error, never a fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C super-resolution_amd/csrc` (there is no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def _err(ctx):
    return lib().sr_last_error(ctx).decode("utf-8", "replace")


def _check(rc, ctx=None):
    if rc == 0:
        return
    msg = _err(ctx)
    # SR_ERR_ARG / SHAPE / WEIGHTS -> ValueError (Keras raises ValueError on shape mismatch); rest RuntimeError
    if rc in (-1, -2, -3):
        raise ValueError(msg)
    raise RuntimeError(msg)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(_vp)


def device_count() -> int:
    return lib().sr_device_count()


class PinnedArray:
    """A numpy view of page-locked host memory (sr_host_alloc): what sr_clip_push_u8 copies from / into asynchronously.
    `.a` is the array, `.ptr` its address; freed on close() / garbage collection (do not keep views of `.a` beyond that)."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(int(v) for v in shape), np.dtype(dtype)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = _vp()
        _check(lib().sr_host_alloc(C.byref(p), nbytes))
        self.ptr = p.value
        self.a = np.frombuffer((C.c_ubyte * nbytes).from_address(self.ptr), dtype=self.dtype).reshape(self.shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.a = None
            lib().sr_host_free(_vp(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One generator (FNet + SRNet + warps) for LR tiles of (h, w, 3) on one GPU."""

    def __init__(self, h, w, ch=3, scale=4, max_batch=1, dtype="fp32", model="frvsr", device=0, pad=False):
        """pad=True: h, w need not be multiples of 8 (sr_create_padded: zero padding inside, outputs cropped)."""
        self.h, self.w, self.max_batch = int(h), int(w), int(max_batch)
        self.dtype = DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
        self.model = {"frvsr": FRVSR, "tecogan": TECOGAN}[model] if isinstance(model, str) else int(model)
        self.device = int(device)
        self._h = _vp()
        create = lib().sr_create_padded if pad else lib().sr_create
        _check(create(C.byref(self._h), device, h, w, ch, scale, max_batch, self.dtype, self.model))

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            lib().sr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(lib().sr_set_stream(self._h, _vp(stream_ptr)), self._h)

    def sync(self):
        _check(lib().sr_sync(self._h), self._h)

    # -- weights
    def load_weights(self, tensors):
        """tensors: ordered mapping / iterable of (keras_name, array), file order."""
        items = tensors.items() if hasattr(tensors, "items") else tensors
        for name, arr in items:
            a = _f32(arr)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            _check(lib().sr_load_tensor(self._h, name.encode(), _ptr(a), shape, a.ndim), self._h)
        _check(lib().sr_commit_weights(self._h), self._h)

    def get_weights(self):
        out = []
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        rank = C.c_int()
        i = 0
        while lib().sr_get_tensor(self._h, i, name, 256, None, shape, C.byref(rank)) == 0:
            a = np.empty(tuple(shape[: rank.value]), np.float32)
            _check(lib().sr_get_tensor(self._h, i, None, 0, _ptr(a), None, None), self._h)
            out.append((name.value.decode(), a))
            i += 1
        return out

    # -- compute
    def _shapes(self, B):
        return (B, self.h, self.w, 3), (B, 4 * self.h, 4 * self.w, 3)

    def _check_in(self, a, shape, what):
        a = _f32(a)
        if a.shape != shape:
            raise ValueError(f"{what}: expected shape {shape}, got {a.shape}")
        return a

    def step(self, lr_t, lr_tm1, est_tm1, want_lr_warp=True):
        B = int(np.shape(lr_t)[0])
        ls, hs = self._shapes(B)
        lr_t, lr_tm1 = self._check_in(lr_t, ls, "I_LR_t"), self._check_in(lr_tm1, ls, "I_LR_t_1")
        est_tm1 = self._check_in(est_tm1, hs, "I_est_t_1")
        est = np.empty(hs, np.float32)
        lw = np.empty(ls, np.float32) if want_lr_warp else None
        _check(lib().sr_step(self._h, B, _ptr(lr_t), _ptr(lr_tm1), _ptr(est_tm1), _ptr(est),
                             _ptr(lw) if want_lr_warp else None), self._h)
        return est, lw

    def step_dev(self, B, lr_t, lr_tm1, est_tm1, est_t, lr_warp=None):
        """Device pointers (ints), asynchronous on the ctx stream."""
        _check(lib().sr_step_dev(self._h, B, _vp(lr_t), _vp(lr_tm1), _vp(est_tm1), _vp(est_t),
                                 _vp(lr_warp) if lr_warp else None), self._h)

    def hr_warp(self, lr_t, lr_tm1, est_tm1):
        B = int(np.shape(lr_t)[0])
        ls, hs = self._shapes(B)
        lr_t, lr_tm1 = self._check_in(lr_t, ls, "I_LR_t"), self._check_in(lr_tm1, ls, "I_LR_t_1")
        est_tm1 = self._check_in(est_tm1, hs, "I_est_t_1")
        out = np.empty(hs, np.float32)
        _check(lib().sr_step_hr_warp(self._h, B, _ptr(lr_t), _ptr(lr_tm1), _ptr(est_tm1), _ptr(out)), self._h)
        return out

    def set_vgg(self, vgg):
        """Attach (or, with None, detach) a Vgg19 to this TecoGAN context: the combined step then evaluates the four perceptual terms."""
        _check(lib().sr_tecogan_set_vgg(self._h, vgg._h if vgg is not None else None), self._h)
        self._vgg = vgg

    def clip_begin(self, B):
        self._clip_B = int(B)
        _check(lib().sr_clip_begin(self._h, B), self._h)

    def clip_push(self, lr_t, want_output=True):
        B = getattr(self, "_clip_B", 0)
        ls, hs = self._shapes(B)
        lr_t = self._check_in(lr_t, ls, "I_LR_t")
        out = np.empty(hs, np.float32) if want_output else None
        _check(lib().sr_clip_push(self._h, _ptr(lr_t), _ptr(out) if want_output else None), self._h)
        return out

    def clip_push_dev(self, lr_t, est_t=None):
        _check(lib().sr_clip_push_dev(self._h, _vp(lr_t), _vp(est_t) if est_t else None), self._h)

    def clip_push_u8(self, lr_t, out_u8, lr_on_device=False):
        """Asynchronous: lr_t / out_u8 are raw pointers (ints) -- a device or page-locked host LR frame batch and a
        (page-locked) host uint8 buffer of (B,4h,4w,3).  Complete after u8_wait(0) / sync()."""
        _check(lib().sr_clip_push_u8(self._h, _vp(lr_t), int(bool(lr_on_device)), _vp(out_u8)), self._h)

    def u8_wait(self, keep_in_flight=0):
        _check(lib().sr_u8_wait(self._h, int(keep_in_flight)), self._h)

    def clip_read_est(self, first=0, n=1, out=None):
        """float32 estimates (n,4h,4w,3) of clips first .. first+n-1 as the last clip_push* left them (synchronous).
        out: a float32 array of that shape to fill (page-locked memory -- PinnedArray.a -- makes the copy several times
        faster than into pageable memory); default a fresh array."""
        shape = (int(n), 4 * self.h, 4 * self.w, 3)
        if out is None:
            out = np.empty(shape, np.float32)
        elif out.shape != shape or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError(f"out: expected a contiguous float32 array of shape {shape}")
        _check(lib().sr_clip_read_est(self._h, int(first), int(n), _ptr(out)), self._h)
        return out

    def clip_read_est_begin(self, out, first=0, n=1):
        """Start the same read-back beside the next step (include/sr_hip.h sr_clip_read_est_begin).  out: a page-locked float32
        array (PinnedArray.a) of shape (n,4h,4w,3); valid after clip_read_est_end().  The caller keeps `out` alive until then."""
        shape = (int(n), 4 * self.h, 4 * self.w, 3)
        if out.shape != shape or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError(f"out: expected a contiguous float32 array of shape {shape}")
        _check(lib().sr_clip_read_est_begin(self._h, int(first), int(n), _ptr(out)), self._h)

    def clip_read_est_end(self):
        _check(lib().sr_clip_read_est_end(self._h), self._h)

    def fnet(self, lr_t, lr_tm1):
        B = int(np.shape(lr_t)[0])
        ls, _ = self._shapes(B)
        lr_t, lr_tm1 = self._check_in(lr_t, ls, "I_LR_t"), self._check_in(lr_tm1, ls, "I_LR_t_1")
        flow = np.empty((B, self.h, self.w, 2), np.float32)
        _check(lib().sr_fnet(self._h, B, _ptr(lr_t), _ptr(lr_tm1), _ptr(flow)), self._h)
        return flow

    def srnet(self, lr_t, s_s):
        B = int(np.shape(lr_t)[0])
        ls, hs = self._shapes(B)
        lr_t = self._check_in(lr_t, ls, "I_LR_t")
        s_s = self._check_in(s_s, (B, self.h, self.w, 48), "S_s")
        out = np.empty(hs, np.float32)
        _check(lib().sr_srnet(self._h, B, _ptr(lr_t), _ptr(s_s), _ptr(out)), self._h)
        return out

    # -- training (reference FRVSR.py:69-71,156)
    def param_count(self):
        n = C.c_int64()
        _check(lib().sr_train_param_count(self._h, C.byref(n)), self._h)
        return n.value

    def _train_inputs(self, lr_t, lr_tm1, est_tm1, hr_t):
        B = int(np.shape(lr_t)[0])
        ls, hs = self._shapes(B)
        return (B, self._check_in(lr_t, ls, "I_LR_t"), self._check_in(lr_tm1, ls, "I_LR_t_1"),
                self._check_in(est_tm1, hs, "I_est_t_1"), self._check_in(hr_t, hs, "I_HR_t"))

    def train_grads(self, lr_t, lr_tm1, est_tm1, hr_t, d_grads=None, read=False):
        """Forward + backward of one batch.  d_grads: optional DEVICE pointer (int) that receives the flat
        gradient (the all-reduce buffer).  Returns [total, sr_loss, flow_loss] (+ the flat gradient as a
        numpy array when read=True)."""
        B, a, b, c, d = self._train_inputs(lr_t, lr_tm1, est_tm1, hr_t)
        losses = (C.c_float * 3)()
        _check(lib().sr_train_grads(self._h, B, _ptr(a), _ptr(b), _ptr(c), _ptr(d),
                                    _vp(d_grads) if d_grads else None, losses), self._h)
        if not read:
            return list(losses)
        g = np.empty(self.param_count(), np.float32)
        _check(lib().sr_train_read_grads(self._h, _ptr(g)), self._h)
        return list(losses), g

    def train_read_grads(self):
        """The ctx's own flat gradient (of the last train_grads / train_clips_grads, or what train_allreduce left there)."""
        g = np.empty(self.param_count(), np.float32)
        _check(lib().sr_train_read_grads(self._h, _ptr(g)), self._h)
        return g

    def train_allreduce(self, comm, d_grads=None):
        """Sum of the flat gradient over the ranks of `comm` (RcclComm), in place, on this ctx's stream (include/sr_hip.h)."""
        _check(lib().sr_train_allreduce(self._h, comm.handle, _vp(d_grads) if d_grads else None), self._h)

    def train_apply(self, d_grads=None, grad_scale=1.0, lr=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        _check(lib().sr_train_apply(self._h, _vp(d_grads) if d_grads else None, grad_scale, lr, beta_1, beta_2, epsilon), self._h)

    def train_on_batch(self, lr_t, lr_tm1, est_tm1, hr_t, lr=1e-4):
        B, a, b, c, d = self._train_inputs(lr_t, lr_tm1, est_tm1, hr_t)
        losses = (C.c_float * 3)()
        _check(lib().sr_train_on_batch(self._h, B, _ptr(a), _ptr(b), _ptr(c), _ptr(d), lr, losses), self._h)
        return list(losses)

    def train_iterations(self):
        return int(lib().sr_train_iterations(self._h))

    def _clips_in(self, hr_clips):
        """[n_clips, c_frames, 4h, 4w, 3] uint8 (decoded frames) or float32 in [0,1] -> (array, n, c, is_u8)."""
        a = np.asarray(hr_clips)
        want = (4 * self.h, 4 * self.w, 3)
        if a.ndim != 5 or tuple(a.shape[2:]) != want:
            raise ValueError(f"HR clips: expected shape (n_clips, c_frames) + {want}, got {tuple(a.shape)}")
        if a.dtype != np.uint8:
            a = a.astype(np.float32, copy=False)
        a = np.ascontiguousarray(a)
        if a.shape[0] * a.shape[1] > self.max_batch:
            raise ValueError(f"n_clips * c_frames = {a.shape[0] * a.shape[1]} exceeds the ctx's max_batch {self.max_batch}")
        return a, int(a.shape[0]), int(a.shape[1]), int(a.dtype == np.uint8)

    def train_clips_grads(self, hr_clips, d_grads=None):
        """Forward + backward of one training iteration built on the device from HR clips (LR synthesis,
        no-gradient recurrence for the previous estimates; reference FRVSR.py:103-156)."""
        a, n, c, u8 = self._clips_in(hr_clips)
        losses = (C.c_float * 3)()
        _check(lib().sr_train_clips_grads(self._h, n, c, a.ctypes.data_as(C.c_void_p), u8,
                                          _vp(d_grads) if d_grads else None, losses), self._h)
        return list(losses)

    def train_on_clips(self, hr_clips, lr=1e-4):
        a, n, c, u8 = self._clips_in(hr_clips)
        losses = (C.c_float * 3)()
        _check(lib().sr_train_on_clips(self._h, n, c, a.ctypes.data_as(C.c_void_p), u8, lr, losses), self._h)
        return list(losses)

    def split_grads(self, flat):
        """flat gradient / parameter vector -> {keras name: array} in file order."""
        out, o = {}, 0
        for name, a in self.get_weights():
            out[name] = np.asarray(flat[o:o + a.size]).reshape(a.shape)
            o += a.size
        return out

    # -- profiling
    def profile_enable(self, on=True):
        _check(lib().sr_profile_enable(self._h, 1 if on else 0), self._h)

    def profile_reset(self):
        _check(lib().sr_profile_reset(self._h), self._h)

    def profile_read(self):
        """[{name, ms_total, launches, flops_per_launch, bytes_per_launch}]"""
        out = []
        name = C.create_string_buffer(128)
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        n = C.c_int64()
        for i in range(lib().sr_profile_count(self._h)):
            _check(lib().sr_profile_read(self._h, i, name, 128, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), self._h)
            out.append(dict(name=name.value.decode(), ms_total=ms.value, launches=n.value,
                            flops_per_launch=fl.value, bytes_per_launch=by.value))
        return out


def _ptr_array(arrays):
    arr = (_vp * len(arrays))()
    for i, a in enumerate(arrays):
        arr[i] = a.ctypes.data if a is not None else None
    return arr


class Discriminator:
    """TecoGAN's spatio-temporal discriminator (reference layers.py:142-186) for HR images of (H, W, 3) on one GPU."""

    FEAT_CH = (64, 64, 128, 256)

    def __init__(self, H, W, max_batch=1, dtype="fp32", device=0):
        self.H, self.W, self.max_batch, self.device = int(H), int(W), int(max_batch), int(device)
        self.dtype = DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
        if self.dtype == F16MX6:      # a generator dtype: these networks have no residual-block trunk, they run its bf16x3 kernels
            self.dtype = BF16X3
        self._h = _vp()
        rc = lib().sr_disc_create(C.byref(self._h), device, H, W, max_batch, self.dtype)
        if rc:
            self._h = None
            self._raise(rc)

    def _raise(self, rc):
        msg = lib().sr_disc_last_error(self._h).decode("utf-8", "replace")
        raise (ValueError if rc in (-1, -2, -3) else RuntimeError)(msg)

    def _check(self, rc):
        if rc:
            self._raise(rc)

    def close(self):
        if getattr(self, "_h", None):
            lib().sr_disc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(lib().sr_disc_sync(self._h))

    def feat_shapes(self, B):
        return [(B, self.H >> (k + 1), self.W >> (k + 1), c) for k, c in enumerate(self.FEAT_CH)]

    def load_weights(self, tensors):
        """tensors: ordered mapping / iterable of (keras_name, array), the 28 tensors in file order (matched by position)."""
        items = list(tensors.items() if hasattr(tensors, "items") else tensors)
        if len(items) != lib().sr_disc_tensor_count():
            raise ValueError(f"the discriminator has 28 weight tensors, got {len(items)}")
        for i, (_, arr) in enumerate(items):
            a = _f32(arr)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(lib().sr_disc_set_tensor(self._h, i, _ptr(a), shape, a.ndim))
        self._check(lib().sr_disc_commit(self._h))

    def get_weights(self):
        out = []
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        rank = C.c_int()
        for i in range(lib().sr_disc_tensor_count()):
            self._check(lib().sr_disc_get_tensor(self._h, i, name, 256, None, shape, C.byref(rank)))
            a = np.empty(tuple(shape[: rank.value]), np.float32)
            self._check(lib().sr_disc_get_tensor(self._h, i, None, 0, _ptr(a), None, None))
            out.append((name.value.decode(), a))
        return out

    def param_count(self):
        n = C.c_int64()
        self._check(lib().sr_disc_param_count(self._h, C.byref(n)))
        return n.value

    def _imgs(self, imgs):
        imgs = [_f32(a) for a in imgs]
        if len(imgs) != 9:
            raise ValueError(f"the discriminator takes 9 images, got {len(imgs)}")
        B = imgs[0].shape[0]
        for a in imgs:
            if a.shape != (B, self.H, self.W, 3):
                raise ValueError(f"expected images of shape {(B, self.H, self.W, 3)}, got {a.shape}")
        return B, imgs

    def predict(self, imgs):
        """[prob (B,1), d_1, d_2, d_3, d_4] like discriminator.predict (inference-mode BatchNormalization)."""
        B, imgs = self._imgs(imgs)
        prob = np.empty((B, 1), np.float32)
        feats = [np.empty(s, np.float32) for s in self.feat_shapes(B)]
        self._check(lib().sr_disc_predict(self._h, B, _ptr_array(imgs), _ptr(prob), _ptr_array(feats)))
        return [prob] + feats

    def train_grads(self, imgs, labels, read=False):
        B, imgs = self._imgs(imgs)
        lab = _f32(np.asarray(labels).reshape(-1))
        if lab.shape != (B,):
            raise ValueError(f"expected {B} labels, got {lab.shape}")
        loss = C.c_float()
        self._check(lib().sr_disc_train_grads(self._h, B, _ptr_array(imgs), _ptr(lab), C.byref(loss)))
        return (loss.value, self.read_grads()) if read else loss.value

    def read_grads(self, d_out=None):
        if d_out is not None:
            self._check(lib().sr_disc_read_grads(self._h, _vp(d_out), 1))
            return None
        g = np.empty(self.param_count(), np.float32)
        self._check(lib().sr_disc_read_grads(self._h, _ptr(g), 0))
        return g

    def apply(self, t, d_grads=None, grad_scale=1.0, lr=5e-5, beta_1=0.9, beta_2=0.999, epsilon=1e-7, slot=0, reset_moments=False):
        self._check(lib().sr_disc_apply(self._h, _vp(d_grads) if d_grads else None, grad_scale, lr, beta_1, beta_2, epsilon, int(t),
                                        int(slot), int(bool(reset_moments))))

    def split_grads(self, flat):
        out, o = {}, 0
        for name, a in self.get_weights():
            out[name] = np.asarray(flat[o:o + a.size]).reshape(a.shape)
            o += a.size
        return out


class Vgg19:
    """The VGG19 feature extractor of TecoGAN's four perceptual terms (reference TecoGAN.py:867-901) for HR images of
    (H, W, 3): weights from the caller (32 arrays in Keras order), losses and the gradient with respect to the estimate.
    Attach it to a TecoGAN Context with ``ctx.set_vgg(v)``; detach (``ctx.set_vgg(None)``) before closing it."""

    def __init__(self, H, W, max_batch=1, dtype="fp32", device=0):
        self.H, self.W, self.max_batch, self.device = int(H), int(W), int(max_batch), int(device)
        self.dtype = DTYPES[dtype] if isinstance(dtype, str) else int(dtype)
        if self.dtype == F16MX6:      # a generator dtype: these networks have no residual-block trunk, they run its bf16x3 kernels
            self.dtype = BF16X3
        self._h = _vp()
        rc = lib().sr_vgg_create(C.byref(self._h), device, H, W, max_batch, self.dtype)
        if rc:
            self._h = None
            self._raise(rc)

    def _raise(self, rc):
        msg = lib().sr_vgg_last_error(self._h).decode("utf-8", "replace")
        raise (ValueError if rc in (-1, -2, -3) else RuntimeError)(msg)

    def _check(self, rc):
        if rc:
            self._raise(rc)

    def close(self):
        if getattr(self, "_h", None):
            lib().sr_vgg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, tensors):
        """tensors: 32 arrays, or an ordered mapping / iterable of (name, array), in Keras' VGG19 order
        (block1_conv1 kernel, bias, ... block5_conv4 kernel, bias)."""
        items = list(tensors.values()) if hasattr(tensors, "values") else [t[1] if isinstance(t, tuple) else t for t in tensors]
        if len(items) != lib().sr_vgg_tensor_count():
            raise ValueError(f"VGG19 (include_top=False) has 32 weight tensors, got {len(items)}")
        for i, arr in enumerate(items):
            a = _f32(arr)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(lib().sr_vgg_set_tensor(self._h, i, _ptr(a), shape, a.ndim))
        self._check(lib().sr_vgg_commit(self._h))

    def loss_grads(self, y_true, y_pred, weights=(3e-5, 1.4e-6, 6e-6, 2e-3), want_grad=True):
        """-> ([vgg_22, vgg_34, vgg_44, vgg_54] unweighted, gradient of sum w_k loss_k w.r.t. y_pred or None)."""
        y_true, y_pred = _f32(y_true), _f32(y_pred)
        B = y_true.shape[0]
        if y_true.shape != (B, self.H, self.W, 3) or y_pred.shape != y_true.shape:
            raise ValueError(f"expected two arrays of shape {(B, self.H, self.W, 3)}")
        w = (C.c_float * 4)(*[float(x) for x in weights])
        losses = (C.c_float * 4)()
        g = np.empty_like(y_pred) if want_grad else None
        self._check(lib().sr_vgg_loss_grads(self._h, B, _ptr(y_true), _ptr(y_pred), w, losses, _ptr(g) if want_grad else None))
        return [float(x) for x in losses], g


def tecogan_grads(ctx, disc, lr_t, lr_tm1, est_tm1, next_est, next_hat_est, bic_past, bic_pres, bic_fut, hr_t, d_targets, pp_target,
                  loss_weights):
    """tecogan.train_on_batch up to the optimizer (include/sr_hip.h sr_tecogan_grads): returns the 13 losses."""
    B = int(np.shape(lr_t)[0])
    ls, hs = ctx._shapes(B)
    a = [ctx._check_in(lr_t, ls, "I_LR_t"), ctx._check_in(lr_tm1, ls, "I_LR_t_1")]
    names = ["I_est_t_1", "I_hr_fut", "I_hr_warp_fut", "I_bic_t_1", "I_bic_t", "I_bic_t_fut", "hr_t"]
    a += [ctx._check_in(x, hs, n) for x, n in zip((est_tm1, next_est, next_hat_est, bic_past, bic_pres, bic_fut, hr_t), names)]
    feats = [_f32(t) for t in d_targets]
    for t, s in zip(feats, disc.feat_shapes(B)):
        if t.shape != s:
            raise ValueError(f"feature-map target: expected {s}, got {t.shape}")
    pp = ctx._check_in(pp_target, hs, "ping-pong target")
    lw = _f32(np.asarray(loss_weights, np.float32))
    if lw.shape != (12,):
        raise ValueError("loss_weights: expected the 12 weights of tecogan.compile")
    losses = (C.c_float * 13)()
    _check(lib().sr_tecogan_grads(ctx._h, disc._h, B, *[_ptr(x) for x in a], _ptr_array(feats), _ptr(pp), lw.ctypes.data_as(_f32p),
                                  losses), ctx._h)
    return list(losses)


def tecogan_apply(ctx, disc, t, lr=5e-5, beta_1=0.9, beta_2=0.999, epsilon=1e-7, train_disc=True, d_gen_grads=None, d_disc_grads=None,
                  grad_scale=1.0):
    _check(lib().sr_tecogan_apply(ctx._h, disc._h, _vp(d_gen_grads) if d_gen_grads else None, _vp(d_disc_grads) if d_disc_grads else None,
                                  grad_scale, lr, beta_1, beta_2, epsilon, int(t), int(bool(train_disc))), ctx._h)


ITERATION_ARRAYS = ("lr", "prev_lr", "hr", "prev_hr", "next_hr", "prev_hat_hr", "next_hat_hr", "est", "prev_est", "next_est",
                    "prev_hat_est", "next_hat_est", "bic", "prev_bic", "next_bic")


def tecogan_iteration(ctx, disc, batch, loss_weights, t0, lr=5e-5, beta_1=0.9, beta_2=0.999, epsilon=1e-7, train_disc=True):
    """One pass of the adversarial loop body (include/sr_hip.h sr_tecogan_iteration); batch: dict with the ITERATION_ARRAYS keys.
    Returns (the 13 losses of the last combined step, the two discriminator losses)."""
    B = int(np.shape(batch["lr"])[0])
    ls, hs = ctx._shapes(B)
    arrs = [ctx._check_in(batch[k], ls if k in ("lr", "prev_lr") else hs, k) for k in ITERATION_ARRAYS]
    lw = _f32(np.asarray(loss_weights, np.float32))
    if lw.shape != (12,):
        raise ValueError("loss_weights: expected the 12 weights of tecogan.compile")
    losses, dl = (C.c_float * 13)(), (C.c_float * 2)()
    _check(lib().sr_tecogan_iteration(ctx._h, disc._h, B, _ptr_array(arrs), lw.ctypes.data_as(_f32p), lr, beta_1, beta_2, epsilon, int(t0),
                                      int(bool(train_disc)), losses, dl), ctx._h)
    return list(losses), list(dl)


def tecogan_read_grads_dev(ctx, disc, d_gen_out=None, d_disc_out=None):
    _check(lib().sr_tecogan_read_grads_dev(ctx._h, disc._h, _vp(d_gen_out) if d_gen_out else None, _vp(d_disc_out) if d_disc_out else None),
           ctx._h)


class RcclComm(object):
    """An RCCL communicator owned by libsr_hip (include/sr_hip.h sr_rccl_*).  id_bytes: the 128 bytes rank 0 got from
    RcclComm.unique_id(), carried to every rank by the caller."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        _check(lib().sr_rccl_unique_id(buf))
        return bytes(buf)

    def __init__(self, id_bytes, rank, nranks, device=0):
        if len(id_bytes) != 128:
            raise ValueError("the communicator id is 128 bytes")
        self.rank, self.nranks = int(rank), int(nranks)
        h = _vp()
        buf = (C.c_ubyte * 128).from_buffer_copy(id_bytes)
        _check(lib().sr_rccl_comm_create(C.byref(h), buf, self.rank, self.nranks, int(device)))
        self.handle = h

    def allreduce_f32(self, d_ptr, n, stream=None):
        _check(lib().sr_rccl_allreduce_f32(self.handle, _vp(d_ptr), int(n), _vp(stream) if stream else None))

    def close(self):
        if self.handle:
            lib().sr_rccl_comm_destroy(self.handle)
            self.handle = None


# ---------------------------------------------------------------- single ops
def conv3x3(x, w, b, act="none", res=None, dtype="fp32", device=0):
    x, w, b = _f32(x), _f32(w), _f32(b)
    B, H, W, Cin = x.shape
    Cout = w.shape[3]
    if w.shape != (3, 3, Cin, Cout) or b.shape != (Cout,):
        raise ValueError(f"kernel {w.shape} / bias {b.shape} do not match input channels {Cin}")
    out = np.empty((B, H, W, Cout), np.float32)
    r = _f32(res) if res is not None else None
    _check(lib().sr_op_conv3x3(device, DTYPES[dtype], B, H, W, Cin, Cout, _ptr(x), _ptr(w), _ptr(b), ACT[act],
                               _ptr(r) if r is not None else None, _ptr(out)))
    return out


def conv3x3_mx6(x, w=None, b=None, relu=False, res=None, mode=0, device=0):
    """One 64 -> 64 conv of a residual block in f16mx6 arithmetic (include/sr_hip.h sr_op_conv3x3_mx6).  mode 0: result read back from
    the mx6 format, 1: residual layer through split-planar records, 2 / 3: x through the format alone (stored value / FP6 image)."""
    x = _f32(x)
    B, H, W, Cc = x.shape
    if Cc != 64:
        raise ValueError(f"64 channels, got {Cc}")
    out = np.empty((B, H, W, 64), np.float32)
    w = _f32(w) if w is not None else None
    b = _f32(b) if b is not None else None
    r = _f32(res) if res is not None else None
    if w is not None and (w.shape != (3, 3, 64, 64) or b is None or b.shape != (64,)):
        raise ValueError("kernel (3, 3, 64, 64) and bias (64,)")
    _check(lib().sr_op_conv3x3_mx6(device, B, H, W, _ptr(x), _ptr(w) if w is not None else None, _ptr(b) if b is not None else None,
                                   1 if relu else 0, _ptr(r) if r is not None else None, mode, _ptr(out)))
    return out


def conv_transpose3x3_s2(x, w, b, act="none", dtype="fp32", device=0):
    x, w, b = _f32(x), _f32(w), _f32(b)
    B, H, W, Cin = x.shape
    Cout = w.shape[2]
    if w.shape != (3, 3, Cout, Cin) or b.shape != (Cout,):
        raise ValueError(f"kernel {w.shape} / bias {b.shape} do not match input channels {Cin}")
    out = np.empty((B, 2 * H, 2 * W, Cout), np.float32)
    _check(lib().sr_op_convT3x3s2(device, DTYPES[dtype], B, H, W, Cin, Cout, _ptr(x), _ptr(w), _ptr(b), ACT[act], _ptr(out)))
    return out


def maxpool2(x, device=0):
    x = _f32(x)
    B, H, W, Cc = x.shape
    out = np.empty((B, (H + 1) // 2, (W + 1) // 2, Cc), np.float32)
    _check(lib().sr_op_maxpool2(device, B, H, W, Cc, _ptr(x), _ptr(out)))
    return out


def resize_bilinear(x, s, device=0):
    x = _f32(x)
    B, H, W, Cc = x.shape
    out = np.empty((B, s * H, s * W, Cc), np.float32)
    _check(lib().sr_op_resize_bilinear(device, B, H, W, Cc, s, _ptr(x), _ptr(out)))
    return out


def warp(img, flow, device=0):
    img, flow = _f32(img), _f32(flow)
    B, H, W, Cc = img.shape
    if flow.shape != (B, H, W, 2):
        raise ValueError(f"flow shape {flow.shape} does not match image {img.shape}")
    out = np.empty_like(img)
    _check(lib().sr_op_warp(device, B, H, W, Cc, _ptr(img), _ptr(flow), _ptr(out)))
    return out


def space_to_depth(x, bs=4, device=0):
    x = _f32(x)
    B, H, W, Cc = x.shape
    if H % bs or W % bs:
        raise ValueError("space_to_depth: H and W must be multiples of the block size")
    out = np.empty((B, H // bs, W // bs, bs * bs * Cc), np.float32)
    _check(lib().sr_op_space_to_depth(device, B, H, W, Cc, bs, _ptr(x), _ptr(out)))
    return out


def blur_cubic_downscale(hr, device=0):
    """[N,H,W,C] -> [N,H/4,W/4,C]: the reference's GaussianBlur 5x5 + INTER_CUBIC x1/4 LR synthesis on the GPU
    (same arithmetic as utils.blur_cubic_downscale, one fused pass)."""
    hr = _f32(hr)
    N, H, W, Cc = hr.shape
    if H % 4 or W % 4:
        raise ValueError("HR height and width must be multiples of 4")
    out = np.empty((N, H // 4, W // 4, Cc), np.float32)
    _check(lib().sr_op_blur_cubic_downscale(device, N, H, W, Cc, _ptr(hr), _ptr(out)))
    return out


def cubic_upscale4(lr, device=0):
    """[N,h,w,C] -> [N,4h,4w,C]: cv2.resize(..., INTER_CUBIC) x4 on the GPU (same arithmetic as utils.cubic_upscale4)."""
    lr = _f32(lr)
    N, h, w, Cc = lr.shape
    out = np.empty((N, 4 * h, 4 * w, Cc), np.float32)
    _check(lib().sr_op_cubic_upscale4(device, N, h, w, Cc, _ptr(lr), _ptr(out)))
    return out


# ---------------------------------------------------------------- gradients of the single ops
def conv3x3_bwd(x, w, dy, dtype="fp32", want_dx=True, device=0):
    x, w, dy = _f32(x), _f32(w), _f32(dy)
    B, H, W, Cin = x.shape
    Cout = w.shape[3]
    if w.shape != (3, 3, Cin, Cout) or dy.shape != (B, H, W, Cout):
        raise ValueError(f"kernel {w.shape} / dy {dy.shape} do not match x {x.shape}")
    dx = np.empty_like(x) if want_dx else None
    dw, db = np.empty_like(w), np.empty(Cout, np.float32)
    _check(lib().sr_op_conv3x3_bwd(device, DTYPES[dtype], B, H, W, Cin, Cout, _ptr(x), _ptr(w), _ptr(dy),
                                   _ptr(dx) if want_dx else None, _ptr(dw), _ptr(db)))
    return dx, dw, db


def conv_transpose3x3_s2_bwd(x, w, dy, dtype="fp32", want_dx=True, device=0):
    x, w, dy = _f32(x), _f32(w), _f32(dy)
    B, H, W, Cin = x.shape
    Cout = w.shape[2]
    if w.shape != (3, 3, Cout, Cin) or dy.shape != (B, 2 * H, 2 * W, Cout):
        raise ValueError(f"kernel {w.shape} / dy {dy.shape} do not match x {x.shape}")
    dx = np.empty_like(x) if want_dx else None
    dw, db = np.empty_like(w), np.empty(Cout, np.float32)
    _check(lib().sr_op_convT3x3s2_bwd(device, DTYPES[dtype], B, H, W, Cin, Cout, _ptr(x), _ptr(w), _ptr(dy),
                                      _ptr(dx) if want_dx else None, _ptr(dw), _ptr(db)))
    return dx, dw, db


def maxpool2_bwd(x, gy, device=0):
    x, gy = _f32(x), _f32(gy)
    B, H, W, Cc = x.shape
    gx = np.empty_like(x)
    _check(lib().sr_op_maxpool2_bwd(device, B, H, W, Cc, _ptr(x), _ptr(gy), _ptr(gx)))
    return gx


def resize_bilinear_bwd(gy, s, device=0):
    gy = _f32(gy)
    B, Ho, Wo, Cc = gy.shape
    gx = np.empty((B, Ho // s, Wo // s, Cc), np.float32)
    _check(lib().sr_op_resize_bilinear_bwd(device, B, Ho // s, Wo // s, Cc, s, _ptr(gy), _ptr(gx)))
    return gx


def warp_bwd_flow(img, flow, g, device=0):
    img, flow, g = _f32(img), _f32(flow), _f32(g)
    B, H, W, Cc = img.shape
    out = np.empty((B, H, W, 2), np.float32)
    _check(lib().sr_op_warp_bwd_flow(device, B, H, W, Cc, _ptr(img), _ptr(flow), _ptr(g), _ptr(out)))
    return out


def conv3x3_wgrad_noalloc(B, H, W, Cin, Cout, d_x, d_dy, d_dw, d_db=None, dtype="fp32", stream=None, device=0):
    """Device pointers (ints); asynchronous on `stream`.  Allocates nothing (include/sr_hip.h)."""
    _check(lib().sr_op_conv3x3_wgrad_noalloc(device, DTYPES[dtype], B, H, W, Cin, Cout, _vp(d_x), _vp(d_dy), _vp(d_dw),
                                             _vp(d_db) if d_db else None, _vp(stream) if stream else None))


def bench_conv(B, H, W, Cin, Cout, dtype="bf16x3", transposed=False, residual=False, iters=20, device=0):
    """Milliseconds per launch of one conv layer on device-resident random data."""
    ms = C.c_float()
    _check(lib().sr_bench_conv(device, DTYPES[dtype], int(transposed), B, H, W, Cin, Cout, int(residual), iters,
                               C.byref(ms)))
    return ms.value
