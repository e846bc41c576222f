"""CPU oracle #2 (torch-CPU functional ops) for the FRVSR / TecoGAN generator step.

TEST INFRASTRUCTURE ONLY (see oracle/ref_numpy.py for the rules and for the
"PARITY UNPINNED" statement, which applies to this file equally).

Independent of oracle/ref_numpy.py on purpose: it works in NCHW on
torch.nn.functional primitives (oneDNN convolutions, conv_transpose2d with the
"full then crop" alignment of SURVEY.md A.5, ceil-mode max pooling,
pixel_unshuffle + channel permutation for space_to_depth, hand-built index
tables for the TF1 legacy bilinear resize and for dense_image_warp).  The two
restatements must agree to <=1e-5 (tests/test_oracle_cross.py).  It is also the
timed "cpu_baseline" (kind "port") in bench.py, because the literal reference
path (Keras 2.2.4 on TensorFlow 1.13.1) cannot run here or on the GPU box.

Follows: reference layers.py:15-140, FRVSR.py:55-66, TecoGAN.py:65-85,
FRVSR.py:231-234,293,309.  Public API is NHWC numpy in / NHWC numpy out.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _t(x, dtype):
    return torch.as_tensor(np.ascontiguousarray(x)).to(dtype)


def to_nchw(x, dtype=torch.float32):
    return _t(x, dtype).permute(0, 3, 1, 2).contiguous()


def to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().numpy()


class TorchWeights:
    """Keras-named weights converted once to torch layouts (by position)."""

    def __init__(self, weights, dtype=torch.float32):
        self.dtype = dtype
        self.groups = {"FNet": [], "SRNet": []}
        names = list(weights.keys())
        for g in self.groups:
            gn = [n for n in names if n.startswith(g + "/")]
            for i in range(0, len(gn), 2):
                k = _t(weights[gn[i]], dtype)
                b = _t(weights[gn[i + 1]], dtype)
                self.groups[g].append((k, b))

    def conv(self, g, i):
        k, b = self.groups[g][i]          # HWIO -> OIHW
        return k.permute(3, 2, 0, 1).contiguous(), b

    def convt(self, g, i):
        k, b = self.groups[g][i]          # (kh,kw,out,in) -> (in,out,kh,kw)
        return k.permute(3, 2, 0, 1).contiguous(), b


def conv3x3_same(x, w, b):
    return F.conv2d(x, w, b, stride=1, padding=1)


def conv_transpose3x3_s2_same(x, w, b):
    H, W = x.shape[-2:]
    full = F.conv_transpose2d(x, w, None, stride=2, padding=0)   # (2H+1, 2W+1)
    return full[..., :2 * H, :2 * W] + b.view(1, -1, 1, 1)


def maxpool2_same(x):
    return F.max_pool2d(x, kernel_size=2, stride=2, ceil_mode=True)


def _legacy_table(n, s, dtype):
    dst = torch.arange(n * s)
    src = dst.to(dtype) / s
    lo = torch.floor(src).long()
    hi = torch.clamp(lo + 1, max=n - 1)
    t = src - lo.to(dtype)
    return lo, hi, t


def resize_bilinear_legacy(x, s):
    H, W = x.shape[-2:]
    ylo, yhi, ty = _legacy_table(H, s, x.dtype)
    xlo, xhi, tx = _legacy_table(W, s, x.dtype)
    r_lo = x.index_select(2, ylo)
    r_hi = x.index_select(2, yhi)
    tl, tr = r_lo.index_select(3, xlo), r_lo.index_select(3, xhi)
    bl, br = r_hi.index_select(3, xlo), r_hi.index_select(3, xhi)
    tx = tx.view(1, 1, 1, -1)
    ty = ty.view(1, 1, -1, 1)
    top = tl + (tr - tl) * tx
    bot = bl + (br - bl) * tx
    return top + (bot - top) * ty


def dense_image_warp(img, flow):
    """img [B,C,H,W], flow [B,2,H,W] (channel 0 = d_row, 1 = d_col)."""
    B, C, H, W = img.shape
    dt = img.dtype
    gy = torch.arange(H, dtype=dt).view(1, H, 1)
    gx = torch.arange(W, dtype=dt).view(1, 1, W)
    qy = gy - flow[:, 0]
    qx = gx - flow[:, 1]
    fy = torch.clamp(torch.floor(qy), 0, H - 2)
    fx = torch.clamp(torch.floor(qx), 0, W - 2)
    ay = torch.clamp(qy - fy, 0, 1).unsqueeze(1)
    ax = torch.clamp(qx - fx, 0, 1).unsqueeze(1)
    fy = fy.long()
    fx = fx.long()
    flat = img.reshape(B, C, H * W)

    def gather(yy, xx):
        idx = (yy * W + xx).view(B, 1, H * W).expand(B, C, H * W)
        return flat.gather(2, idx).view(B, C, H, W)

    tl = gather(fy, fx)
    tr = gather(fy, fx + 1)
    bl = gather(fy + 1, fx)
    br = gather(fy + 1, fx + 1)
    top = ax * (tr - tl) + tl
    bot = ax * (br - bl) + bl
    return ay * (bot - top) + top


def space_to_depth(x, bs=4):
    """NCHW in; output channel = (dy*bs+dx)*C + c (TF order), not torch's c*bs*bs+..."""
    B, C, H, W = x.shape
    y = F.pixel_unshuffle(x, bs)                       # channel = c*bs*bs + dy*bs + dx
    y = y.view(B, C, bs * bs, H // bs, W // bs).permute(0, 2, 1, 3, 4)
    return y.reshape(B, bs * bs * C, H // bs, W // bs)


def fnet(tw, lr_t, lr_tm1):
    k = 0

    def block(x):
        nonlocal k
        for _ in range(2):
            w, b = tw.conv("FNet", k)
            k += 1
            x = F.leaky_relu(conv3x3_same(x, w, b), 0.2)
        return x

    x = torch.cat([lr_t, lr_tm1], dim=1)
    x = maxpool2_same(block(x))
    x = maxpool2_same(block(x))
    x = maxpool2_same(block(x))
    x = resize_bilinear_legacy(block(x), 2)
    x = resize_bilinear_legacy(block(x), 2)
    x = resize_bilinear_legacy(block(x), 2)
    w, b = tw.conv("FNet", 12)
    x = F.leaky_relu(conv3x3_same(x, w, b), 0.2)
    w, b = tw.conv("FNet", 13)
    return torch.tanh(conv3x3_same(x, w, b))


def srnet(tw, lr_t, s_s):
    x = torch.cat([lr_t, s_s], dim=1)
    w, b = tw.conv("SRNet", 0)
    x = F.relu(conv3x3_same(x, w, b))
    for r in range(10):
        w1, b1 = tw.conv("SRNet", 1 + 2 * r)
        w2, b2 = tw.conv("SRNet", 2 + 2 * r)
        x = x + conv3x3_same(F.relu(conv3x3_same(x, w1, b1)), w2, b2)
    w, b = tw.convt("SRNet", 21)
    x = F.relu(conv_transpose3x3_s2_same(x, w, b))
    w, b = tw.convt("SRNet", 22)
    x = F.relu(conv_transpose3x3_s2_same(x, w, b))
    w, b = tw.conv("SRNet", 23)
    return torch.clamp(conv3x3_same(x, w, b), 0, 1)


def generator_step_t(tw, lr_t, lr_tm1, est_tm1, model="frvsr", taps=None):
    """NCHW torch tensors in/out."""
    f_lr = fnet(tw, lr_t, lr_tm1)
    lr_warp = dense_image_warp(lr_tm1, f_lr)
    f_hr = resize_bilinear_legacy(f_lr, 4)
    hr_warp = dense_image_warp(est_tm1, f_hr)
    s_s = space_to_depth(hr_warp, 4)
    out = srnet(tw, lr_t, s_s)
    if model == "tecogan":
        out = torch.tanh(out + resize_bilinear_legacy(lr_t, 4))
    if taps is not None:
        taps.update(f_lr=f_lr, lr_warp=lr_warp, f_hr=f_hr, hr_warp=hr_warp, s_s=s_s)
    return out, lr_warp


def generator_step(weights, lr_t, lr_tm1, est_tm1, model="frvsr", dtype=torch.float32, taps=None):
    """NHWC numpy in / out; same contract as ref_numpy.generator_step."""
    tw = weights if isinstance(weights, TorchWeights) else TorchWeights(weights, dtype)
    with torch.no_grad():
        tt = {} if taps is not None else None
        out, lw = generator_step_t(tw, to_nchw(lr_t, tw.dtype), to_nchw(lr_tm1, tw.dtype),
                                   to_nchw(est_tm1, tw.dtype), model, tt)
        if taps is not None:
            taps.update({k: to_nhwc(v) for k, v in tt.items()})
    return to_nhwc(out), to_nhwc(lw)


def run_clip(weights, lr_frames, model="frvsr", dtype=torch.float32):
    """[T,B,h,w,3] -> [T,B,4h,4w,3]; zero initial state (reference FRVSR.py:231-234)."""
    tw = weights if isinstance(weights, TorchWeights) else TorchWeights(weights, dtype)
    T, B, h, w, c = lr_frames.shape
    outs = []
    with torch.no_grad():
        prev_lr = torch.zeros(B, c, h, w, dtype=tw.dtype)
        est = torch.zeros(B, c, 4 * h, 4 * w, dtype=tw.dtype)
        for t in range(T):
            cur = to_nchw(lr_frames[t], tw.dtype)
            est, _ = generator_step_t(tw, cur, prev_lr, est, model)
            prev_lr = cur
            outs.append(to_nhwc(est))
    return np.stack(outs)
