"""CPU oracle for the FRVSR training step (reference FRVSR.py:69-71,154-156): torch autograd over
oracle/ref_torch.py's differentiable restatement of the generator, Keras 2.2.4's loss reduction and
its Adam update.

TEST INFRASTRUCTURE ONLY (rules and the "PARITY UNPINNED" statement of oracle/ref_numpy.py apply:
Keras 2.2.4 / TensorFlow 1.13.1 cannot run here, so the reference's own gradients are not available).

Semantics restated from Keras 2.2.4:
* ``compile(loss=['mean_squared_error'] * 2, loss_weights=[1., 1.])`` (FRVSR.py:70-71): each output's
  loss is ``mean(square(y_pred - y_true), axis=-1)`` averaged over every remaining axis and the batch,
  i.e. the mean over all elements; total = 1 * sr_loss + 1 * flow_loss.  ``train_on_batch`` returns
  ``[total, sr_loss, flow_loss]`` (FRVSR.py:161 prints them in that order).
* ``Adam(lr=1e-4)`` (FRVSR.py:69; keras/optimizers.py Adam.get_updates): beta_1 0.9, beta_2 0.999,
  epsilon 1e-7 (K.epsilon()), no decay, no amsgrad; with t = iterations + 1
      lr_t = lr * sqrt(1 - beta_2**t) / (1 - beta_1**t)
      m = beta_1 * m + (1 - beta_1) * g ;  v = beta_2 * v + (1 - beta_2) * g*g
      p = p - lr_t * m / (sqrt(v) + epsilon)
* gradients flow to FNet through both outputs: the LR warp (flow loss) and the HR warp of the constant
  previous estimate (dense_image_warp is differentiated with respect to the flow only: its floor / clamp
  index arithmetic has zero gradient, the clamped interpolation weights pass it inside [0, 1]).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ref_torch as rt


def _leaf_weights(weights, dtype):
    tw = rt.TorchWeights(weights, dtype)
    for g in tw.groups:
        tw.groups[g] = [(k.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, b in tw.groups[g]]
    return tw


def loss_and_grads(weights, lr_t, lr_tm1, est_tm1, hr_t, model="frvsr", dtype=torch.float64):
    """NHWC numpy in.  Returns (losses [total, sr, flow], grads {keras name: ndarray in the file's layout},
    outputs (est_t, lr_warp) NHWC)."""
    tw = _leaf_weights(weights, dtype)
    a, b, c, hr = (rt.to_nchw(v, dtype) for v in (lr_t, lr_tm1, est_tm1, hr_t))
    est, lw = rt.generator_step_t(tw, a, b, c, model)
    sr_loss = torch.mean((est - hr) ** 2)
    flow_loss = torch.mean((lw - a) ** 2)
    total = sr_loss + flow_loss
    total.backward()
    names = list(weights.keys())
    grads = {}
    for g in ("FNet", "SRNet"):
        gn = [n for n in names if n.startswith(g + "/")]
        for i, (k, bias) in enumerate(tw.groups[g]):
            grads[gn[2 * i]] = k.grad.numpy().astype(np.float64)
            grads[gn[2 * i + 1]] = bias.grad.numpy().astype(np.float64)
    return ([total.item(), sr_loss.item(), flow_loss.item()], grads,
            (rt.to_nhwc(est.detach()), rt.to_nhwc(lw.detach())))


class KerasAdam:
    """keras.optimizers.Adam (2.2.4) on a dict of numpy arrays."""

    def __init__(self, lr=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, params, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        out = {}
        for n, p in params.items():
            g = np.asarray(grads[n], np.float64)
            m = self.m.get(n, 0.0) * self.b1 + (1.0 - self.b1) * g
            v = self.v.get(n, 0.0) * self.b2 + (1.0 - self.b2) * g * g
            self.m[n], self.v[n] = m, v
            out[n] = (np.asarray(p, np.float64) - lr_t * m / (np.sqrt(v) + self.eps)).astype(np.float32)
        return out


def train_on_batch(weights, opt, lr_t, lr_tm1, est_tm1, hr_t, model="frvsr", dtype=torch.float64):
    """One reference train_on_batch: returns (new weights, [total, sr, flow])."""
    losses, grads, _ = loss_and_grads(weights, lr_t, lr_tm1, est_tm1, hr_t, model, dtype)
    return opt.step(weights, grads), losses
