"""Per-tensor gradient error of sr_train_grads against the autograd oracle (diagnostic)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import sr_hip
from oracle import ref_train, ref_numpy as rn

def step_data(seed, B, h, w):
    rng = np.random.default_rng(seed)
    lr_t = rng.random((B, h, w, 3), dtype=np.float32)
    lr_tm1 = np.clip(lr_t + 0.1 * rng.standard_normal((B, h, w, 3)).astype(np.float32), 0, 1)
    est = rng.random((B, 4 * h, 4 * w, 3), dtype=np.float32)
    hr = rng.random((B, 4 * h, 4 * w, 3), dtype=np.float32)
    return lr_t, lr_tm1, est, hr

B, h, w, dtype, sub = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
W = rn.make_weights(seed=7)
data = step_data(100 + h, B, h, w)
ctx = sr_hip.Context(h, w, max_batch=B, dtype=dtype)
ctx.load_weights(W)
if sub:
    ctx.train_grads(*data)
    data = tuple(a[:sub] for a in data)
losses, flat = ctx.train_grads(*data, read=True)
want_l, want, _ = ref_train.loss_and_grads(W, *data)
print("losses", losses, want_l)
got = ctx.split_grads(flat)
for n, g in want.items():
    e = np.abs(got[n] - g).max() / np.abs(g).max()
    print(f"{n:40s} {e:.2e}  max|g| {np.abs(g).max():.2e}" + ("   <<<<" if e > 1e-3 else ""))
