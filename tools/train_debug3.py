import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import sr_hip
from oracle import ref_train, ref_numpy as rn
from train_debug_data import step_data
W = rn.make_weights(seed=7)
for (B, h, w) in [(3, 16, 16), (2, 32, 8), (3, 32, 16), (4, 32, 8), (3, 32, 8), (3, 8, 32), (5, 8, 8), (1, 8, 8)]:
    data = step_data(100 + h, B, h, w)
    ctx = sr_hip.Context(h, w, max_batch=B, dtype="fp32")
    ctx.load_weights(W)
    losses, flat = ctx.train_grads(*data, read=True)
    _, flat2 = ctx.train_grads(*data, read=True)
    want_l, want, _ = ref_train.loss_and_grads(W, *data)
    got = ctx.split_grads(flat)
    errs = {n: np.abs(got[n] - g).max() / np.abs(g).max() for n, g in want.items()}
    bad = [n.split('/')[1] for n, e in errs.items() if e > 1e-4 and 'kernel' in n]
    print((B, h, w), "worst %.1e" % max(errs.values()), "deterministic", np.array_equal(flat, flat2), "bad:", bad)
    ctx.close()
