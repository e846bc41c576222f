import numpy as np
def step_data(seed, B, h, w):
    rng = np.random.default_rng(seed)
    lr_t = rng.random((B, h, w, 3), dtype=np.float32)
    lr_tm1 = np.clip(lr_t + 0.1 * rng.standard_normal((B, h, w, 3)).astype(np.float32), 0, 1)
    est = rng.random((B, 4 * h, 4 * w, 3), dtype=np.float32)
    hr = rng.random((B, 4 * h, 4 * w, 3), dtype=np.float32)
    return lr_t, lr_tm1, est, hr
