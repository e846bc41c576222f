"""Times whole adversarial iterations of stage.train through the facade (TecoGAN.gan_iteration: ping-pong batch from HR
clips + the five model steps), reference config: batch_size 4 clips x c_frames 10 at 128x128 HR."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import synth  # noqa: E402
from TecoGAN import TecoGAN  # noqa: E402


def main():
    n, c, H = 4, 10, 128
    cfg = {"root_dir": "/tmp", "dtype": "bf16x3", "data": {"rows": 1, "cols": 1, "videos": "/tmp/none"},
           "cnn": {"model": "tecogan", "pretrained_model": None}, "train": {"batch_size": n, "c_frames": c}}
    t = TecoGAN((H // 4, H // 4, 3), (H, H, 3), "/tmp/gan_out", cfg)
    t.build()
    hr = np.stack([synth.synth_clip(50 + k, c, H, H, scale=4) for k in range(n)]).astype(np.float32)
    t.step_by_step = "--step-by-step" in sys.argv
    for it in range(4):
        t0 = time.perf_counter()
        b = t.make_gan_batch(hr)
        t1 = time.perf_counter()
        t.make_gan_batch = lambda _hr, b=b: b
        loss = t.gan_iteration(hr)
        del t.make_gan_batch
        t2 = time.perf_counter()
        print(f"iteration {it}: batch {1e3 * (t1 - t0):.1f} ms, model steps {1e3 * (t2 - t1):.1f} ms, total loss {loss[0]:.5f}", flush=True)


if __name__ == "__main__":
    main()
