#!/usr/bin/env python3
"""`python main.py --yaml_file options/180p.yaml` on 16 synthetic 720p clips: the product CLI's eval stage
against the rate bench.py's `end_to_end` leg measures for the same work (LR frames in from page-locked memory, uint8 720p
frames out, padded 180-row context, 16 clips in lock-step).

    python tools/eval_bench.py [--clips 16] [--frames 20] [--keep DIR]

Prints one JSON line: frames/s of the eval loop as main.py reports it (FrameRecurrentVideoSR.eval_fps: time inside the
lock-step loop -- staging, push, uint8 copies, the float read-back of the metric frames -- excluding clip loading, LR
synthesis and the host-side PSNR / SSIM arithmetic, like the reference's `inference_time` bracket, FRVSR.py:292-296,316-317),
the whole stage's wall time, and a bare end_to_end loop on the same context size for comparison."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "super-resolution_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=16)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--keep", default=None, help="work directory to keep (default: a temporary one, removed)")
    args = ap.parse_args()
    import yaml
    import synth
    import weights_io
    import main as sr_main
    from FRVSR import FrameRecurrentVideoSR

    work = args.keep or tempfile.mkdtemp(prefix="sr_eval_")
    os.makedirs(os.path.join(work, "clips"), exist_ok=True)
    for c in range(args.clips):
        hr = synth.synth_clip(c, args.frames, 720, 1280, scale=4)
        np.save(os.path.join(work, "clips", "clip%02d.npy" % c), (hr * 255).astype(np.uint8))
    weights_io.write_tensors(os.path.join(work, "w.npz"), list(synth.make_weights(7).items()))
    cfg = yaml.safe_load(open(os.path.join(PKG, "options", "180p.yaml")))
    cfg["root_dir"] = work
    cfg["data"]["videos"] = os.path.join(work, "clips")
    cfg["cnn"]["pretrained_model"] = "w.npz"
    cfg["eval"] = {"output_dir": os.path.join(work, "eval_out"), "clips_in_flight": args.clips}
    yml = os.path.join(work, "180p.yaml")
    yaml.dump(cfg, open(yml, "w"))

    # the CLI itself, in-process so that the model object (eval_fps) can be read afterwards
    holder = {}
    orig_eval = FrameRecurrentVideoSR.eval

    def eval_and_keep(self):
        holder["m"] = self
        return orig_eval(self)

    FrameRecurrentVideoSR.eval = eval_and_keep
    t0 = time.perf_counter()
    sr_main.main(["--yaml_file", yml])
    wall = time.perf_counter() - t0
    FrameRecurrentVideoSR.eval = orig_eval
    m = holder["m"]

    # the same context's bare loop: what bench.py's end_to_end leg does (no metric frames, no staging from clip arrays)
    import sr_hip
    ctx = m.core.ctx(args.clips)
    lr_pin = sr_hip.PinnedArray((args.clips, 180, 320, 3), np.float32)
    lr_pin.a[...] = synth.synth_clip(0, 1, 180, 320)[0]
    outs = [sr_hip.PinnedArray((args.clips, 720, 1280, 3), np.uint8) for _ in range(2)]
    ctx.clip_begin(args.clips)
    n = 3 * args.frames
    for phase in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        for i in range(n):
            ctx.clip_push_u8(lr_pin.ptr, outs[i % 2].ptr)
            ctx.u8_wait(1)
        ctx.u8_wait(0)
        ctx.sync()
        bare = args.clips * n / (time.perf_counter() - t0)
    print(json.dumps({"what": "main.py --yaml_file options/180p.yaml, %d synthetic clips x %d frames, clips_in_flight %d"
                              % (args.clips, args.frames, args.clips),
                      "eval_fps": round(m.eval_fps, 1), "bare_end_to_end_fps": round(bare, 1),
                      "eval_over_bare": round(m.eval_fps / bare, 3), "stage_wall_s": round(wall, 2)}), flush=True)
    if not args.keep:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
