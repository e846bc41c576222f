import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import sr_hip as sh
os.environ["SR_BENCH_TRUNK"] = "1"
for B in (4, 16):
    for res in (False, True):
        a = sh.bench_conv(B, 184, 320, 64, 64, dtype="bf16x3", residual=res, iters=50)
        b = sh.bench_conv(B, 184, 320, 64, 64, dtype="f16mx6", residual=res, iters=50)
        print(f"B={B} 184x320 {'residual' if res else 'relu    '}: bf16x3 {a:.4f} ms, f16mx6 {b:.4f} ms  ({a / b:.2f}x)", flush=True)
