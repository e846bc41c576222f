"""Ablations of the bf16x3 trunk kernel at the production sub-batch (4 clips of 184x320, tensors resident in the Infinity Cache)
and at 16: SR_CONV_DBG bits 1 no tile DMA, 2 no MFMA loop (and no LDS reads), 4 no epilogue memory traffic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import sr_hip as sh
os.environ["SR_BENCH_TRUNK"] = "1"
for B in (4, 16):
    for dbg in (0, 1, 2, 4, 5, 6, 3, 7):
        os.environ["SR_CONV_DBG"] = str(dbg)
        t = sh.bench_conv(B, 184, 320, 64, 64, dtype="bf16x3", residual=True, iters=100)
        print(f"B={B:2d} dbg={dbg} ({'-'.join(n for b, n in ((1, 'noDMA'), (2, 'noMFMA'), (4, 'noEPI')) if dbg & b) or 'full'}): {t:.4f} ms", flush=True)
    os.environ.pop("SR_CONV_DBG")
    print(f"B={B:2d} f16mx6 residual: {sh.bench_conv(B, 184, 320, 64, 64, dtype='f16mx6', residual=True, iters=100):.4f} ms", flush=True)
