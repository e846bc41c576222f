"""Ablations of the f16mx6 trunk kernel (conv_trunk6.hip), residual layer, 184x320: SR_CONV_DBG bits 1 no tile DMA, 2 no MFMAs,
4 no epilogue loads / stores, 8 no fragment reads, 16 no epilogue arithmetic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import sr_hip as sh
names = {0: "full", 1: "no DMA", 4: "no epilogue memory", 16: "no epilogue arithmetic", 5: "no DMA, no epilogue memory",
         21: "MFMAs + fragment reads only", 29: "MFMAs only", 23: "fragment reads only", 30: "tile DMA only", 27: "epilogue memory only",
         31: "empty pipeline"}
for B in (4, 16):
    for dbg in (0, 1, 4, 16, 5, 21, 29, 23, 30, 27, 31):
        os.environ["SR_CONV_DBG"] = str(dbg)
        t = sh.bench_conv(B, 184, 320, 64, 64, dtype="f16mx6", residual=True, iters=100)
        print(f"B={B:2d} dbg={dbg:2d} {names[dbg]:32s}: {t:.4f} ms", flush=True)
    os.environ.pop("SR_CONV_DBG")
