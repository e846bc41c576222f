import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import sr_hip as sh
np.set_printoptions(linewidth=220, precision=6, suppress=True)
x = np.zeros((1, 1, 2, 64), np.float32)
x[0, 0, 0] = np.arange(64) * 0.37 + 0.1
x[0, 0, 1] = -np.arange(64) * 0.001
raw = sh.conv3x3_mx6(x, mode=4).view(np.uint8).reshape(4, 2, 64)      # plane, pixel, byte
def fp6(code):
    s, e, m = code >> 5, (code >> 3) & 3, code & 7
    a = m / 8 if e == 0 else (1 + m / 8) * 2.0 ** (e - 1)
    return -a if s else a
def c6(e): return ((e >> 1) & 3) + 8 * (e >> 3) + 4 * (e & 1)
for p in range(2):
    hi = np.zeros(64)
    for P in range(2):
        rec = raw[P, p].view(np.float16)
        for u in range(4):
            h, ks = u >> 1, u & 1
            for e in range(8):
                ch = 32 * P + 16 * ks + (4 * h + e if e < 4 else 8 + 4 * h + e - 4)
                hi[ch] = rec[8 * u + e]
    print("pixel", p, "hi - f16(x):", np.abs(hi - x[0, 0, p].astype(np.float16).astype(np.float64)).max())
    for plane, name in ((2, "x6"), (3, "lo6")):
        val = np.zeros(64)
        for blk in range(2):
            b = raw[plane, p, 32 * blk:32 * blk + 32]
            bits = int.from_bytes(bytes(b[:24]), "little")
            sc = 2.0 ** (int(b[24]) - 127)
            for e in range(32):
                val[32 * blk + c6(e)] = fp6((bits >> (6 * e)) & 63) * sc
            print(name, "block", blk, "scale byte", b[24], "pad", b[25:32])
        tgt = x[0, 0, p].astype(np.float64) if plane == 2 else x[0, 0, p].astype(np.float64) - hi
        print(name, "decoded", val)
        print(name, "target ", tgt)

back = sh.conv3x3_mx6(x, mode=2)
x6 = sh.conv3x3_mx6(x, mode=3)
for p in range(2):
    print("pixel", p, "back err", np.abs(back[0, 0, p] - x[0, 0, p]).max(), "x6 err", np.abs(x6[0, 0, p] - x[0, 0, p]).max())
    print(back[0, 0, p])
# identity conv on an 8 x 64 image
rng = np.random.default_rng(0)
xi = np.abs(rng.standard_normal((1, 8, 64, 64))).astype(np.float32)
w = np.zeros((3, 3, 64, 64), np.float32)
for c in range(64):
    w[1, 1, c, c] = 1.0
b = np.zeros(64, np.float32)
y = sh.conv3x3_mx6(xi, w, b, relu=True)
print("identity conv: max err", np.abs(y - xi).max(), "nonzero fraction", (y != 0).mean())
print("x[0,0,0,:16]", xi[0, 0, 0, :16])
print("y[0,0,0,:16]", y[0, 0, 0, :16])
print("y[0,3,17,:16]", y[0, 3, 17, :16], "\nx[0,3,17,:16]", xi[0, 3, 17, :16])
