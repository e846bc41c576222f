import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import sr_hip as sh
np.set_printoptions(linewidth=250, precision=4, suppress=True)
w = np.zeros((3, 3, 64, 64), np.float32)
for c in range(64):
    w[1, 1, c, c] = 1.0
b = np.zeros(64, np.float32)
rng = np.random.default_rng(0)
xi = np.abs(rng.standard_normal((1, 8, 64, 64))).astype(np.float32)
y = sh.conv3x3_mx6(xi, w, b, relu=True)
print("identity: y[0,3,17,:]\n", y[0, 3, 17], "\nx\n", xi[0, 3, 17])
for k in (0, 5, 21, 40):
    x = np.zeros((1, 8, 64, 64), np.float32)
    x[0, 3, 17, k] = 1.0
    y = sh.conv3x3_mx6(x, w, b, relu=True)
    nz = np.argwhere(np.abs(y) > 1e-3)
    print("one-hot channel", k, "at (3,17): nonzero outputs (b,y,x,c) value:", [(tuple(int(v) for v in i), float(y[tuple(i)])) for i in nz[:24]], "count", len(nz))
x = np.zeros((1, 8, 64, 64), np.float32)
bb = np.arange(64, dtype=np.float32) * 0.01 + 0.01
y = sh.conv3x3_mx6(x, w * 0, bb, relu=True)
print("bias only: y[0,2,9,:]\n", y[0, 2, 9])
