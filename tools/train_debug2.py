import sys, os
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import sr_hip as sh
from oracle import ref_torch as rt
rng = np.random.default_rng(0)
def t64(a): return torch.tensor(np.asarray(a, np.float64))
for (B,H,W,Cin,Cout) in [(3,4,1,128,256),(1,4,1,128,256),(3,4,2,128,256),(3,1,4,128,256),(3,4,1,64,64),(2,4,1,32,32),(3,8,2,128,128),(3,2,1,16,16)]:
    x = rng.standard_normal((B,H,W,Cin)).astype(np.float32); w = (0.1*rng.standard_normal((3,3,Cin,Cout))).astype(np.float32)
    dy = rng.standard_normal((B,H,W,Cout)).astype(np.float32)
    xt = t64(x).permute(0,3,1,2).requires_grad_(True); wt = t64(w).requires_grad_(True)
    F.conv2d(xt, wt.permute(3,2,0,1), None, padding=1).backward(t64(dy).permute(0,3,1,2))
    dx, dw, db = sh.conv3x3_bwd(x, w, dy)
    e = lambda a,b: np.abs(a-b).max()/np.abs(b).max()
    print("conv_bwd", (B,H,W,Cin,Cout), "dx %.1e dw %.1e" % (e(dx, xt.grad.permute(0,2,3,1).numpy()), e(dw, wt.grad.numpy())))
    y = sh.conv3x3(x, w, np.zeros(Cout, np.float32))
    yt = F.conv2d(t64(x).permute(0,3,1,2), t64(w).permute(3,2,0,1), None, padding=1).permute(0,2,3,1).numpy()
    print("   fwd %.1e" % e(y, yt))
for (B,H,W,C) in [(3,8,2,128),(3,4,1,32),(1,8,2,128)]:
    x = rng.standard_normal((B,H,W,C)).astype(np.float32); gy = rng.standard_normal((B,(H+1)//2,(W+1)//2,C)).astype(np.float32)
    xt = t64(x).permute(0,3,1,2).requires_grad_(True)
    rt.maxpool2_same(xt).backward(t64(gy).permute(0,3,1,2))
    gx = sh.maxpool2_bwd(x, gy)
    print("pool_bwd", (B,H,W,C), np.abs(gx - xt.grad.permute(0,2,3,1).numpy()).max())
