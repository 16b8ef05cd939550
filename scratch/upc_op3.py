import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import _lib
from oracle import rdgan_torch as ot
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 1
# x[b,d,h,w,c] = position code in channel ci; single-tap weights pick one channel -> the output tells which source position was read
pos = torch.arange(12 * 64, dtype=torch.float32).reshape(1, 12, 8, 8, 1) % 251 + 1      # exact in bf16? values < 256 are
x = torch.zeros((B, 12, 8, 8, 128)); 
for ci in (0, 77, 127):
    x[..., ci] = pos[..., 0]
bias = torch.zeros(64)
tot_bad = 0
for ci in (0, 77):
  for tap in (0, 13, 26, 4, 22):
    w = torch.zeros((3, 3, 3, 128, 64)); w[tap // 9, (tap // 3) % 3, tap % 3, ci, 5] = 1.0
    u = ot.upsample3d(x.double())
    yr = ot._conv3d_tf(u, w.double(), bias.double(), 1, (1, 1, 1), u.shape[1:4])[..., 5]
    y = torch.empty((B, 24, 16, 16, 64), device="cuda"); rinv = torch.empty((B, 24, 16, 16), device="cuda")
    dbg = torch.zeros((B * 24 * 256, 4), device="cuda")
    xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
    rc = lib.rdgan_op_upconv_slab16(p(xd), p(wd), p(bd), p(y), p(rinv), p(dbg), B, st)
    assert rc == 0
    got = dbg.cpu().reshape(B, 24, 16, 16, 4)[..., 0].double().sqrt()
    bad = torch.nonzero((got - yr).abs() > 0.5)
    tot_bad += len(bad)
    if len(bad):
        i = tuple(bad[0].tolist())
        srcpos = lambda v: ((int(v) - 1) // 64, ((int(v) - 1) % 64) // 8, (int(v) - 1) % 8)
        for i2 in bad[:6]:
            i2 = tuple(i2.tolist())
            print("      out", i2[1:], "got code", float(got[i2]), "=src", srcpos(float(got[i2])) if float(got[i2]) >= 1 else None, "want code", float(yr[i2]), "=src", srcpos(float(yr[i2])) if float(yr[i2]) >= 1 else None)
        print(f"ci {ci} tap {tap} (kd {tap//9} kh {(tap//3)%3} kw {tap%3}): {len(bad)} wrong positions; first {i}: got {float(got[i])} want {float(yr[i])};",
              "by d", torch.bincount(bad[:, 1] % 2, minlength=2).tolist(), "h", torch.bincount(bad[:, 2] % 2, minlength=2).tolist(),
              "w", torch.bincount(bad[:, 3] % 2, minlength=2).tolist())
print("total wrong", tot_bad)
