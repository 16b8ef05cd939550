import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import _lib
from oracle import rdgan_torch as ot
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator(); g.manual_seed(1)
x = torch.randn((B, 12, 8, 8, 128), generator=g)
w = 0.02 * torch.randn((3, 3, 3, 128, 64), generator=g)
bias = 0.05 * torch.randn((64,), generator=g)
xb = x.to(torch.bfloat16).double()
u = ot.upsample3d(xb)
yr = ot._conv3d_tf(u, w.double(), bias.double(), 1, (1, 1, 1), u.shape[1:4])
ss_ref = (yr * yr).sum(-1)
yref = ot._lrelu(ot.pixel_norm(yr)).float()
xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
y = torch.empty((B, 24, 16, 16, 64), device="cuda"); rinv = torch.empty((B, 24, 16, 16), device="cuda")
dbg = torch.zeros((B * 24 * 256, 4), device="cuda")
rc = lib.rdgan_op_upconv_slab16(p(xd), p(wd), p(bd), p(y), p(rinv), p(dbg), B, st)
assert rc == 0, rc
y = y.cpu(); d = dbg.cpu().reshape(B, 24, 16, 16, 4)
err = (y - yref).abs()
ratio = d[..., 0].double() / ss_ref
print("ss ratio kernel/ref: min %.3f max %.3f" % (float(ratio.min()), float(ratio.max())))
badrows = torch.nonzero((ratio - 1).abs() > 0.03)
print("rows with a wrong sum of squares:", len(badrows), "of", ratio.numel())
if len(badrows):
    for name, col, n in (("b", 0, B), ("d", 1, 24), ("h", 2, 16), ("w", 3, 16)):
        print("  by", name, torch.bincount(badrows[:, col], minlength=n).tolist())
print("max err", float(err.max()))
# second opinion: the validated streaming bf16 conv op on the upsampled tensor (direct 27-tap form), then PixelNorm + LeakyReLU in torch
ud = u.float().cuda().contiguous()
ys = torch.empty((B, 24, 16, 16, 64), device="cuda")
rc = lib.rdgan_op_conv3d_bf16(p(ud), p(wd), p(bd), p(ys), B, 24, 16, 16, 128, 64, 24, 16, 16, 1, 1, 1, 1, 0, st)
assert rc == 0, rc
ys = ys.cpu().double()
print("streaming op vs oracle (pre-norm): max rel", float((ys - yr).abs().max() / yr.abs().max()))
ss_s = (ys * ys).sum(-1)
print("ss ratio kernel/streaming-op: min %.3f max %.3f" % (float((d[..., 0].double() / ss_s).min()), float((d[..., 0].double() / ss_s).max())))
# which taps does the slab kernel drop?  zero all weights except one tap (kd,kh,kw) and compare pre-norm energy per output parity
