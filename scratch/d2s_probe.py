"""one call of rdgan_op_d2_dgrad_slab16 with the library given on the command line; mismatch pattern against the definition"""
import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import _lib
from oracle import rdgan_np as onp
_lib.LIB_PATH = os.path.abspath(sys.argv[1]); B = int(sys.argv[2])
lib = _lib.load()
rng = np.random.default_rng(1)
r16 = lambda a: torch.from_numpy(a).bfloat16().float().numpy()
gy_h = r16(rng.standard_normal((B, 6, 4, 4, 128)).astype(np.float32))
w_h = r16((0.05 * rng.standard_normal((3, 3, 3, 64, 128))).astype(np.float32))
aux_h = np.ones((B, 11, 7, 7, 64), np.float32)
gy, w, aux = (torch.from_numpy(a).cuda() for a in (gy_h, w_h, aux_h))
gx = torch.full((B, 11, 7, 7, 64), float("nan"), device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
torch.cuda.synchronize()
rc = lib.rdgan_op_d2_dgrad_slab16(p(gy), p(w), p(aux), p(gx), B, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
got = gx.cpu().numpy().astype(np.float64)
ref = onp.conv3d_input_grad(gy_h.astype(np.float64), w_h.astype(np.float64), (11, 7, 7), 2, (1, 1, 1))
bad = np.abs(got - ref) > 2.0 ** -7 * np.abs(ref) + 1e-3
print(sys.argv[1], "B", B, "rc", rc, "bad", int(bad.sum()), "of", bad.size, flush=True)
if bad.any():
    for cls in range(8):
        pd, ph, pw = cls >> 2, (cls >> 1) & 1, cls & 1
        sub = bad[:, (1 - pd)::2, (1 - ph)::2, (1 - pw)::2, :]
        print(" phase", cls, "bad", int(sub.sum()), "of", sub.size, "by channel block of 8:", sub.reshape(-1, 8, 8).sum(axis=(0, 2)).tolist(),
              "by d:", sub.sum(axis=(0, 2, 3, 4)).tolist())
    idx = np.argwhere(bad)[:6]
    for i in idx:
        print("  ", i.tolist(), got[tuple(i)], ref[tuple(i)])
