import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 1
bias = torch.zeros(64, device="cuda")
y = torch.empty((B, 24, 16, 16, 64), device="cuda"); rinv = torch.empty((B, 24, 16, 16), device="cuda")
for c0 in (0, 5, 40, 127):
    x = torch.zeros((B, 12, 8, 8, 128), device="cuda"); x[..., c0] = 1.0
    hits = []
    for ci in range(128):
        w = torch.zeros((3, 3, 3, 128, 64), device="cuda"); w[1, 1, 1, ci, 5] = 1.0
        dbg = torch.zeros((B * 24 * 256, 4), device="cuda")
        rc = lib.rdgan_op_upconv_slab16(p(x), p(w), p(bias), p(y), p(rinv), p(dbg), B, st)
        assert rc == 0
        s = float(dbg[:, 0].sum())
        if s > 0:
            hits.append((ci, round(s, 1)))
    print("x channel", c0, "-> nonzero output for weight input-channels", hits, flush=True)
