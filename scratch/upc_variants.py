import os, sys, glob
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import _lib, weights as W
from pr_disagg_radar_gan_amd.engine import Engine
from oracle import rdgan_torch as ot
rng = np.random.default_rng(5)
g = W.init_generator(rng, 16)
for B in (2, 64):
    x, cond, z = ot.synthetic_batch(B, 16, 3)
    zd, cd = torch.from_numpy(z).cuda(), torch.from_numpy(cond).cuda()
    ref = None
    for path in sorted(glob.glob(os.path.join(ROOT, "scratch", "librdgan_upc_*.so"))):
        _lib._lib = None; _lib.LIB_PATH = path
        eng = Engine(16, B); gs = eng.to_slab(g); eng.set_option("bf16", 1)
        if ref is None:
            eng.set_option("upconv_slab", 0); eng.gen_forward(gs, zd, cd); ref = eng.debug_activation(3, (B, 24, 16, 16, 64)).clone()
        eng.set_option("upconv_slab", 1)
        res = []
        for rep in range(3):
            eng.gen_forward(gs, zd, cd); h = eng.debug_activation(3, (B, 24, 16, 16, 64))
            res.append(int((((h - ref).abs() / ref.abs().clamp_min(1e-3)) > 2**-6).sum()))
        print(f"B {B} {os.path.basename(path)}: elements off by more than 2^-6: {res} of {ref.numel()}", flush=True)
        eng.close()
