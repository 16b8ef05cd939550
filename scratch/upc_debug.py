import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from oracle import rdgan_torch as ot
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eng = Engine(16, B)
rng = np.random.default_rng(5)
g = W.init_generator(rng, 16)
g = [p if p.ndim > 1 else (0.05 * rng.standard_normal(p.shape)).astype(np.float32) for p in g]
gs = eng.to_slab(g)
x, cond, z = ot.synthetic_batch(B, 16, 3)
zd, cd = torch.from_numpy(z).cuda(), torch.from_numpy(cond).cuda()
eng.set_option("bf16", 1)
eng.set_option("upconv_slab", 0)
eng.gen_forward(gs, zd, cd); h0 = eng.debug_activation(3, (B, 24, 16, 16, 64)).clone()
eng.set_option("upconv_slab", 1)
for rep in range(3):
    eng.gen_forward(gs, zd, cd); h1 = eng.debug_activation(3, (B, 24, 16, 16, 64)).clone()
    bad = ((h1 - h0).abs() / h0.abs().clamp_min(1e-3) > 2**-6) | ~torch.isfinite(h1)
    print("rep", rep, "bad", int(bad.sum()))
    idx = torch.nonzero(bad)
    if len(idx):
        for name, col, n in (("b", 0, B), ("d", 1, 24), ("h", 2, 16), ("w", 3, 16), ("c", 4, 64)):
            cnt = torch.bincount(idx[:, col], minlength=n).tolist()
            print("  by", name, cnt)
        print("  first few", idx[:8].tolist())
        i = idx[0]
        print("  values slab", h1[tuple(i[:4])][:16].tolist(), "\n  stream", h0[tuple(i[:4])][:16].tolist())
