import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from oracle import rdgan_torch as ot
B = 4
eng = Engine(ndomain=16, max_batch=B)
rng = np.random.default_rng(3)
g, d = W.init_generator(rng, 16), W.init_critic(rng, 16)
x, cond, z = ot.synthetic_batch(B, 16, 5)
dev = lambda a: torch.from_numpy(a).cuda()
gs, ds = eng.to_slab(g), eng.to_slab(d)
eng.set_option("bf16", 1)
for seed in (0, 9):
    v = eng.critic_forward(ds, dev(x), dev(cond), seed)
    h1 = eng.debug_activation(4, (B, 539, 64)).cpu().numpy()
    gb = eng.debug_activation(8, (B, 539, 16)).cpu().numpy().astype(np.uint8)
    h1b = torch.from_numpy(h1).view(torch.int32).numpy()
    pos = (h1 > 0).astype(np.uint8); drp = ((h1b == 0) & (seed != 0)).astype(np.uint8)
    code = (pos | (drp << 1)).reshape(B, 539, 16, 4)
    want = (code[..., 0] | (code[..., 1] << 2) | (code[..., 2] << 4) | (code[..., 3] << 6)).astype(np.uint8)
    bad = want != gb
    print("seed", seed, "bytes differing", int(bad.sum()), "of", bad.size, "dropped share", float(drp.mean()), flush=True)
    if bad.any():
        i = np.argwhere(bad)[:5]
        for j in i: print("  ", j.tolist(), "want", bin(want[tuple(j)]), "got", bin(gb[tuple(j)]), "h1", h1[j[0], j[1], 4 * j[2]:4 * j[2] + 4])
