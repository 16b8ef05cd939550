"""Ablation builds of k_d2_dgrad_slab16 (-DRD_D2S_ABL_*): time of the critic layer-2 input-gradient launch at 3 x 2048 samples."""
import os, sys, glob
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import _lib, weights as W
from pr_disagg_radar_gan_amd.engine import Engine
from oracle import rdgan_torch as ot
rng = np.random.default_rng(5)
g, d = W.init_generator(rng, 16), W.init_critic(rng, 16)
B = int(os.environ.get("ABL_B", "2048"))
x, cond, z = ot.synthetic_batch(64, 16, 3)
rep = lambda a: torch.from_numpy(np.concatenate([a] * (B // 64))).cuda()
xd, cd, zd = rep(x), rep(cond), rep(z)
engs = []
for path in sorted(glob.glob(os.path.join(ROOT, "scratch", "librdgan_d2abl_*.so"))):
    _lib._lib = None; _lib.LIB_PATH = path
    e = Engine(16, B); e.set_option("bf16", 1)
    engs.append((os.path.basename(path), e, e.to_slab(g), e.to_slab(d)))
tot = {n: [] for n, *_ in engs}
for rnd in range(4):
    for n, e, gs, ds in engs:
        e.profile_launches(True)
        for _ in range(3):
            e.critic_grad(ds, gs, xd, cd, zd, 7)
        rows = [r for r in e.launch_table() if "layer2 dgrad" in r["name"]]
        e.profile_launches(False)
        tot[n].append(rows[0]["ms"] / rows[0]["launches"])
for n, v in tot.items():
    print(f"B {B} {n}: {np.median(v):.4f} ms (min {min(v):.4f})", flush=True)
