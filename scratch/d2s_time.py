"""Per-launch time of the critic layer-2 input gradient (3 x B samples) for several builds, interleaved rounds."""
import os, sys, glob
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ctypes
from pr_disagg_radar_gan_amd import _lib, weights as W
from pr_disagg_radar_gan_amd.engine import Engine
from oracle import rdgan_torch as ot
rng = np.random.default_rng(5)
g, d = W.init_generator(rng, 16), W.init_critic(rng, 16)
B = int(os.environ.get("ABL_B", "2048"))
what = os.environ.get("ABL_WHAT", "layer2 dgrad")
x, cond, z = ot.synthetic_batch(64, 16, 3)
rep = lambda a: torch.from_numpy(np.concatenate([a] * (B // 64))).cuda()
xd, cd, zd = rep(x), rep(cond), rep(z)
engs = []
for path in sys.argv[1:]:
    _lib._lib = None; _lib.LIB_PATH = os.path.abspath(path)
    probe = ctypes.CDLL(_lib.LIB_PATH); sigs = dict(_lib.SIGNATURES)
    _lib.SIGNATURES = {k: v for k, v in sigs.items() if hasattr(probe, k)}
    e = Engine(16, B); e.set_option("bf16", 1)
    _lib.SIGNATURES = sigs
    engs.append((os.path.basename(path), e, e.to_slab(g), e.to_slab(d)))
tot = {n: {} for n, *_ in engs}
for rnd in range(4):
    for n, e, gs, ds in engs:
        e.profile_launches(True)
        for _ in range(3):
            e.critic_grad(ds, gs, xd, cd, zd, 7)
        for r in e.launch_table():
            if what in r["name"]:
                tot[n].setdefault((r["name"], r["kind"], r["kernel"]), []).append(r["ms"] / r["launches"])
        e.profile_launches(False)
for n, dd in tot.items():
    for k, v in dd.items():
        print(f"B {B} {n}: {k}: {np.median(v):.4f} ms (min {min(v):.4f})", flush=True)
