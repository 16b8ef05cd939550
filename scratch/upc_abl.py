"""Ablation builds of k_upconv_slab16 (-DRD_UPC_ABL_*): time of the block-3 forward launch at bs 256 and 2048, interleaved rounds."""
import os, sys, glob
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import _lib, weights as W
from pr_disagg_radar_gan_amd.engine import Engine
from oracle import rdgan_torch as ot
rng = np.random.default_rng(5)
g = W.init_generator(rng, 16)
for B in (256, 2048):
    x, cond, z = ot.synthetic_batch(64, 16, 3)
    z = np.concatenate([z] * (B // 64)); cond = np.concatenate([cond] * (B // 64))
    zd, cd = torch.from_numpy(z).cuda(), torch.from_numpy(cond).cuda()
    engs = []
    for path in sorted(glob.glob(os.path.join(ROOT, "scratch", "librdgan_abl_*.so"))):
        _lib._lib = None; _lib.LIB_PATH = path
        e = Engine(16, B); e.set_option("bf16", 1)
        engs.append((os.path.basename(path), e, e.to_slab(g)))
    tot = {n: [] for n, _, _ in engs}
    for rnd in range(4):
        for n, e, gs in engs:
            e.profile_launches(True)
            for _ in range(4):
                e.gen_forward(gs, zd, cd)
            rows = [r for r in e.launch_table() if "block3" in r["name"] and r["kind"] == "gemm"]
            e.profile_launches(False)
            tot[n].append(rows[0]["ms"] / rows[0]["launches"])
    for n, v in tot.items():
        print(f"B {B} {n}: {np.median(v):.4f} ms (min {min(v):.4f})", flush=True)
    for _, e, _ in engs:
        e.close()
