"""Which tensors of the gradient slab differ between two identical gen_grad calls (bf16, ndomain 64)?"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from oracle import rdgan_torch as ot
nd = 64
for B in (64, 8):
    eng = Engine(ndomain=nd, max_batch=B)
    rng = np.random.default_rng(16)
    g, d = W.init_generator(rng, nd), W.init_critic(rng, nd)
    gs, ds = eng.to_slab(g), eng.to_slab(d)
    x, cond, z = ot.synthetic_batch(B, nd, 9)
    dev = lambda a: torch.from_numpy(a).cuda()
    zd, cd = dev(z), dev(cond)
    for opts in ({"bf16": 1}, {"bf16": 1, "side_stream": 0}, {"bf16": 1, "edge_kernels": 0}, {"bf16": 1, "fast_bwd": 1, "fast_fwd": 1},
                 {"bf16": 1, "resident": 0}, {"bf16": 1, "ws_ksplit": 0}):
        for k, v in {"side_stream": 1, "edge_kernels": 1, "fast_bwd": -1, "fast_fwd": -1, "resident": 1, "ws_ksplit": 1}.items():
            eng.set_option(k, v)
        for k, v in opts.items():
            eng.set_option(k, v)
        runs = [eng.gen_grad(ds, gs, zd, cd, 31338).clone() for _ in range(4)]
        torch.cuda.synchronize()
        bad = {}
        off = 0
        for name, s in eng.gen_shapes:
            n = int(np.prod(s))
            for r in runs[1:]:
                dd = (r[off:off + n] - runs[0][off:off + n]).abs()
                if float(dd.max()) > 0:
                    bad[name] = (float(dd.max()), int((dd > 0).sum()), n, float(runs[0][off:off + n].abs().max()))
            off += n
        print(f"B {B} opts {opts}: differing tensors: {bad if bad else 'none'}", flush=True)
    eng.close()
