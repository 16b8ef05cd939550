"""nd 48, bf16 storage: does a forward pass change after a critic / generator step?  (round 4 probe)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import rdgan_torch as ot
from pr_disagg_radar_gan_amd import Engine
from tests.hip_util import dev
from tests.test_hip_step import _params
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 48
for opts in ({}, {"dense_skinny": 0}, {"dense16": 0}, {"d1_dgrad_fused": 0}, {"bf16off": 1}):
    eng = Engine(ndomain=nd, max_batch=1)
    g, d = _params(nd, 71)
    x, cond, z = ot.synthetic_batch(1, nd, 61)
    gs, ds = eng.to_slab(g), eng.to_slab(d)
    if "bf16off" not in opts:
        eng.set_option("bf16", 1)
    for k, v in opts.items():
        if k != "bf16off":
            eng.set_option(k, v)
    o0 = eng.gen_forward(gs, dev(z), dev(cond)).clone()
    h = [eng.debug_activation(i, s).clone() for i, s in enumerate([(1, 3, nd // 8, nd // 8, 256), (1, 6, nd // 4, nd // 4, 256), (1, 12, nd // 2, nd // 2, 128), (1, 24, nd, nd, 64)])]
    eng.critic_grad(ds, gs, dev(x), dev(cond), dev(z), 0)
    o1 = eng.gen_forward(gs, dev(z), dev(cond)).clone()
    h1 = [eng.debug_activation(i, s.shape).clone() for i, s in enumerate(h)]
    eng.gen_grad(ds, gs, dev(z), dev(cond), 0)
    o2 = eng.gen_forward(gs, dev(z), dev(cond)).clone()
    print(opts, "after critic step:", float((o1 - o0).abs().max() / o0.max()), [float((a - b).abs().max()) for a, b in zip(h, h1)],
          "after gen step:", float((o2 - o0).abs().max() / o0.max()))
    eng.close()
