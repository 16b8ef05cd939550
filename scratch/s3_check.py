"""split3 accuracy / speed check against the fp64 oracle (B = 33): python3 scratch/s3_check.py"""
import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rdgan_torch as ot
from pr_disagg_radar_gan_amd import Engine
from tests.hip_util import dev, rel_err, hip_gates
from tests.test_hip_step import _params, _t64, _grad_errors
B = 33
eng = Engine(ndomain=16, max_batch=B)
g, d = _params(16, 91)
gs, ds = eng.to_slab(g), eng.to_slab(d)
x, cond, z = ot.synthetic_batch(B, 16, 191)
ref = ot.generator_forward(_t64(g), torch.from_numpy(z).double(), torch.from_numpy(cond).double()).numpy()
for s3 in (0, 1):
    eng.set_option("split3", s3)
    eng.set_option("wave_specialized", 2)
    out = eng.gen_forward(gs, dev(z), dev(cond)).cpu().numpy()
    print("split3", s3, "forward rel err vs fp64 %.2e" % rel_err(out, ref))
    slab = eng.gen_grad(ds, gs, dev(z), dev(cond), 4712).cpu().numpy()
    gates = hip_gates(eng, B)
    loss, grads = ot.gen_step_grads(_t64(d), _t64(g), torch.from_numpy(z).double(), torch.from_numpy(cond).double(), 4712, gates=gates)
    errs = _grad_errors(slab[:eng.n_gen], grads, eng.gen_shapes)
    print("  gen grads max rel err %.2e" % max(errs.values()))
    cs = eng.critic_grad(ds, gs, dev(x), dev(cond), dev(z), 4711).cpu().numpy()
    losses, cg = ot.critic_step_grads(_t64(d), _t64(g), torch.from_numpy(x).double(), torch.from_numpy(cond).double(), torch.from_numpy(z).double(), 4711)
    errs = _grad_errors(cs[:eng.n_critic], cg, eng.critic_shapes)
    print("  critic grads max rel err %.2e" % max(errs.values()))
eng.close()
