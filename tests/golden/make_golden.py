"""Generates tests/golden/rdgan_nd16_b2.npz from the CPU oracle (fp64).

The reference ships no golden vectors and cannot run here (TensorFlow absent, weights absent), so this
fixture is NOT reference output: it freezes the oracle's answers for one seeded case so that (a) oracle
drift is caught by the CPU suite and (b) the GPU suite can check the HIP path without re-deriving them.
Weights are regenerated from the seed (pr_disagg_radar_gan_amd.weights initialisers), not stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rdgan_torch as ot          # noqa: E402
from pr_disagg_radar_gan_amd import weights as W   # noqa: E402

WEIGHT_SEED, DATA_SEED, STEP_SEED, B, ND = 2024, 141, 4242, 2, 16   # data seed picked for the largest LeakyReLU kink margin in 100..159


def case():
    rng = np.random.default_rng(WEIGHT_SEED)
    g, d = W.init_generator(rng, ND), W.init_critic(rng, ND)
    g = [p if p.ndim > 1 else (0.05 * rng.standard_normal(p.shape)).astype(np.float32) for p in g]
    d = [p if p.ndim > 1 else (0.05 * rng.standard_normal(p.shape)).astype(np.float32) for p in d]
    x, cond, z = ot.synthetic_batch(B, ND, DATA_SEED)
    return g, d, x, cond, z


def main():
    g, d, x, cond, z = case()
    t64 = lambda arrs: [torch.from_numpy(a).double() for a in arrs]
    tx, tc, tz = (torch.from_numpy(a).double() for a in (x, cond, z))
    fwd = ot.generator_forward(t64(g), tz, tc).numpy()
    v0 = ot.critic_forward(t64(d), torch.from_numpy(fwd), tc, None).numpy()
    v1 = ot.critic_forward(t64(d), torch.from_numpy(fwd), tc, ot.critic_masks(STEP_SEED, B, ND, torch.float64)).numpy()
    closs, cgrads = ot.critic_step_grads(t64(d), t64(g), tx, tc, tz, STEP_SEED)
    gloss, ggrads = ot.gen_step_grads(t64(d), t64(g), tz, tc, STEP_SEED)
    margin = ot.kink_margin(t64(d), t64(g), tx, tc, tz, STEP_SEED, True)
    out = dict(weight_seed=WEIGHT_SEED, data_seed=DATA_SEED, step_seed=STEP_SEED, x=x, cond=cond, z=z,
               gen_out=fwd.astype(np.float32), critic_nodrop=v0, critic_drop=v1, critic_losses=closs.numpy(),
               gen_loss=np.array(gloss.item()), kink_margin=np.array(margin))
    for i, t in enumerate(cgrads):                # per-tensor digests keep the file small
        a = t.numpy().ravel()
        out[f"cgrad{i}_l2"] = np.array(np.sqrt((a * a).sum())); out[f"cgrad{i}_head"] = a[:16].copy()
        out[f"cgrad{i}_absmax"] = np.array(np.abs(a).max())
    for i, t in enumerate(ggrads):
        a = t.numpy().ravel()
        out[f"ggrad{i}_l2"] = np.array(np.sqrt((a * a).sum())); out[f"ggrad{i}_head"] = a[:16].copy()
        out[f"ggrad{i}_absmax"] = np.array(np.abs(a).max())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rdgan_nd16_b2.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; kink margin", margin)


if __name__ == "__main__":
    main()
