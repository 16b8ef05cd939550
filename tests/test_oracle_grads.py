"""Pins the autograd restatement's gradients (incl. the gradient-penalty double backward)
against fp64 central differences of the numpy from-the-definitions loss."""
import numpy as np
import torch

from oracle import rdgan_np as onp
from oracle import rdgan_torch as ot
from oracle import rng as orng


def _setup(B=2, nd=16, seed=5):
    rng = np.random.default_rng(seed)
    gpar = [p.astype(np.float64) for p in onp.init_generator(rng, nd)]
    dpar = [p.astype(np.float64) for p in onp.init_critic(rng, nd)]
    # non-zero biases so bias gradients are exercised
    dpar = [p if p.ndim > 1 else 0.05 * rng.standard_normal(p.shape) for p in dpar]
    gpar = [p if p.ndim > 1 else 0.05 * rng.standard_normal(p.shape) for p in gpar]
    x, cond, z = ot.synthetic_batch(B, nd, seed + 1, np.float64)
    return rng, gpar, dpar, x, cond, z


def test_input_gradient_chain_vs_autograd():
    rng, gpar, dpar, x, cond, z = _setup()
    masks = [orng.dropout_scale_mask(3, 1 + i, (2,) + onp.critic_geometry(16)[i][1] + (c,)).astype(np.float64)
             for i, c in enumerate((64, 128, 256, 256))]
    v, g = onp.critic_input_gradient(dpar, x, cond, masks)
    xt = torch.from_numpy(x).requires_grad_(True)
    vt = ot.critic_forward([torch.from_numpy(p) for p in dpar], xt, torch.from_numpy(cond),
                           [torch.from_numpy(m) for m in masks])
    gt, = torch.autograd.grad(vt.sum(), xt)
    np.testing.assert_allclose(v, vt.detach().numpy(), rtol=1e-10)
    np.testing.assert_allclose(g, gt.numpy(), rtol=1e-9, atol=1e-14)


def test_critic_step_grads_vs_finite_differences():
    rng, gpar, dpar, x, cond, z = _setup()
    seed = 77
    B = x.shape[0]
    tg = [torch.from_numpy(p) for p in gpar]
    td = [torch.from_numpy(p) for p in dpar]
    losses, grads = ot.critic_step_grads(td, tg, torch.from_numpy(x), torch.from_numpy(cond),
                                         torch.from_numpy(z), seed)
    fake = onp.generator_forward(gpar, z, cond)
    alpha = orng.uniform(seed, orng.STREAM_ALPHA, B).astype(np.float64)
    masks3 = [orng.dropout_scale_mask(seed, 1 + i, (3 * B,) + onp.critic_geometry(16)[i][1] + (c,)).astype(np.float64)
              for i, c in enumerate((64, 128, 256, 256))]
    base = onp.critic_loss(dpar, x, fake, cond, alpha, masks3)
    np.testing.assert_allclose(losses.numpy(), np.array(base), rtol=1e-9)
    # bias of the GP term must be exactly zero-gradient: checked implicitly by the FD below.
    eps = 1e-6
    for trial in range(3):
        direction = [rng.standard_normal(p.shape) for p in dpar]
        nrm = np.sqrt(sum(float((d * d).sum()) for d in direction))
        direction = [d / nrm for d in direction]   # tiny step: the GP term is only piecewise smooth in the weights
        plus = [p + eps * d for p, d in zip(dpar, direction)]
        minus = [p - eps * d for p, d in zip(dpar, direction)]
        fd = (onp.critic_loss(plus, x, fake, cond, alpha, masks3)[0]
              - onp.critic_loss(minus, x, fake, cond, alpha, masks3)[0]) / (2 * eps)
        an = sum(float((g.numpy() * d).sum()) for g, d in zip(grads, direction))
        assert abs(fd - an) <= 2e-5 * max(1.0, abs(an)), (fd, an)
    # per-tensor directional checks (catches one wrong tensor hidden in the sum)
    for i in range(len(dpar)):
        d = rng.standard_normal(dpar[i].shape); d /= np.linalg.norm(d)
        plus = list(dpar); minus = list(dpar)
        plus[i] = dpar[i] + eps * d; minus[i] = dpar[i] - eps * d
        fd = (onp.critic_loss(plus, x, fake, cond, alpha, masks3)[0]
              - onp.critic_loss(minus, x, fake, cond, alpha, masks3)[0]) / (2 * eps)
        an = float((grads[i].numpy() * d).sum())
        assert abs(fd - an) <= 5e-5 * max(1e-3, abs(an)) + 1e-8, (i, fd, an)


def test_gen_step_grads_vs_finite_differences():
    rng, gpar, dpar, x, cond, z = _setup(seed=9)
    seed = 123
    B = z.shape[0]
    loss, grads = ot.gen_step_grads([torch.from_numpy(p) for p in dpar], [torch.from_numpy(p) for p in gpar],
                                    torch.from_numpy(z), torch.from_numpy(cond), seed)
    masks = [orng.dropout_scale_mask(seed, 1 + i, (B,) + onp.critic_geometry(16)[i][1] + (c,)).astype(np.float64)
             for i, c in enumerate((64, 128, 256, 256))]

    def L(gp_):
        img = onp.generator_forward(gp_, z, cond)
        return float(np.mean(-onp.critic_forward(dpar, img, cond, masks)))

    np.testing.assert_allclose(loss.item(), L(gpar), rtol=1e-10)
    eps = 1e-6
    for i in (0, 1, 2, 3, 5, 6, 8, 9):
        d = rng.standard_normal(gpar[i].shape); d /= np.linalg.norm(d)
        plus = list(gpar); minus = list(gpar)
        plus[i] = gpar[i] + eps * d; minus[i] = gpar[i] - eps * d
        fd = (L(plus) - L(minus)) / (2 * eps)
        an = float((grads[i].numpy() * d).sum())
        assert abs(fd - an) <= 1e-4 * max(1e-4, abs(an)) + 1e-9, (i, fd, an)


def test_slope_pattern_of_the_run_itself_changes_nothing():
    """gen_step_grads(gates=...) differentiates the LeakyReLU branch given by an external slope pattern (the GPU tests pass
    the fp32 run's); with the oracle's own pattern it must reproduce the plain call exactly."""
    import torch
    from oracle import rdgan_np as onp
    from oracle import rdgan_torch as ot
    rng = np.random.default_rng(5)
    g = [torch.from_numpy(a) for a in onp.init_generator(rng, 8, dtype=np.float64)]
    d = [torch.from_numpy(a) for a in onp.init_critic(rng, 8, dtype=np.float64)]
    x, cond, z = (torch.from_numpy(a) for a in ot.synthetic_batch(2, 8, 3, dtype=np.float64))
    with torch.no_grad():
        img, gi = ot.generator_forward(g, z, cond, True)
        _, di = ot.critic_forward(d, img, cond, ot.critic_masks(7, 2, 8, torch.float64), True)
    gates = ([gi[k] > 0 for k in ("h0", "h1", "h2", "h3")], [h > 0 for h in di["h"]])
    l0, g0 = ot.gen_step_grads(d, g, z, cond, 7)
    l1, g1 = ot.gen_step_grads(d, g, z, cond, 7, gates=gates)
    assert abs(float(l0) - float(l1)) < 1e-14
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= 1e-13 * float(a.abs().max() + 1e-30)
    # and a different pattern really is a different function
    flipped = ([~t for t in gates[0]], gates[1])
    _, g2 = ot.gen_step_grads(d, g, z, cond, 7, gates=flipped)
    assert float((g2[2] - g0[2]).abs().max()) > 1e-3 * float(g0[2].abs().max())
