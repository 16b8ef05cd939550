"""torch-CPU restatement of the cWGAN-GP training step (autograd gives the gradients).

TEST INFRASTRUCTURE (see oracle/__init__.py; parity unpinned).  Second, independent
restatement of the reference arithmetic: forward ops through torch.nn.functional with
explicit TF-style padding, gradients (incl. the gradient-penalty double backward of
GradientPenalty.call, T:238-241) through torch.autograd.  Must agree with oracle/rdgan_np.py
on the forward pass and with fp64 finite differences on gradients
(tests/test_oracle_*.py).  Also timed on the host cores as bench.py's ``cpu_baseline``
("port": TensorFlow, the reference's runtime, is not installed).

T = gan_train_cwgangp_pixelnorm.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import rdgan_np as onp
from . import rng as orng

LRELU = onp.LRELU_ALPHA


def _conv3d_tf(x, w, b, stride, pad_before, out_dims):
    """x (B,D,H,W,Cin) NDHWC, w (3,3,3,Cin,Cout) TF layout -> (B,Do,Ho,Wo,Cout)."""
    B, D, H, W, Cin = x.shape
    need = [(o - 1) * stride + 3 for o in out_dims]
    after = [max(nd - n - p, 0) for nd, n, p in zip(need, (D, H, W), pad_before)]
    xc = x.permute(0, 4, 1, 2, 3)
    xc = F.pad(xc, (pad_before[2], after[2], pad_before[1], after[1], pad_before[0], after[0]))
    wc = w.permute(4, 3, 0, 1, 2)
    y = F.conv3d(xc, wc, b, stride=stride)
    y = y[:, :, :out_dims[0], :out_dims[1], :out_dims[2]]
    return y.permute(0, 2, 3, 4, 1)


def pixel_norm(x):
    """T:255-266."""
    return x / torch.sqrt(torch.mean(x * x, dim=-1, keepdim=True) + onp.PIXELNORM_EPS)


def upsample3d(x):
    """T:330."""
    return x.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)


def _lrelu(x, gate=None):
    """LeakyReLU(0.2) (T:327,333,288).  gate (bool tensor, True = slope 1): differentiate the piecewise-linear branch an
    external run took instead of deciding by the sign of x -- the loss is only piecewise smooth, and an fp32 run and this
    oracle legitimately pick different slopes for an input within rounding of zero (see kink_margin); with the fp32 run's
    own pattern both sides evaluate the same smooth function."""
    if gate is None:
        return F.leaky_relu(x, LRELU)
    return x * torch.where(gate, torch.ones((), dtype=x.dtype), torch.full((), LRELU, dtype=x.dtype))


def generator_forward(p, z, cond, return_intermediates=False, gates=None):
    """T:312-357.  gates: optional [h0, h1, h2, h3] slope patterns for _lrelu."""
    Wd, bd, W1, b1, W2, b2, W3, b3, W4, b4 = p
    B = z.shape[0]
    nd = cond.shape[1]
    s = nd // 8
    x = torch.cat([z, cond.reshape(B, -1)], dim=1)
    gt = gates if gates is not None else [None] * 4
    h0 = _lrelu((x @ Wd + bd).reshape(B, 3, s, s, 256), gt[0])
    hs = [h0]
    h = h0
    for li, (W, b) in enumerate(((W1, b1), (W2, b2), (W3, b3))):
        u = upsample3d(h)
        y = _conv3d_tf(u, W, b, 1, (1, 1, 1), u.shape[1:4])
        h = _lrelu(pixel_norm(y), gt[li + 1])
        hs.append(h)
    logits = _conv3d_tf(h, W4, b4, 1, (1, 1, 1), h.shape[1:4])
    out = torch.softmax(logits, dim=1)
    if return_intermediates:
        return out, dict(h0=hs[0], h1=hs[1], h2=hs[2], h3=hs[3], logits=logits)
    return out


def critic_forward(p, sample, cond, masks=None, return_intermediates=False, gates=None):
    """T:272-309.  masks: list of 4 tensors (0 or 1/0.75) or None.  gates: optional 4 slope patterns for _lrelu."""
    nd = cond.shape[1]
    geo = onp.critic_geometry(nd)
    cond_rep = cond[:, None].expand(-1, onp.NHOURS, -1, -1, -1)
    x = torch.cat([sample, cond_rep], dim=-1)
    hs = []
    for li in range(4):
        _, out_dims, pad = geo[li]
        a = _conv3d_tf(x, p[2 * li], p[2 * li + 1], 2, pad, out_dims)
        x = _lrelu(a, None if gates is None else gates[li])
        if masks is not None:
            x = x * masks[li]
        hs.append(x)
    v = x.reshape(x.shape[0], -1) @ p[8] + p[9]
    if return_intermediates:
        return v, dict(h=hs)
    return v


def critic_masks(seed, batch, ndomain, dtype=torch.float32):
    """Dropout masks for a critic pass over ``batch`` samples (flat NDHWC index per layer,
    streams D1..D4 of oracle/rng.py).  seed == 0 -> None (dropout off)."""
    if seed == 0:
        return None
    geo = onp.critic_geometry(ndomain)
    chans = (64, 128, 256, 256)
    out = []
    for li in range(4):
        shp = (batch,) + tuple(geo[li][1]) + (chans[li],)
        out.append(torch.from_numpy(orng.dropout_scale_mask(seed, orng.STREAM_D1 + li, shp)).to(dtype))
    return out


def critic_step_grads(dp, gp, x_real, cond, z, seed, alpha_offset=0, gates=None, fake=None, return_intermediates=False):
    """One critic ``train_on_batch`` graph (T:363-392,472) up to the gradients.

    The three critic passes of the reference (T:372,373,379) are evaluated as ONE batch
    [real; fake; interpolated] of 3B samples so that dropout-mask element indices match the
    HIP path; alpha of sample k comes from stream ALPHA at index alpha_offset + k (alpha_offset = the global index of
    this shard's first sample, the HIP option "sample_offset").  Returns (losses[4] = total, valid, fake, gp
    as Keras reports them, grads list in critic weight order).

    gates: optional 4 slope patterns (3B samples each, [real; fake; interpolated]) for the critic's LeakyReLUs, so that this
    oracle differentiates -- twice, for the penalty -- the piecewise-linear branch an external run took (see _lrelu).
    fake: optional generator output to use instead of this oracle's own forward pass: the generator is frozen in this
    step (T:363), its output is a constant input of the graph; a reduced-precision run passes what IT fed its critic.
    return_intermediates: also return the critic's four post-dropout activations of the 3B batch (for check_gates)."""
    B = x_real.shape[0]
    nd = cond.shape[1]
    dt = x_real.dtype
    dp = [t.detach().clone().requires_grad_(True) for t in dp]
    if fake is None:
        with torch.no_grad():
            fake = generator_forward(gp, z, cond)                    # generator frozen, T:363
    else:
        fake = fake.detach().to(dt)
    alpha = torch.from_numpy(orng.uniform(seed, orng.STREAM_ALPHA, B, start=alpha_offset)).to(dt).reshape(B, 1, 1, 1, 1)
    xhat = (alpha * x_real + (1 - alpha) * fake).detach().requires_grad_(True)   # T:221-224
    masks = critic_masks(seed, 3 * B, nd, dt)
    v, inter = critic_forward(dp, torch.cat([x_real, fake, xhat], 0), torch.cat([cond, cond, cond], 0), masks, True, gates=gates)
    v_real, v_fake, v_hat = v[:B], v[B:2 * B], v[2 * B:]
    g, = torch.autograd.grad(v_hat.sum(), xhat, create_graph=True)   # K.gradients, T:240
    gpen = torch.sqrt((g * g).reshape(B, -1).sum(1, keepdim=True)) - 1   # T:241
    l_valid = torch.mean(-1.0 * v_real)                               # W(valid=-1), T:452
    l_fake = torch.mean(1.0 * v_fake)                                 # W(fake=+1),  T:453
    l_gp = torch.mean(gpen * gpen)                                    # 'mse' vs dummy 0, T:390,454
    total = l_valid + l_fake + onp.GP_WEIGHT * l_gp                   # loss_weights, T:392
    grads = torch.autograd.grad(total, dp, allow_unused=True)
    grads = [torch.zeros_like(p) if gg is None else gg for p, gg in zip(dp, grads)]
    losses = torch.stack([total, l_valid, l_fake, l_gp]).detach()
    if return_intermediates:
        return losses, [gg.detach() for gg in grads], [hh.detach() for hh in inter["h"]]
    return losses, [gg.detach() for gg in grads]


def gen_step_grads(dp, gp, z, cond, seed, gates=None, return_intermediates=False):
    """One generator ``train_on_batch`` graph (T:395-408,482): loss = mean(-D(G(z,c))),
    critic frozen but dropout active.  gates: optional (generator [h0..h3], critic [4 layers]) slope patterns.
    return_intermediates: also return (generator h0..h3, critic post-dropout activations) for check_gates."""
    B = z.shape[0]
    nd = cond.shape[1]
    gp = [t.detach().clone().requires_grad_(True) for t in gp]
    img, gi = generator_forward(gp, z, cond, True, gates=None if gates is None else gates[0])
    masks = critic_masks(seed, B, nd, z.dtype)
    v, di = critic_forward(dp, img, cond, masks, True, gates=None if gates is None else gates[1])
    loss = torch.mean(-1.0 * v)
    grads = torch.autograd.grad(loss, gp)
    if return_intermediates:
        return loss.detach(), [gg.detach() for gg in grads], ([gi[k].detach() for k in ("h0", "h1", "h2", "h3")],
                                                              [hh.detach() for hh in di["h"]])
    return loss.detach(), [gg.detach() for gg in grads]


def check_gates(gates, acts, masks=None, max_margin=1e-3, max_fraction=1e-3, observed=None):
    """Guard for the ``gates=`` mechanism: an externally supplied slope pattern may differ from this oracle's own decision
    only where the oracle's LeakyReLU input is within rounding of zero.  gates / acts: matching lists of bool patterns and of
    the post-activation tensors this oracle computed WITH those patterns (x * slope(gate), so sign(x) = sign(act) is the
    oracle's own decision for the inputs it saw); masks: optional dropout masks (elements dropped by the mask are ignored:
    their gradient is zero whatever the slope, and an external run reads them back as 0).  Asserts that disagreements are
    rare (max_fraction of the layer) and all sit within max_margin of the kink, measured as |LeakyReLU input| / RMS of the
    layer.  Returns (worst margin, worst fraction) met over the layers; with `observed` (a dict) the maxima are also
    accumulated there (keys "margin", "fraction"), so that a test run can report the guard's headroom."""
    worst, worst_frac = 0.0, 0.0
    for li, (g, a) in enumerate(zip(gates, acts)):
        live = torch.ones_like(g) if masks is None else (masks[li] != 0)
        pre = a / torch.where(g, torch.ones((), dtype=a.dtype), torch.full((), LRELU, dtype=a.dtype))
        if masks is not None:
            pre = pre / torch.where(live, masks[li], torch.ones((), dtype=a.dtype))
        own = pre > 0
        rms = float(pre[live].pow(2).mean().sqrt())
        bad = (g != own) & live
        nbad = int(bad.sum())
        if nbad:
            frac = nbad / max(1, int(live.sum()))
            margin = float(pre[bad].abs().max()) / rms
            worst, worst_frac = max(worst, margin), max(worst_frac, frac)
            if observed is not None:
                observed["margin"] = max(observed.get("margin", 0.0), margin)
                observed["fraction"] = max(observed.get("fraction", 0.0), frac)
            assert frac <= max_fraction, f"layer {li}: {nbad} slope disagreements ({frac:.2e} of the layer)"
            assert margin <= max_margin, f"layer {li}: slope disagreement {margin:.2e} RMS away from the kink"
    return worst, worst_frac


def adam_update(params, grads, vs, t, lr=1e-4, beta2=0.9, eps=1e-7):
    """tf.optimizers.Adam(lr=1e-4, beta_1=0, beta_2=0.9) (T:385), Keras optimizer_v2 form:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = g (beta_1 = 0); v = b2*v + (1-b2)*g^2;
    p -= lr_t*m/(sqrt(v)+eps) with eps = 1e-7 (Keras default).  ``t`` is the optimizer's
    shared iteration counter AFTER increment (both models use one optimizer, T:391,408)."""
    lr_t = lr * float(np.sqrt(1.0 - beta2 ** t))
    for p, g, v in zip(params, grads, vs):
        v.mul_(beta2).add_(g * g, alpha=1 - beta2)
        p.sub_(lr_t * g / (torch.sqrt(v) + eps))


class Trainer:
    """Minimal CPU trainer used for the cpu_baseline timing and for end-to-end parity of a
    few iterations (T:466-482: n_critic critic steps then one generator step)."""

    def __init__(self, ndomain=16, seed=0, dtype=torch.float32):
        rng = np.random.default_rng(seed)
        nd = dtype == torch.float64 and np.float64 or np.float32
        self.ndomain = ndomain
        self.gp = [torch.from_numpy(a) for a in onp.init_generator(rng, ndomain, dtype=nd)]
        self.dp = [torch.from_numpy(a) for a in onp.init_critic(rng, ndomain, dtype=nd)]
        self.gv = [torch.zeros_like(p) for p in self.gp]
        self.dv = [torch.zeros_like(p) for p in self.dp]
        self.t = 0

    def critic_step(self, x_real, cond, z, seed):
        losses, grads = critic_step_grads(self.dp, self.gp, x_real, cond, z, seed)
        self.t += 1
        adam_update(self.dp, grads, self.dv, self.t)
        return losses

    def gen_step(self, z, cond, seed):
        loss, grads = gen_step_grads(self.dp, self.gp, z, cond, seed)
        self.t += 1
        adam_update(self.gp, grads, self.gv, self.t)
        return loss


def synthetic_batch(batch, ndomain, seed, dtype=np.float32):
    """Synthetic inputs of SURVEY 8(d): real tiles = softmax over hours of 2*N(0,1) (in [0,1],
    sum over hours 1, as asserted at T:167-172); cond = Gamma(2, 5 mm)/127.4; z ~ N(0,1)."""
    r = np.random.default_rng(seed)
    x = onp.softmax_hours(2.0 * r.standard_normal((batch, 24, ndomain, ndomain, 1))).astype(dtype)
    cond = (r.gamma(2.0, 5.0, (batch, ndomain, ndomain, 1)) / onp.NORM_SCALE).astype(dtype)
    z = r.standard_normal((batch, onp.LATENT_DIM)).astype(dtype)
    return x, cond, z


def kink_margin(dp, gp, x_real, cond, z, seed, critic_step=True):
    """Smallest |LeakyReLU input| (relative to the layer's RMS) met by a step.  The loss is only
    piecewise smooth in the activations: an fp32 run and this fp64 oracle legitimately pick
    different slopes for an activation within rounding of zero, which moves gradients by
    ~1e-3.  Parity tests therefore use inputs whose margin is comfortably above fp32 noise."""
    B = z.shape[0]
    with torch.no_grad():
        fake, inter = generator_forward(gp, z, cond, True)
        m = min(float((inter[k].abs() / inter[k].pow(2).mean().sqrt()).min()) for k in ("h0", "h1", "h2", "h3"))
        if critic_step:
            alpha = torch.from_numpy(orng.uniform(seed, orng.STREAM_ALPHA, B)).to(z.dtype).reshape(B, 1, 1, 1, 1)
            xs = torch.cat([x_real, fake, alpha * x_real + (1 - alpha) * fake], 0)
            cs = torch.cat([cond, cond, cond], 0)
        else:
            xs, cs = fake, cond
        _, di = critic_forward(dp, xs, cs, None, True)
        for h in di["h"]:
            m = min(m, float((h.abs() / h.pow(2).mean().sqrt()).min()))
    return m
