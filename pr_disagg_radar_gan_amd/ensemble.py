"""Batched ensemble inference and on-device CRPS: the body of the reference's
generate_and_evaluate_crps.py loop (:177-195) -- n_fake_per_real scenarios for one real day, CRPS of the
ensemble against the observed hourly field per grid point, area mean per hour -- without leaving the GPU."""
import ctypes

import numpy as np
import torch

from . import _lib, models
from . import weights as W


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def crps_ensemble_device(ens, obs, scale=None):
    """ens (n, ...) float32 CUDA, obs (...) -> crps (...): properscoring.crps_ensemble(obs, ens, axis=0)."""
    lib = _lib.load()
    ens = ens.contiguous(); obs = obs.contiguous()
    n = ens.shape[0]
    npix = obs.numel()
    if ens.numel() != n * npix:
        raise ValueError("ens must have shape (n,) + obs.shape")
    out = torch.empty_like(obs)
    st = ctypes.c_void_p(torch.cuda.current_stream(ens.device).cuda_stream)
    _lib.check(lib.rdgan_crps_ensemble(_p(ens), _p(obs), _p(scale.contiguous() if scale is not None else None), _p(out),
                                       n, npix, st), None, "rdgan_crps_ensemble")
    return out


def generate_ensemble_device(gen, cond_norm, n_members, chunk=1024, seed=None, latent=None):
    """n_members generator samples for ONE normalised condition (nd,nd,1); returns the (n,24,nd,nd) fractions on the
    device.  Latent noise from the global numpy RNG (as reference :183) unless a torch seed is given, or `latent`
    (n_members, 100; numpy or a tensor) supplies it -- the reference's same-noise comparison, generate_and_evaluate.py:551-560."""
    nd = gen.ndomain
    if latent is not None:
        latent = torch.as_tensor(latent, dtype=torch.float32)
        if tuple(latent.shape) != (n_members, W.LATENT_DIM):
            raise ValueError(f"latent must have shape ({n_members}, {W.LATENT_DIM}), got {tuple(latent.shape)}")
    eng = models.get_engine(nd, min(chunk, n_members))
    slab = gen.device_slab(eng)
    cond_t = torch.from_numpy(np.ascontiguousarray(cond_norm, dtype=np.float32).reshape(1, nd, nd, 1)).to(eng.device)
    out = torch.empty((n_members, W.NHOURS, nd, nd, 1), dtype=torch.float32, device=eng.device)
    g = None
    if seed is not None:
        g = torch.Generator(device=eng.device); g.manual_seed(int(seed))
    for i in range(0, n_members, eng.max_batch):
        m = min(eng.max_batch, n_members - i)
        if latent is not None:
            z = latent[i:i + m].to(eng.device).contiguous()
        elif g is None:
            z = torch.from_numpy(np.random.normal(size=(m, W.LATENT_DIM)).astype(np.float32)).to(eng.device)
        else:
            z = torch.randn((m, W.LATENT_DIM), generator=g, device=eng.device)
        eng.gen_forward(slab, z, cond_t.expand(m, nd, nd, 1).contiguous(), out=out[i:i + m])
        eng.check_numerics()                     # reference T:349-350
    return out.view(n_members, W.NHOURS, nd, nd)


def generate_same_noise_pair(gen, cond1_norm, cond2_norm, n_members=1000, latent=None):
    """The reference's condition-sensitivity experiment (generate_and_evaluate.py:551-560): ONE latent block (n_members, 100)
    drawn once from the global numpy RNG (:550), pushed through the generator under two different normalised conditions, so that
    differences between the two ensembles come from the condition alone.  Returns (fractions1, fractions2, latent), the
    fractions (n,24,nd,nd) on the device."""
    if latent is None:
        latent = np.random.normal(size=(n_members, W.LATENT_DIM)).astype(np.float32)
    f1 = generate_ensemble_device(gen, cond1_norm, n_members, latent=latent)
    f2 = generate_ensemble_device(gen, cond2_norm, n_members, latent=latent)
    return f1, f2, latent


def crps_for_day(gen, real_precip, n_fake_per_real=1000, norm_scale=W.NORM_SCALE, seed=None):
    """One iteration of the reference's CRPS loop (:177-190): real_precip (24,nd,nd) mm/h -> area-mean CRPS per hour."""
    real = torch.from_numpy(np.ascontiguousarray(real_precip, dtype=np.float32))
    dsum = real.sum(0)                                            # reals_dsum, :168
    cond = (dsum / norm_scale).numpy()[..., None]
    ens = generate_ensemble_device(gen, cond, n_fake_per_real, seed=seed)       # fractions
    dev = ens.device
    scale = (dsum / norm_scale * norm_scale).to(dev)             # generated * cond * norm_scale, :186
    scale = scale.unsqueeze(0).expand(W.NHOURS, -1, -1).contiguous()
    crps = crps_ensemble_device(ens, real.to(dev), scale)
    return crps.mean(dim=(1, 2)).cpu().numpy()                    # crps_areamean, :190
