"""Helpers for the -m gpu parity tests: call the C ABI with torch CUDA tensors."""
import ctypes

import numpy as np
import torch

from pr_disagg_radar_gan_amd import _lib


def dev(a, dtype=torch.float32):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    return a.to(dtype).contiguous().cuda()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def lib():
    return _lib.load()     # raises loudly if the HIP library is missing


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def hip_gates(eng, B):
    """LeakyReLU slope pattern (True = slope 1) of the engine's last generator-step forward: generator h0..h3 and critic
    layers 1..4 over B samples, from the activations left in the workspace (rdgan_debug_activation).  A dropped critic
    activation reads 0 -> False, which is immaterial: its gradient is multiplied by the zero mask anyway."""
    from oracle import rdgan_np as onp
    nd = eng.ndomain
    s = nd // 8
    gshape = [(B, 3, s, s, 256), (B, 6, 2 * s, 2 * s, 256), (B, 12, 4 * s, 4 * s, 128), (B, 24, 8 * s, 8 * s, 64)]
    geo = onp.critic_geometry(nd)
    chans = (64, 128, 256, 256)
    g = [eng.debug_activation(i, shp).cpu() > 0 for i, shp in enumerate(gshape)]
    d = [eng.debug_activation(4 + li, (B,) + tuple(geo[li][1]) + (chans[li],)).cpu() > 0 for li in range(4)]
    return g, d


def hip_critic_gates(eng, B):
    """The same for the engine's last CRITIC step: the four critic layers over the 3B batch [real; fake; interpolated].
    Needs the option "keep_gates" set before the step (the penalty's second sweep overwrites the interpolated third of the
    activations in place; the option keeps a copy)."""
    from oracle import rdgan_np as onp
    geo = onp.critic_geometry(eng.ndomain)
    chans = (64, 128, 256, 256)
    return [eng.debug_activation(4 + li, (3 * B,) + tuple(geo[li][1]) + (chans[li],)).cpu() > 0 for li in range(4)]


# how far from a LeakyReLU kink (in RMS of the layer's inputs) the engine's slope decision may differ from the fp64 oracle's,
# and for what share of a layer: fp32 rounding of sums of ~1e3..1e4 terms; bf16 storage: every stored tensor rounded to 2^-9.
# Round 4: set to 3x the largest values the whole -m gpu session meets (tests/conftest.py prints them and leaves them in
# gpurun_out/gate_observed.json): fp32 margin 7.8e-7 / fraction 2.0e-5, bf16 margin 2.63e-2 / fraction 1.70e-3 (round 3 allowed
# 2e-4 / 1e-4 and 0.15 / 3e-2 without knowing the margins).
GATE_TOL = {"f32": dict(max_margin=2.5e-6, max_fraction=6e-5), "bf16": dict(max_margin=8e-2, max_fraction=5e-3)}
# the largest margin / fraction any test of this process has met, per mode (printed by every helper call, and once more by
# tests/test_hip_bf16.py::test_zz_gate_guard_headroom): the tolerances above are meant to stay within 3x of these
GATE_OBSERVED = {"f32": {}, "bf16": {}}


def _gate_report(mode, what):
    o = GATE_OBSERVED[mode]
    print(f"gate guard [{mode}] after {what}: worst margin {o.get('margin', 0.0):.3e} (limit {GATE_TOL[mode]['max_margin']:.1e}), "
          f"worst fraction {o.get('fraction', 0.0):.3e} (limit {GATE_TOL[mode]['max_fraction']:.1e})")


def gen_step_on_engine_branch(eng, ds, gs, d, g, z, cond, seed, mode="f32"):
    """Generator-step gradient slab of the engine and the fp64 oracle's (loss, grads) on the LeakyReLU branch the engine
    took, after checking that the engine's branch differs from the oracle's own only at the kinks."""
    from oracle import rdgan_torch as ot
    B = z.shape[0]
    slab = eng.gen_grad(ds, gs, dev(z), dev(cond), seed).cpu().numpy()
    gates = hip_gates(eng, B)
    t64 = lambda arrs: [torch.from_numpy(a).double() for a in arrs]
    loss, grads, (gh, dh) = ot.gen_step_grads(t64(d), t64(g), torch.from_numpy(z).double(), torch.from_numpy(cond).double(),
                                              seed, gates=gates, return_intermediates=True)
    ot.check_gates(gates[0], gh, None, observed=GATE_OBSERVED[mode], **GATE_TOL[mode])
    ot.check_gates(gates[1], dh, ot.critic_masks(seed, B, eng.ndomain, torch.float64), observed=GATE_OBSERVED[mode], **GATE_TOL[mode])
    _gate_report(mode, f"generator step nd{eng.ndomain} B{B}")
    return slab, loss, grads


def critic_step_on_engine_branch(eng, ds, gs, d, g, x, cond, z, seed, mode="f32", fake=None, alpha_offset=0):
    """The same for the critic step (gradient penalty double backward included): needs "keep_gates"; `fake` = the generator
    output the engine fed its critic (a constant of this step), for runs whose forward pass is not fp32-exact."""
    from oracle import rdgan_torch as ot
    B = x.shape[0]
    eng.set_option("keep_gates", 1)
    try:
        slab = eng.critic_grad(ds, gs, dev(x), dev(cond), dev(z), seed).cpu().numpy()
        gates = hip_critic_gates(eng, B)
    finally:
        eng.set_option("keep_gates", 0)
    t64 = lambda arrs: [torch.from_numpy(a).double() for a in arrs]
    losses, grads, dh = ot.critic_step_grads(t64(d), t64(g), torch.from_numpy(x).double(), torch.from_numpy(cond).double(),
                                             torch.from_numpy(z).double(), seed, alpha_offset=alpha_offset, gates=gates,
                                             fake=None if fake is None else torch.from_numpy(np.asarray(fake)).double(),
                                             return_intermediates=True)
    ot.check_gates(gates, dh, ot.critic_masks(seed, 3 * B, eng.ndomain, torch.float64), observed=GATE_OBSERVED[mode], **GATE_TOL[mode])
    _gate_report(mode, f"critic step nd{eng.ndomain} B{B}")
    return slab, losses, grads
