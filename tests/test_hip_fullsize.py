"""-m gpu: the kernel variants and configurations that only occur at BASELINE sizes.

* B = 96 at ndomain 16 with DEFAULT options is the smallest batch at which the launcher picks what the bs = 256 step
  runs -- the 256-row weight-gradient tile ``k_wgrad_gemm_ws<256, 64>`` (``B * L >= 65536``, rdgan_api.hip
  ``wgrad_tiling``), the unforced 256x64 conv tile and the automatic K splits -- and is still small enough for the fp64
  torch oracle, so both step gradients are compared with it directly.
* BASELINE configs[2] (bs = 2048, n_critic = 5), configs[3]'s per-rank shard (bs = 1024) and configs[4]'s per-rank
  shard (ndomain 64, bs = 64) are too big for the oracle: size-independent properties instead (mass conservation,
  batch independence against a small-batch run, run-to-run determinism, finite losses through a whole iteration).
* ndomain 64: the generator-step gradients (206 M-parameter Dense weight gradient included) against the oracle at B = 1.
"""
import numpy as np
import pytest
import torch

from oracle import rdgan_torch as ot
from pr_disagg_radar_gan_amd import Engine
from pr_disagg_radar_gan_amd import weights as W
from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer
from tests.hip_util import dev, rel_err, gen_step_on_engine_branch, critic_step_on_engine_branch
from tests.test_hip_step import _params, _t64, _grad_errors, TIGHT

pytestmark = pytest.mark.gpu


def test_b96_default_options_step_gradients_vs_oracle():
    B = 96
    eng = Engine(ndomain=16, max_batch=B)
    try:
        g, d = _params(16, 91)
        gs, ds = eng.to_slab(g), eng.to_slab(d)
        x, cond, z = ot.synthetic_batch(B, 16, 191)
        ref = ot.generator_forward(_t64(g), torch.from_numpy(z).double(), torch.from_numpy(cond).double()).numpy()
        out = eng.gen_forward(gs, dev(z), dev(cond)).cpu().numpy()
        np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-7)          # north_star tolerance
        assert rel_err(out, ref) < 2e-5
        # critic step: compared directly (observed 1e-7...8e-7).  Generator step: at this batch some of the ~2e8 LeakyReLU
        # inputs always sit within fp32 rounding of zero and take the other slope in fp64, which moves gradients by ~1e-3
        # whatever the batch (the number of flips and the gradient norm both scale as sqrt(B)); the oracle therefore
        # differentiates the branch the fp32 run took (its slope pattern, read back from the workspace) -- the same smooth
        # function on both sides, so the comparison is tight at any size.
        x, cond, z = ot.synthetic_batch(B, 16, 300)
        slab, losses, grads = critic_step_on_engine_branch(eng, ds, gs, d, g, x, cond, z, 4711)
        n = eng.n_critic
        np.testing.assert_allclose(slab[n:n + 4], losses.numpy(), rtol=2e-4, atol=1e-6)
        assert slab[n + 4] == 0.0
        errs = _grad_errors(slab[:n], grads, eng.critic_shapes)
        print("critic grad rel errors:", {k: float(f"{v:.2e}") for k, v in errs.items()})
        assert max(errs.values()) < TIGHT, errs
        slab, loss, grads = gen_step_on_engine_branch(eng, ds, gs, d, g, z, cond, 4712)
        n = eng.n_gen
        np.testing.assert_allclose(slab[n], loss.item(), rtol=2e-4, atol=1e-6)
        errs = _grad_errors(slab[:n], grads, eng.gen_shapes)
        print("gen grad rel errors:", {k: float(f"{v:.2e}") for k, v in errs.items()})
        assert max(errs.values()) < TIGHT, errs
    finally:
        eng.close()


def test_nd64_gen_step_gradients_vs_oracle():
    """largedomain variant (L:59,325,335), B = 1: generator-step gradients incl. the Dense kernel (4196 x 49152) and the
    shared-centre backward at ndomain 64, against the fp64 oracle differentiating the fp32 run's LeakyReLU branch (see the
    B = 96 test)."""
    eng = Engine(ndomain=64, max_batch=1)
    try:
        g, d = _params(64, 15)
        gs, ds = eng.to_slab(g), eng.to_slab(d)
        x, cond, z = ot.synthetic_batch(1, 64, 8)
        slab, loss, grads = gen_step_on_engine_branch(eng, ds, gs, d, g, z, cond, 6)
        n = eng.n_gen
        np.testing.assert_allclose(slab[n], loss.item(), rtol=2e-4, atol=1e-6)
        assert slab[n + 4] == 0.0
        errs = _grad_errors(slab[:n], grads, eng.gen_shapes)
        print("nd64 gen-step grad rel errors:", {k: float(f"{v:.2e}") for k, v in errs.items()})
        assert max(errs.values()) < TIGHT, errs
    finally:
        eng.close()


def _fullsize_properties(nd, B, n_critic, probe, small, bf16=0):
    """bf16 = 1: the bf16 storage mode, the mode BASELINE configs[2..4] are quoted in (bench.py --config 3|4|5).
    Properties that hold at any size: softmax mass conservation, batch independence of sample `probe` against a run of
    `small` samples around it, bit-identical repeats (no atomics anywhere on the path), finite gradient slabs with a clear
    non-finite flag, and one whole training iteration (n_critic critic updates + 1 generator update) with finite losses
    that really moved both weight slabs."""
    eng = Engine(ndomain=nd, max_batch=B)
    try:
        if bf16:
            eng.set_option("bf16", 1)
        g, d = _params(nd, 16)
        gs, ds = eng.to_slab(g), eng.to_slab(d)
        x, cond, z = ot.synthetic_batch(B, nd, 9)
        xd, cd, zd = dev(x), dev(cond), dev(z)
        out = eng.gen_forward(gs, zd, cd)
        o = out.cpu().numpy()
        assert o.shape == (B, 24, nd, nd, 1) and np.all(np.isfinite(o)) and o.min() >= 0
        np.testing.assert_allclose(o.sum(axis=1), 1.0, atol=3e-6)
        lo = max(0, probe - small // 2); hi = lo + small
        part = eng.gen_forward(gs, dev(z[lo:hi]), dev(cond[lo:hi])).cpu().numpy()
        # tile / split-K choices depend on the batch size: fp32 sums round differently (2e-5); in the bf16 storage mode such a
        # difference can move a stored activation by one bf16 ulp (2^-8 relative), which the later layers carry along
        if bf16:
            assert rel_err(part, o[lo:hi]) < 2e-2
        else:
            np.testing.assert_allclose(part, o[lo:hi], rtol=2e-5, atol=1e-8)
        again = eng.gen_forward(gs, zd, cd)
        assert torch.equal(out, again)
        c1 = eng.critic_grad(ds, gs, xd, cd, zd, 31337).clone()
        c2 = eng.critic_grad(ds, gs, xd, cd, zd, 31337)
        assert torch.equal(c1, c2)
        g1 = eng.gen_grad(ds, gs, zd, cd, 31338).clone()
        g2 = eng.gen_grad(ds, gs, zd, cd, 31338)
        assert torch.equal(g1, g2)
        for slab, n in ((c1, eng.n_critic), (g1, eng.n_gen)):
            s = slab.cpu().numpy()
            assert np.all(np.isfinite(s)) and s[n + 4] == 0
            assert np.abs(s[:n]).max() > 0
        # critic loss parts as Keras reports them: valid = mean(-D(real)), fake = mean(D(fake)), total = sum with 10 gp
        t = c1[eng.n_critic:eng.n_critic + 4].cpu().numpy()
        np.testing.assert_allclose(t[0], t[1] + t[2] + 10.0 * t[3], rtol=1e-5, atol=1e-6)
        if bf16 and B >= 171:
            # from 171 samples on the column GEMM of the critic's input gradient takes the resident-tile kernel
            # (conv16_resident_ok accepts its plan): same products in the same order as the streaming kernel
            eng.set_option("resident", 0)
            c3 = eng.critic_grad(ds, gs, xd, cd, zd, 31337)
            eng.set_option("resident", 1)
            assert torch.equal(c1, c3)
        tr = WGANGPTrainer(eng, g, d, n_disc=n_critic)
        g_before, d_before = tr.gparams.clone(), tr.dparams.clone()
        d_loss, g_loss, bad = tr.iteration([(xd, cd, zd)] * n_critic, (zd, cd))
        assert float(bad) == 0 and np.isfinite(float(d_loss)) and np.isfinite(float(g_loss))
        assert tr.t == n_critic + 1
        assert not torch.equal(tr.gparams, g_before) and not torch.equal(tr.dparams, d_before)
        assert bool(torch.isfinite(tr.gparams).all()) and bool(torch.isfinite(tr.dparams).all())
    finally:
        eng.close()


@pytest.mark.parametrize("bf16", [0, 1])
def test_config2_bs2048_ncritic5_properties(bf16):
    """BASELINE configs[2]: ndomain 16, bs = 2048, n_critic = 5 (30 GiB workspace, > 2 GiB tensors); quoted in bf16."""
    _fullsize_properties(16, 2048, 5, probe=1777, small=4, bf16=bf16)


@pytest.mark.parametrize("bf16", [0, 1])
def test_config3_shard_bs1024_properties(bf16):
    """BASELINE configs[3]: global bs 8192 over 8 GPUs = 1024 per rank; quoted in bf16."""
    _fullsize_properties(16, 1024, 5, probe=1000, small=3, bf16=bf16)


@pytest.mark.parametrize("bf16", [0, 1])
def test_config4_shard_nd64_bs64_properties(bf16):
    """BASELINE configs[4]: ndomain 64, global bs 512 over 8 GPUs = 64 per rank; quoted in bf16."""
    _fullsize_properties(64, 64, 5, probe=41, small=2, bf16=bf16)
