"""-m gpu: the extra-condition variants of revision 1: condition with 2 channels (daily sum +
longitude index, revision1/additional_inputs/gan_train_cwgangp_pixelnorm_lon.py:136) or 3 (daily sum + sin/cos of
the day of year, …_doy.py:135).  Only the generator's Dense width (356 -> 612 / 868) and the critic's first
Conv3D (C_in 2 -> 3 / 4) change; forward passes and both step gradients are compared with the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import rdgan_torch as ot
from pr_disagg_radar_gan_amd import Engine
from pr_disagg_radar_gan_amd import weights as W
from tests.hip_util import dev, rel_err, gen_step_on_engine_branch, critic_step_on_engine_branch
from tests.test_hip_step import TIGHT, _grad_errors, _t64

pytestmark = pytest.mark.gpu


def _params(nd, nc, seed, bias_scale=0.05):
    rng = np.random.default_rng(seed)
    g = W.init_generator(rng, nd, nc)
    d = W.init_critic(rng, nd, nc)
    g = [p if p.ndim > 1 else (bias_scale * rng.standard_normal(p.shape)).astype(np.float32) for p in g]
    d = [p if p.ndim > 1 else (bias_scale * rng.standard_normal(p.shape)).astype(np.float32) for p in d]
    return g, d


def _batch(B, nd, nc, seed):
    """synthetic tiles + the extra channels as the reference builds them: constant planes per sample"""
    x, cond, z = ot.synthetic_batch(B, nd, seed)
    r = np.random.default_rng(seed + 1000)
    if nc == 2:
        extra = [r.uniform(0, 1, B)]                                   # normalised longitude index
    else:
        doy = r.integers(1, 366, B)
        extra = [np.sin(2 * np.pi * doy / 365), np.cos(2 * np.pi * doy / 365)]
    planes = [np.broadcast_to(e.astype(np.float32)[:, None, None, None], (B, nd, nd, 1)) for e in extra]
    return x, np.ascontiguousarray(np.concatenate([cond] + planes, axis=-1)), z


@pytest.fixture(scope="module", params=[2, 3])
def eng(request):
    e = Engine(ndomain=16, max_batch=8, n_cond_channels=request.param)
    yield e
    e.close()


def test_param_counts(eng):
    nc = eng.n_cond_channels
    assert eng.n_gen == W.param_count(W.gen_param_shapes(16, nc))
    assert eng.n_critic == W.param_count(W.critic_param_shapes(16, nc))
    assert eng.gen_shapes[0][1] == (100 + 256 * nc, 3072)               # Dense width 612 / 868
    assert eng.critic_shapes[0][1] == (3, 3, 3, 1 + nc, 64)


@pytest.mark.parametrize("B", [1, 5])
def test_generator_forward(eng, B):
    nc = eng.n_cond_channels
    g, _ = _params(16, nc, 21)
    x, cond, z = _batch(B, 16, nc, 3)
    ref = ot.generator_forward(_t64(g), torch.from_numpy(z).double(), torch.from_numpy(cond).double()).numpy()
    out = eng.gen_forward(eng.to_slab(g), dev(z), dev(cond)).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-7)          # north_star tolerance: 1e-4 relative, fp32
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("seed", [0, 77])
def test_critic_forward(eng, seed):
    nc = eng.n_cond_channels
    _, d = _params(16, nc, 22)
    B = 4
    x, cond, z = _batch(B, 16, nc, 5)
    masks = ot.critic_masks(seed, B, 16, torch.float64)
    ref = ot.critic_forward(_t64(d), torch.from_numpy(x).double(), torch.from_numpy(cond).double(), masks).numpy()
    out = eng.critic_forward(eng.to_slab(d), dev(x), dev(cond), seed=seed).cpu().numpy()
    assert rel_err(out, ref) < 2e-5


def test_critic_step_grads(eng):
    """one seeded batch against the fp64 oracle on the engine's own LeakyReLU branch (tests/test_hip_step.py)"""
    nc = eng.n_cond_channels
    g, d = _params(16, nc, 23)
    B, seed = 3, 1234
    x, cond, z = _batch(B, 16, nc, 100)
    slab, losses, grads = critic_step_on_engine_branch(eng, eng.to_slab(d), eng.to_slab(g), d, g, x, cond, z, seed)
    n = eng.n_critic
    np.testing.assert_allclose(slab[n:n + 4], losses.numpy(), rtol=2e-4, atol=1e-6)
    assert slab[n + 4] == 0.0
    errs = _grad_errors(slab[:n], grads, eng.critic_shapes)
    assert max(errs.values()) < TIGHT, errs


def test_gen_step_grads(eng):
    nc = eng.n_cond_channels
    g, d = _params(16, nc, 24)
    B, seed = 2, 4321
    x, cond, z = _batch(B, 16, nc, 100)
    slab, loss, grads = gen_step_on_engine_branch(eng, eng.to_slab(d), eng.to_slab(g), d, g, z, cond, seed)
    n = eng.n_gen
    np.testing.assert_allclose(slab[n], float(loss), rtol=2e-4, atol=1e-6)
    errs = _grad_errors(slab[:n], grads, eng.gen_shapes)
    assert max(errs.values()) < TIGHT, errs


def test_bf16_storage_step_grads(eng):
    """The bf16 storage mode with extra condition channels: the first critic layer then runs through the tiled kernels
    (k_conv_gemm<..., BK = 8, OUT16> forward / second sweep on 4 floats per voxel, k_wgrad_gemm<..., DY16> weight gradient)
    instead of the K = 64 edge kernels.  Both step gradients against the fp64 oracle on the run's own branch, at the
    tolerances of tests/test_hip_bf16.py."""
    from tests.test_hip_bf16 import GRAD_TOL
    nc = eng.n_cond_channels
    g, d = _params(16, nc, 26)
    B = 5
    x, cond, z = _batch(B, 16, nc, 101)
    gs, ds = eng.to_slab(g), eng.to_slab(d)
    eng.set_option("bf16", 1)
    try:
        fake = eng.gen_forward(gs, dev(z), dev(cond)).cpu().numpy()
        slab, losses, grads = critic_step_on_engine_branch(eng, ds, gs, d, g, x, cond, z, 77, mode="bf16", fake=fake)
        n = eng.n_critic
        np.testing.assert_allclose(slab[n:n + 4], losses.numpy(), rtol=5e-2, atol=5e-3)
        errs = _grad_errors(slab[:n], grads, eng.critic_shapes)
        print(f"nc {nc} bf16 critic-step grad rel errors:", {k: float(f"{v:.2e}") for k, v in errs.items()})
        assert max(errs.values()) < GRAD_TOL, errs
        slab, loss, grads = gen_step_on_engine_branch(eng, ds, gs, d, g, z, cond, 78, mode="bf16")
        errs = _grad_errors(slab[:eng.n_gen], grads, eng.gen_shapes)
        assert max(errs.values()) < GRAD_TOL, errs
    finally:
        eng.set_option("bf16", 0)


def test_critic_form_cache_survives_another_critic_on_the_shared_engine(eng):
    """with extra condition channels (CP != Cin) the fp32 path of rdgan_critic_forward rewrites the padded
    layer-1 kernel W1P from ITS weights.  A trainer that vouches for its critic slab (content version) between its critic and
    generator steps must not find its cached forms still marked valid after another Critic predicted on the shared engine
    (models.get_engine): the generator step would silently use the other critic's first layer."""
    from pr_disagg_radar_gan_amd.engine import new_version
    nc = eng.n_cond_channels
    g, dA = _params(16, nc, 61)
    _, dB = _params(16, nc, 62)
    gs, dsA, dsB = eng.to_slab(g), eng.to_slab(dA), eng.to_slab(dB)
    x, cond, z = _batch(4, 16, nc, 63)
    want = eng.gen_grad(dsA, gs, dev(z), dev(cond), 7).clone()                   # versions 0: forms rebuilt from dsA
    vg, vA, vB = new_version(), new_version(), new_version()
    eng.critic_grad(dsA, gs, dev(x), dev(cond), dev(z), 5, gen_version=vg, critic_version=vA)       # caches (dsA, vA)
    builds = eng.form_builds()[1]
    eng.critic_forward(dsB, dev(x), dev(cond), critic_version=vB)                 # another model on the same engine
    got = eng.gen_grad(dsA, gs, dev(z), dev(cond), 7, gen_version=vg, critic_version=vA)
    assert torch.equal(got, want)
    # 2 condition channels: C_in 3 is padded to CP 4, the forward rewrote W1P, so the forms were rebuilt, not trusted;
    # 3 channels: C_in = CP = 4, no padded copy exists, the fp32 forward touches no form and the cache rightly stays valid
    rebuilt = 1 if nc == 2 else 0
    assert eng.form_builds()[1] == builds + rebuilt
    again = eng.gen_grad(dsA, gs, dev(z), dev(cond), 7, gen_version=vg, critic_version=vA)
    assert torch.equal(again, want) and eng.form_builds()[1] == builds + rebuilt  # ... and are cached (again) afterwards


def test_cond_shape_checked(eng):
    g, _ = _params(16, eng.n_cond_channels, 25)
    x, cond, z = ot.synthetic_batch(2, 16, 1)                           # one-channel condition: wrong for this engine
    with pytest.raises(ValueError):
        eng.gen_forward(eng.to_slab(g), dev(z), dev(cond))


@pytest.mark.parametrize("kind", ["lon", "doy"])
def test_device_dataset_extra_condition(kind):
    """DeviceDataset.gather appends the planes the reference concatenates (…_lon.py:175-184, …_doy.py:173-186)."""
    from pr_disagg_radar_gan_amd.data_pipeline import DeviceDataset
    r = np.random.default_rng(5)
    n_days, ny, nx, nd = 6, 40, 48, 16
    data = r.gamma(2.0, 1.0, (n_days, 24, ny, nx)).astype(np.float32)
    idx = np.stack([r.integers(0, n_days, 20), r.integers(0, ny - nd, 20), r.integers(0, nx - nd, 20)], 1)
    ds = DeviceDataset(data, idx, ndomain=nd)
    doy = r.integers(1, 366, n_days)
    lo, hi = idx[:, 2].min(), idx[:, 2].max()
    if kind == "lon":
        ds.set_extra_condition("lon", min_lonidx=lo, max_lonidx=hi)
        want = [(idx[:, 2] - lo) / hi]
    else:
        ds.set_extra_condition("doy", timelist=doy)
        want = [np.sin(2 * np.pi * doy[idx[:, 0]] / 365), np.cos(2 * np.pi * doy[idx[:, 0]] / 365)]
    batch, cond = ds.gather(np.arange(20))
    cond = cond.cpu().numpy()
    assert cond.shape == (20, nd, nd, ds.n_cond_channels) and ds.n_cond_channels == 1 + len(want)
    ds.set_extra_condition(None)
    _, cond1 = ds.gather(np.arange(20))
    np.testing.assert_array_equal(cond[..., :1], cond1.cpu().numpy())
    for k, w in enumerate(want):
        np.testing.assert_allclose(cond[..., 1 + k], np.broadcast_to(w.astype(np.float32)[:, None, None], (20, nd, nd)),
                                   rtol=0, atol=1e-7)
