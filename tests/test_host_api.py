"""CPU tests of the host side: C-ABI surface, weight containers, the reference-API mirrors'
plumbing, and the rule that the product path neither imports the oracle nor falls back to CPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import pr_disagg_radar_gan_amd as pkg
from pr_disagg_radar_gan_amd import _lib, models, weights as W
from pr_disagg_radar_gan_amd import raindisagg_gan_pretrained as P
from pr_disagg_radar_gan_amd import gan_train_cwgangp_pixelnorm as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rdgan.h")).read()
    declared = set(re.findall(r"\b(rdgan_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.load()                                   # raises if the .so is missing or lacks a symbol
    for name in declared:
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr)


def test_product_path_never_imports_the_oracle_or_falls_back():
    pkgdir = os.path.dirname(pkg.__file__)
    for fn in os.listdir(pkgdir):
        if fn.endswith(".py"):
            src = open(os.path.join(pkgdir, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(_lib.RdganError):
            pkg.Engine(ndomain=16, max_batch=2)
        g = T.create_generator(seed=0)
        with pytest.raises(_lib.RdganError):
            g.predict([np.zeros((1, 100), np.float32), np.zeros((1, 16, 16, 1), np.float32)])


def test_weight_layouts_and_containers(tmp_path):
    assert W.param_count(W.gen_param_shapes(16)) == 3974273
    assert W.param_count(W.critic_param_shapes(16)) == 2880065
    assert W.param_count(W.gen_param_shapes(64)) == 209168513
    assert W.param_count(W.critic_param_shapes(64)) == 2887745
    rng = np.random.default_rng(0)
    g = W.init_generator(rng, 16)
    flat = W.flatten(g)
    back = W.unflatten(flat, W.gen_param_shapes(16))
    assert all(np.array_equal(a, b) for a, b in zip(g, back))
    assert W.infer_ndomain_from_gen(g) == 16
    assert abs(g[2].std() - 0.02) < 1e-3 and not g[1].any()          # RandomNormal(0.02), zero bias
    d = W.init_critic(rng, 16)
    lim = np.sqrt(6.0 / (27 * 64 + 27 * 128))
    assert abs(np.abs(d[2]).max() - lim) < 1e-3 * lim                 # glorot_uniform limit
    path = str(tmp_path / "gen.npz")
    W.save_weights(path, g, W.gen_param_shapes(16), "generator")
    loaded = W.load_weights(path)
    assert all(np.array_equal(a, b) for a, b in zip(g, loaded))
    with pytest.raises(FileNotFoundError):
        W.load_weights(str(tmp_path / "missing.h5"))
    with pytest.raises(ValueError):
        models.Generator(g[:-1], 16)
    W.save_weights(str(tmp_path / "gen.h5"), g, W.gen_param_shapes(16), "generator")    # dependency-free h5lite writer
    assert all(np.array_equal(a, b) for a, b in zip(g, W.load_weights(str(tmp_path / "gen.h5"))))


class _FakeGen:
    """softmax-like fake generator: uniform fractions 1/24 plus latent-dependent noise, renormalised"""
    inputs = [type("I", (), {"shape": (None, 100)})()]

    def predict(self, inputs):
        latent, cond = inputs
        n = latent.shape[0]
        assert cond.shape == (n, 16, 16, 1) and latent.shape == (n, 100)
        w = 1.0 + 0.1 * np.tanh(latent[:, :24]).astype(np.float32)
        w = w / w.sum(1, keepdims=True)
        return np.broadcast_to(w[:, :, None, None, None], (n, 24, 16, 16, 1)).astype(np.float32).copy()


def test_generate_scenarios_plumbing(monkeypatch):
    monkeypatch.setattr(P, "gen", _FakeGen())
    assert P.norm_scale == 127.4 and P.latent_dim == 100
    assert P.generator_file.endswith("ndomain16_stride16_0020.h5")
    cond = 10 * np.ones((16, 16, 1))                    # example.py:8
    np.random.seed(1)
    a = P.generate_scenarios(cond, 10)
    assert a.shape == (10, 24, 16, 16) and a.dtype == np.float64
    np.testing.assert_allclose(a.sum(axis=1), 10.0, rtol=1e-6)       # mass conservation (softmax over hours)
    np.random.seed(1)
    b = P.generate_scenarios(cond, 10)
    assert np.array_equal(a, b)                          # honours the global numpy RNG (reference :56)
    assert cond.max() == 10                              # input not mutated
    one = P.generate_scenarios(cond, 1)
    assert one.shape == (24, 16, 16)                     # the reference's squeeze quirk (:62)


def test_missing_generator_file_is_a_clear_error(monkeypatch, tmp_path):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(P, "gen", P._LazyGenerator())
    with pytest.raises(FileNotFoundError) as e:
        P.generate_scenarios(np.ones((16, 16, 1)), 2)
    assert "trained_models" in str(e.value)


def test_plot_scenarios_figure():
    import matplotlib
    matplotlib.use("agg")
    sc = np.random.default_rng(0).random((3, 24, 16, 16)) + 0.02
    fig = P.plot_scenarios(sc)
    assert tuple(fig.get_size_inches()) == (24.0, 3.0)
    assert len(fig.axes) == 3 * 24 + 1                   # panels + colorbar axes
    import matplotlib.pyplot as plt
    plt.close(fig)


def test_training_data_plumbing():
    rng = np.random.default_rng(0)
    data = rng.gamma(0.3, 2.0, (6, 24, 40, 40)).astype(np.float32) + 1e-3
    idx = [(t, y, x) for t in range(6) for y in (0, 16) for x in (0, 24)]
    T.configure(ndomain=16)
    T.use_arrays(data, idx)
    assert T.n_samples == len(idx)
    np.random.seed(0)
    batch, cond = next(T.generate_real_samples(5))
    assert batch.shape == (5, 24, 16, 16, 1) and cond.shape == (5, 16, 16, 1)
    np.testing.assert_allclose(batch.sum(axis=1), 1.0, rtol=1e-5)    # fractions of the daily sum
    from oracle import data_np as od
    np.random.seed(5)
    b_host, c_host = T._real_batch(6)
    np.random.seed(5)
    ixs = np.random.randint(T.n_samples, size=6)
    b_ref, c_ref = od.gather_real(data, np.array(idx), ixs, 16)
    assert np.array_equal(b_host, b_ref) and np.array_equal(c_host, c_ref)       # host path == restatement of T:149-166
    assert od.valid_indices(data, 16, 16, 5, 20) == [t for t in od.valid_indices(data, 16, 16, 5, 20)]
    latent, c2 = T.generate_latent_points(4)
    assert latent.shape == (4, 100) and c2.shape == (4, 16, 16, 1)
    assert T.params == "20090101-20161231-tp_thresh_daily5_n_thresh20_ndomain16_stride16"
    T.configure(ndomain=64, n_thresh=40)
    assert T.params.endswith("n_thresh40_ndomain64_stride16")
    T.configure(ndomain=16, n_thresh=20)
    assert T.wasserstein_loss(-np.ones((4, 1)), np.arange(4.).reshape(4, 1)) == -1.5
    pn = T.PixelNormalization()(np.full((1, 1, 1, 1, 8), 3.0))
    np.testing.assert_allclose(pn, 3 / np.sqrt(9 + 1e-8))
    gp = T.GradientPenalty()(np.ones((2, 3, 1, 1, 1)))
    np.testing.assert_allclose(gp, np.sqrt(3) - 1)


def test_extra_condition_variants():
    """n_channel = 2 / 3 (revision1/additional_inputs): shapes, values of the appended planes, config inference."""
    from oracle import data_np as od
    rng = np.random.default_rng(1)
    data = rng.gamma(0.3, 2.0, (6, 24, 40, 48)).astype(np.float32) + 1e-3
    idx = np.array([(t, y, x) for t in range(6) for y in (0, 16) for x in (3, 24, 30)])
    doy = np.array([1, 90, 180, 200, 300, 365])
    try:
        for nc, kind in ((2, "lon"), (3, "doy")):
            T.configure(ndomain=16, n_channel=nc)
            if nc == 3:
                with pytest.raises(ValueError):
                    T.use_arrays(data, idx)                     # day-of-year list missing
            T.use_arrays(data, idx, timelist=doy)
            np.random.seed(3)
            batch, cond = T._real_batch(7)
            np.random.seed(3)
            ixs = np.random.randint(T.n_samples, size=7)
            b_ref, c_ref = od.gather_real(data, idx, ixs, 16)
            c_ref = od.extra_condition(c_ref, idx[ixs], 16, kind, doy, idx[:, 2].min(), idx[:, 2].max())
            assert cond.shape == (7, 16, 16, nc) and cond.dtype == np.float32
            assert np.array_equal(batch, b_ref) and np.array_equal(cond, c_ref.astype(np.float32))
            latent, c2 = T.generate_latent_points(4)
            assert c2.shape == (4, 16, 16, nc)
            g = W.init_generator(rng, 16, nc)
            assert g[0].shape == (100 + 256 * nc, 3072)          # Dense width 612 / 868
            assert W.init_critic(rng, 16, nc)[0].shape == (3, 3, 3, 1 + nc, 64)
            assert W.infer_config_from_gen(g) == (16, nc)
        assert W.infer_config_from_gen(W.init_generator(rng, 64, 1)[:1]) == (64, 1)
        with pytest.raises(ValueError):
            T.configure(n_channel=4)
    finally:
        T.configure(ndomain=16, n_channel=1)
        T.use_arrays(data, idx)


def test_create_rejects_bad_configuration():
    """argument validation happens before any HIP call, so it is checkable without a GPU"""
    lib = _lib.load()
    h = ctypes.c_void_p()
    for nd, nc, mb in ((16, 0, 4), (16, 4, 4), (12, 1, 4), (16, 1, 0)):
        assert lib.rdgan_create(ctypes.byref(h), nd, nc, mb) == -2 and not h.value
