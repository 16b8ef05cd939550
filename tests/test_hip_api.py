"""-m gpu: the reference-API mirrors end to end on the HIP engine (BASELINE configs[0] plumbing case and a
miniature training run), checked against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import rdgan_np as onp
from oracle import rdgan_torch as ot
from pr_disagg_radar_gan_amd import gan_train_cwgangp_pixelnorm as T
from pr_disagg_radar_gan_amd import models, weights as W
from pr_disagg_radar_gan_amd import raindisagg_gan_pretrained as P

pytestmark = pytest.mark.gpu


def test_generate_scenarios_example_case(tmp_path, monkeypatch):
    """example.py: 10 scenarios for cond = 10 mm/day everywhere, through a Keras-layout .h5 on disk."""
    rng = np.random.default_rng(3)
    g = W.init_generator(rng, 16)
    path = str(tmp_path / "gen_test_0020.h5")
    models.Generator(g, 16).save(path)
    monkeypatch.setattr(P, "generator_file", path)
    monkeypatch.setattr(P, "gen", P._LazyGenerator())
    cond1 = 10 * np.ones((16, 16, 1))
    np.random.seed(0)
    sc = P.generate_scenarios(cond1, 10)
    assert sc.shape == (10, 24, 16, 16) and sc.dtype == np.float64
    np.testing.assert_allclose(sc.sum(axis=1), 10.0, rtol=1e-5)          # every scenario sums to the daily total
    np.random.seed(0)
    ref = onp.generate_scenarios([a.astype(np.float64) for a in g], cond1, 10)
    np.testing.assert_allclose(sc, ref, rtol=1e-4, atol=1e-7)            # north_star tolerance
    assert P.latent_dim == 100 and P.gen.inputs[0].shape[1] == 100


def test_predict_batches_and_critic_predict():
    rng = np.random.default_rng(4)
    gen = models.Generator(W.init_generator(rng, 16), 16)
    crit = models.Critic(W.init_critic(rng, 16), 16)
    x, cond, z = ot.synthetic_batch(7, 16, 11)
    full = gen.predict([z, cond])
    parts = gen.predict([z, cond], batch_size=3)                          # ragged chunks 3+3+1
    np.testing.assert_allclose(full, parts, rtol=2e-5, atol=1e-8)   # tile / split-K choices depend on the batch size: fp32 rounding differs
    v = crit.predict([full, cond])
    ref = ot.critic_forward([torch.from_numpy(a).double() for a in crit.get_weights()], torch.from_numpy(full).double(),
                            torch.from_numpy(cond).double(), None).numpy()
    np.testing.assert_allclose(v, ref, rtol=1e-4, atol=1e-6)


def test_train_mirror_runs_and_matches_oracle(tmp_path, monkeypatch):
    """Two iterations of T.train (n_disc critic steps + 1 generator step each) on synthetic tiles; the weights
    after training equal an oracle replay that is fed the same batches and step seeds."""
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(0)
    data = (rng.gamma(0.3, 2.0, (4, 24, 32, 32)).astype(np.float32) + 1e-3)
    idx = [(t, y, x) for t in range(4) for y in (0, 16) for x in (0, 16)]
    T.configure(ndomain=16, n_disc=2)
    T.use_arrays(data, idx)
    T.build_networks(seed=5)
    g0, d0 = T.generator.get_weights(), T.critic.get_weights()
    T.hist["d_loss"].clear(); T.hist["g_loss"].clear()

    # record what train() feeds the engine
    from pr_disagg_radar_gan_amd import trainer as TR
    calls = []
    orig_c, orig_g = TR.WGANGPTrainer.critic_step, TR.WGANGPTrainer.gen_step

    def rec_c(self, x, c, z, seed=None):
        seed = self._next_seed() if seed is None else seed
        calls.append(("c", x.cpu().numpy(), c.cpu().numpy(), z.cpu().numpy(), seed))
        return orig_c(self, x, c, z, seed)

    def rec_g(self, z, c, seed=None):
        seed = self._next_seed() if seed is None else seed
        calls.append(("g", z.cpu().numpy(), c.cpu().numpy(), seed))
        return orig_g(self, z, c, seed)

    monkeypatch.setattr(TR.WGANGPTrainer, "critic_step", rec_c)
    monkeypatch.setattr(TR.WGANGPTrainer, "gen_step", rec_g)
    np.random.seed(1)
    hist = T.train(1, 4, max_batches_per_epoch=2)
    assert len(hist["d_loss"]) == 2 and np.all(np.isfinite(hist["d_loss"])) and np.all(np.isfinite(hist["g_loss"]))
    assert [c[0] for c in calls] == ["c", "c", "g", "c", "c", "g"]        # T:468-482
    assert os.path.exists("hist.csv")
    saved = [f for f in os.listdir(T.outdir) if f.startswith("gen_") and f.endswith("_0001.h5")]
    assert saved, os.listdir(T.outdir)
    g1 = T.generator.get_weights()
    assert all(np.array_equal(a, b) for a, b in zip(g1, W.load_weights(os.path.join(T.outdir, saved[0]))))

    # oracle replay (fp32 torch, same Adam, shared iteration counter)
    tr = ot.Trainer(16, seed=0)
    tr.gp = [torch.from_numpy(a.copy()) for a in g0]; tr.dp = [torch.from_numpy(a.copy()) for a in d0]
    tr.gv = [torch.zeros_like(p) for p in tr.gp]; tr.dv = [torch.zeros_like(p) for p in tr.dp]
    for c in calls:
        if c[0] == "c":
            tr.critic_step(torch.from_numpy(c[1]), torch.from_numpy(c[2]), torch.from_numpy(c[3]), c[4])
        else:
            tr.gen_step(torch.from_numpy(c[1]), torch.from_numpy(c[2]), c[3])
    assert tr.t == 6
    # Adam's first steps move every weight by ~lr regardless of gradient size, so compare the UPDATE directions
    for name, new, old, ref in (("gen", g1, g0, tr.gp), ("critic", T.critic.get_weights(), d0, tr.dp)):
        for a, o, r in zip(new, old, ref):
            da, dr = a - o, r.numpy() - o
            if np.abs(dr).max() == 0 or a.size == 1:
                continue        # (the last conv's bias has an analytically zero gradient: Adam then amplifies pure rounding noise)
            big = np.abs(dr) > 0.5 * np.abs(dr).max()
            assert np.mean(np.sign(da[big]) == np.sign(dr[big])) > 0.99, name
            assert abs(np.abs(da).max() - np.abs(dr).max()) <= 0.05 * np.abs(dr).max() + 1e-7, name
    T.configure(n_disc=5)


def test_train_mirror_with_device_dataset(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(1)
    data = (rng.gamma(0.3, 2.0, (4, 24, 40, 40)).astype(np.float32) + 1e-3)
    T.configure(ndomain=16, n_disc=1)
    T.use_arrays(data, [(t, y, x) for t in range(4) for y in (0, 16) for x in (0, 20)])
    ds = T.use_device_dataset()
    assert ds.valid_indices(stride=16, tp_thresh_daily=0.5, n_thresh=5) != []
    T.build_networks(seed=9)
    T.hist["d_loss"].clear(); T.hist["g_loss"].clear()
    hist = T.train(1, 4, max_batches_per_epoch=2, save_models=False)
    assert len(hist["g_loss"]) == 2 and np.all(np.isfinite(hist["g_loss"])) and np.all(np.isfinite(hist["d_loss"]))
    T.use_device_dataset(False)
    T.configure(n_disc=5)


@pytest.mark.parametrize("n_channel", [1, 3])
def test_train_mirror_resume_is_bit_identical(tmp_path, monkeypatch, n_channel):
    """T.resume: epoch 1 + epoch 2 in one process state equals epoch 2 continued from the checkpoint
    written after epoch 1 (weights, Adam moments, shared step counter, dropout/alpha seeds, numpy RNG).  Also runs the
    n_channel = 3 variant (sin/cos day of year) end to end through the mirror."""
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(2)
    data = (rng.gamma(0.3, 2.0, (4, 24, 32, 32)).astype(np.float32) + 1e-3)
    idx = [(t, y, x) for t in range(4) for y in (0, 16) for x in (0, 16)]
    try:
        T.configure(ndomain=16, n_disc=1, n_channel=n_channel)
        T.use_arrays(data, idx, timelist=[10, 100, 200, 300])
        T.build_networks(seed=7)
        T.hist["d_loss"].clear(); T.hist["g_loss"].clear()
        np.random.seed(3)
        T.train(1, 4, max_batches_per_epoch=2)
        ck = os.path.join(T.outdir, f"checkpoint_{T.params}.npz")
        assert os.path.exists(ck)
        T.train(1, 4, start_epoch=1, max_batches_per_epoch=2, save_models=False)
        want_g, want_d, want_hist = T.generator.get_weights(), T.critic.get_weights(), list(T.hist["g_loss"])

        T.build_networks(seed=99)                    # fresh, different networks; everything must come from the file
        T.hist["d_loss"].clear(); T.hist["g_loss"].clear()
        np.random.seed(12345)
        T.resume(ck)
        T.train(1, 4, start_epoch=1, max_batches_per_epoch=2, save_models=False)
        assert T.resume_from is None
        assert all(np.array_equal(a, b) for a, b in zip(T.generator.get_weights(), want_g))
        assert all(np.array_equal(a, b) for a, b in zip(T.critic.get_weights(), want_d))
        assert T.hist["g_loss"] == want_hist[2:]
        with pytest.raises(FileNotFoundError):
            T.resume(str(tmp_path / "nope.npz"))
    finally:
        T.configure(n_disc=5, n_channel=1)
        T.use_arrays(data, idx)


def test_train_mirror_in_bf16_storage_mode(tmp_path, monkeypatch):
    """T.train with the engine's "bf16" option (bf16 storage and MFMA operands, fp32 master weights and optimizer state): a few iterations stay finite and
    the losses track the fp32 run of the same batches to bf16 accuracy."""
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(3)
    data = (rng.gamma(0.3, 2.0, (4, 24, 32, 32)).astype(np.float32) + 1e-3)
    idx = [(t, y, x) for t in range(4) for y in (0, 16) for x in (0, 16)]
    T.configure(ndomain=16, n_disc=1)
    T.use_arrays(data, idx)
    runs = {}
    try:
        for mode in (0, 1):
            T.build_networks(seed=21)
            T.hist["d_loss"].clear(); T.hist["g_loss"].clear()
            models.get_engine(16, 8).set_option("bf16", mode)
            np.random.seed(5)
            T.train(1, 8, max_batches_per_epoch=3, save_models=False)
            runs[mode] = (list(T.hist["d_loss"]), list(T.hist["g_loss"]))
            assert np.all(np.isfinite(runs[mode][0])) and np.all(np.isfinite(runs[mode][1]))
        np.testing.assert_allclose(runs[1][0], runs[0][0], rtol=0.1, atol=0.02)
        np.testing.assert_allclose(runs[1][1], runs[0][1], rtol=0.1, atol=0.02)
        assert runs[1] != runs[0]
    finally:
        models.get_engine(16, 8).set_option("bf16", 0)
        T.configure(n_disc=5)


@pytest.mark.parametrize("bf16", [0, 1])
def test_weight_form_cache_is_bit_identical_and_builds_once_per_update(bf16):
    """rdgan_set_weight_versions: the trainer vouches for the content of its slabs (a fresh version after every Adam update),
    so a network's weight forms are built once per update instead of once per call: per iteration of n_disc = 3 critic steps
    + 1 generator step the generator's forward forms once (4 calls use them) and the critic's forms 3 times (the generator
    step and the critic step behind it share one build).  Same kernels on the same data: weights, Adam state and losses are
    bit-identical to a run that rebuilds the forms in every call."""
    from pr_disagg_radar_gan_amd import Engine
    from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer, synthetic_batch_device
    B, n_disc, iters = 6, 3, 4
    eng = Engine(ndomain=16, max_batch=B)
    try:
        if bf16:
            eng.set_option("bf16", 1)
        rng = np.random.default_rng(12)
        g, d = W.init_generator(rng, 16), W.init_critic(rng, 16)
        batches = [synthetic_batch_device(B, 16, 700 + i, eng.device) for i in range(4)]

        def run(cache):
            tr = WGANGPTrainer(eng, g, d, n_disc=n_disc, base_seed=21)
            tr.weight_cache = cache
            if not cache:
                tr.gver = tr.dver = 0
            g0, c0 = eng.form_builds()
            losses = []
            for it in range(iters):
                crit = [batches[(it + j) % 4] for j in range(n_disc)]
                x, c, z = batches[(it + 3) % 4]
                losses.append(torch.stack([t.reshape(()) for t in tr.iteration(crit, (z, c))]))
            torch.cuda.synchronize()
            g1, c1 = eng.form_builds()
            return (tr.gparams.clone(), tr.dparams.clone(), tr.gv.clone(), tr.dv.clone(), torch.stack(losses)), (g1 - g0, c1 - c0)

        a, na = run(True)
        b, nb = run(False)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        assert nb == (iters * (n_disc + 1), iters * (n_disc + 1))            # every call rebuilds
        # once per update of the network (the critic: n_disc updates per iteration, + 1 for the very first call)
        assert na == (iters, iters * n_disc + 1), na
        assert bool(torch.isfinite(a[4]).all())
    finally:
        eng.close()


def test_weight_form_cache_cannot_go_stale_through_the_model_and_trainer_api(tmp_path):
    """Whoever writes a slab takes a fresh version: Generator.set_weights / load_weights drop the device slab (a new slab, a
    new version), a trainer-owned slab adopted by a model is passed with version 0 (rebuild), load_checkpoint bumps both
    versions.  So predictions always follow the weights -- also when torch hands the new slab the address of the old one."""
    from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer
    rng = np.random.default_rng(13)
    ga, gb = W.init_generator(rng, 16), W.init_generator(rng, 16)
    x, cond, z = ot.synthetic_batch(3, 16, 5)
    gen = models.Generator(ga, 16)
    out_a = gen.predict([z, cond])
    np.testing.assert_array_equal(out_a, gen.predict([z, cond]))            # second call: forms reused, same output
    gen.set_weights(gb)                                                      # frees the slab; the next one may reuse its address
    out_b = gen.predict([z, cond])
    np.testing.assert_array_equal(out_b, models.Generator(gb, 16).predict([z, cond]))
    assert np.abs(out_a - out_b).max() > 1e-4
    # trainer: adopt, train, predict through the model (version unknown -> rebuilt), checkpoint round trip
    eng = models.get_engine(16, 3)
    d = W.init_critic(rng, 16)
    tr = WGANGPTrainer(eng, ga, d, n_disc=1)
    gen.adopt_slab(tr.gparams)
    dev = lambda a: torch.from_numpy(a).to(eng.device)
    before = gen.predict([z, cond])
    np.testing.assert_array_equal(before, out_a)
    tr.iteration([(dev(x), dev(cond), dev(z))], (dev(z), dev(cond)))
    after = gen.predict([z, cond])
    assert np.abs(after - before).max() > 0
    path = str(tmp_path / "ck.npz")
    tr.save_checkpoint(path)
    want = [t.clone() for t in tr.iteration([(dev(x), dev(cond), dev(z))], (dev(z), dev(cond)))]
    w_want = tr.gparams.clone()
    tr.load_checkpoint(path)                                                 # slabs rewritten in place: versions must change
    got = tr.iteration([(dev(x), dev(cond), dev(z))], (dev(z), dev(cond)))
    for u, v in zip(want, got):
        assert torch.equal(u, v)
    assert torch.equal(w_want, tr.gparams)
