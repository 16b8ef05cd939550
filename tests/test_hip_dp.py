"""-m gpu: the data-parallel path on the REAL HIP engine -- two ranks with different shards against the global batch,
the self-launching bench.py, and the stream ordering of the overlapped exchange -- plus the non-finite guard."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from pr_disagg_radar_gan_amd import Engine, _lib, models
from pr_disagg_radar_gan_amd import weights as W
from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer
from tests import dp_case
from tests.hip_util import dev

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_ranks(world, out_dir, data_seed, backend="gloo", overlap=-1, gates=0, exchange=None):
    """world fresh processes (children of this one; none of them replaces a GPU-initialised program), all on cuda:0"""
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT)
    extra = ["--exchange", exchange] if exchange else []
    procs = [subprocess.Popen([sys.executable, "-m", "tests.dp_case", "--rank", str(r), "--world", str(world), "--port", str(port),
                               "--out", str(out_dir), "--backend", backend, "--data-seed", str(data_seed), "--overlap", str(overlap),
                               "--gates", str(gates)] + extra,
                              cwd=ROOT, env=env) for r in range(world)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0] * world, rcs
    return [torch.load(os.path.join(out_dir, f"rank{r}.pt")) for r in range(world)]


def _oracle_of_the_global_batch(ranks, data_seed):
    """fp64 oracle of the GLOBAL batch on the LeakyReLU branch the ranks took: their slope patterns, re-assembled into the
    global batch order ([real; fake; interpolated] of all samples for the critic step), checked against the oracle's own
    decisions away from the kinks.  Returns (critic losses, critic grads, generator loss, generator grads)."""
    from oracle import rdgan_torch as ot
    from tests.hip_util import GATE_TOL
    g, d = dp_case.initial_weights()
    (x, c, z), (c2, z2) = dp_case.global_batches(data_seed)
    t64 = lambda arrs: [torch.from_numpy(np.asarray(a)).double() for a in arrs]
    per = dp_case.N_GLOBAL // len(ranks)
    cg = [torch.cat([r["cgates"][li][k * per:(k + 1) * per] for k in range(3) for r in ranks]) for li in range(4)]
    closs, cgrads, ch = ot.critic_step_grads(t64(d), t64(g), *t64((x, c, z)), 0, gates=cg, return_intermediates=True)
    ot.check_gates(cg, ch, None, **GATE_TOL["f32"])
    # the generator step runs against the critic weights the critic step's Adam update left (replicas identical)
    d1 = W.unflatten(ranks[0]["dparams"].numpy(), W.critic_param_shapes(dp_case.NDOMAIN))
    gg = ([torch.cat([r["ggates"][0][i] for r in ranks]) for i in range(4)],
          [torch.cat([r["ggates"][1][i] for r in ranks]) for i in range(4)])
    gloss, ggrads, (gh, dh) = ot.gen_step_grads(t64(d1), t64(g), *t64((z2, c2)), 0, gates=gg, return_intermediates=True)
    ot.check_gates(gg[0], gh, None, **GATE_TOL["f32"])
    ot.check_gates(gg[1], dh, None, **GATE_TOL["f32"])
    return closs, cgrads, gloss, ggrads


@pytest.mark.parametrize("exchange", ["allreduce", "sharded"])
def test_two_ranks_different_shards_equal_the_global_batch_on_the_hip_engine(tmp_path, exchange):
    """SURVEY 8e on the real engine: two processes, each with its own HIP engine and a DIFFERENT shard of 2 samples, the
    gradient slabs exchanged through torch.distributed (gloo here: one GPU; the production backend is RCCL), against the
    fp64 oracle of the global batch of 4 on the LeakyReLU branch the ranks took (one seeded batch, tight: see
    tests/test_hip_step.py).  The overlapped exchange (communication stream) is what runs in the ranks.  exchange = "sharded":
    reduce-scatter, Adam on the owned half, all-gather of the updated weights -- the exchange of the 837 MB generator slab of
    ndomain 64, forced here on the ndomain-16 slabs."""
    from tests.test_hip_step import TIGHT
    data_seed = 21
    r0, r1 = _spawn_ranks(2, tmp_path, data_seed, gates=1, exchange=exchange)
    assert r0["overlap"] and r1["overlap"]
    assert r0["exchange"] == {"g": exchange, "d": exchange}
    for k in ("dgrad", "ggrad", "dl", "gl", "dparams", "gparams", "dv", "gv"):
        assert torch.equal(r0[k], r1[k]), k                    # replicas bit-identical
    closs, cgrads, gloss, ggrads = _oracle_of_the_global_batch([r0, r1], data_seed)
    nd_ = W.param_count(W.critic_param_shapes(dp_case.NDOMAIN))
    ng_ = W.param_count(W.gen_param_shapes(dp_case.NDOMAIN))
    e = dp_case.grad_errors(r0["dgrad"][:nd_], torch.cat([t.reshape(-1) for t in cgrads]), W.critic_param_shapes(dp_case.NDOMAIN),
                            skip=("dense_1/bias:0",))
    e.update({"g/" + k: v for k, v in dp_case.grad_errors(r0["ggrad"][:ng_], torch.cat([t.reshape(-1) for t in ggrads]),
                                                          W.gen_param_shapes(dp_case.NDOMAIN)).items()})
    print(f"DP ({exchange}) vs the fp64 oracle of the global batch, per-tensor gradient errors:", {k: float(f"{v:.1e}") for k, v in e.items()})
    assert max(e.values()) < TIGHT, e
    assert torch.allclose(r0["dl"][:4].double(), closs, rtol=2e-4, atol=1e-6)
    assert abs(float(r0["gl"][0]) - float(gloss)) <= 2e-4 * abs(float(gloss)) + 1e-6
    assert float(r0["dl"][4]) == 0 and float(r0["gl"][4]) == 0
    # Adam's first step on the oracle's gradients (t = 1 critic, t = 2 generator, shared counter): the updated weights
    g, d = dp_case.initial_weights()
    for params, grads, t, got in ((d, cgrads, 1, r0["dparams"]), (g, ggrads, 2, r0["gparams"])):
        ps = [torch.from_numpy(a).double() for a in params]
        vs = [torch.zeros_like(p) for p in ps]
        ot_adam(ps, list(grads), vs, t)
        want = torch.cat([p.reshape(-1) for p in ps])
        # lr * g / (|g| + eps'): a step of at most 1e-4 * sqrt(1 - 0.9^t)/sqrt(0.1); equal wherever g is not within rounding of 0
        assert float((got.double() - want).abs().mean()) < 1e-8


def ot_adam(ps, grads, vs, t):
    from oracle import rdgan_torch as ot
    ot.adam_update(ps, grads, vs, t)


def test_overlapped_exchange_is_ordered_by_events():
    """The trainer's overlapped mode (exchange + Adam on a communication stream, generator forward of the next gradient
    call in front of the wait for the critic weights) against the single-stream mode, world 1, with a 20 ms delay kernel
    in place of the all-reduce: any kernel that read weights, Adam state or a gradient slab without its event would see
    stale or half-updated data.  n_disc = 2 so that critic -> critic, critic -> generator and generator -> critic hand-offs
    all occur.  Results must be bit-identical."""
    from oracle import rdgan_torch as ot
    eng = Engine(ndomain=16, max_batch=4)
    try:
        rng = np.random.default_rng(7)
        g, d = W.init_generator(rng, 16), W.init_critic(rng, 16)
        batches = []
        for i in range(3):
            x, c, z = ot.synthetic_batch(4, 16, 300 + i)
            batches.append((dev(x), dev(c), dev(z)))

        def run(overlap):
            hook = (lambda slab: torch.cuda._sleep(40_000_000)) if overlap else None
            tr = WGANGPTrainer(eng, g, d, n_disc=2, overlap=overlap, comm_hook=hook, base_seed=5)
            assert tr.overlap == overlap
            res = []
            for it in range(3):
                x, c, z = batches[it]
                x2, c2, z2 = batches[(it + 1) % 3]
                res.append(tr.iteration([(x, c, z), (x2, c2, z2)], (z, c)))
            torch.cuda.synchronize()
            return tr, res

        a, ra = run(False)
        b, rb = run(True)
        for name in ("gparams", "dparams", "gv", "dv"):
            assert torch.equal(getattr(a, name), getattr(b, name)), name
        for u, v in zip(ra, rb):
            for s, t in zip(u, v):
                assert torch.equal(s, t)
        assert a.t == b.t == 9
    finally:
        eng.close()


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: the script starts two fresh rank processes itself (here both on
    cuda:0 with gloo: a one-GPU rehearsal of the RCCL path) and prints ONE JSON line from rank 0."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--backend", "gloo", "--steps", "3",
           "--warmup", "1", "--batch", "8", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, timeout=900)
    assert p.returncode == 0
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world"] == 2 and out["config"]["rccl_ranks_seen"] == 2
    assert out["config"]["global_batch"] == 16 and out["value"] > 0
    assert "side stream" in out["config"]["exchange"]
    assert out["iteration_ms"]["n"] == 3 and out["roofline"]["iteration"]["executed_gflop"] > 0


@pytest.mark.parametrize("config,batch,ex", [(5, 2, "sharded"), (4, 8, "allreduce")])
def test_bench_strong_scaling_configs_rehearsal(config, batch, ex):
    """`bench.py --gpus 2 --config 5|4` (BASELINE configs[4] / configs[3]: bf16 storage, n_critic 5) rehearsed on one GPU over
    gloo with a small per-rank batch: at ndomain 64 the 837 MB generator slab takes the sharded exchange by default
    (reduce-scatter, Adam on the owned half, all-gather), the 11.6 MB critic slab the all-reduce."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--backend", "gloo", "--steps", "1",
           "--warmup", "1", "--config", str(config), "--batch", str(batch), "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, timeout=900)
    assert p.returncode == 0
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world"] == 2 and out["config"]["rccl_ranks_seen"] == 2
    assert out["config"]["n_critic"] == 5 and out["config"]["global_batch"] == 2 * batch and out["value"] > 0
    assert out["config"]["exchange_by_slab"] == {"g": ex, "d": "allreduce"}
    assert "bf16" in out["dtype"]
    assert np.isfinite(out["final_losses"]["d_loss"]) and np.isfinite(out["final_losses"]["g_loss"])


def test_bench_exits_nonzero_when_a_rank_dies(tmp_path):
    """launch_children polls its ranks: when one exits non-zero at start-up the others are terminated and bench.py returns
    that code at once instead of leaving rank 0 inside a collective until a watchdog fires."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--backend", "gloo", "--steps", "1",
           "--warmup", "0", "--batch", "4", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["RDGAN_BENCH_FAIL_RANK"] = "1"             # test hook: that rank exits with code 3 before the rendezvous
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0
    assert time.time() - t0 < 300
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_check_numerics_guard():
    """tf.debugging.check_numerics behind the generator's softmax (T:349-350) and the NaN-loss guard (T:487-488): a NaN
    generator weight makes `predict` raise, and both gradient entries report it in slot 4 of their loss tail."""
    from oracle import rdgan_torch as ot
    rng = np.random.default_rng(9)
    g, d = W.init_generator(rng, 16), W.init_critic(rng, 16)
    x, cond, z = ot.synthetic_batch(3, 16, 5)
    gen = models.Generator(g, 16)
    out = gen.predict([z, cond])                                   # clean weights: no error
    assert np.all(np.isfinite(out))
    bad = [a.copy() for a in g]
    bad[6][1, 1, 1, 5, 7] = np.nan                                 # one kernel entry of the third block
    with pytest.raises(_lib.NumericsError, match="per_gridpoint_softmax"):
        models.Generator(bad, 16).predict([z, cond])
    eng = Engine(ndomain=16, max_batch=3)
    try:
        gs, ds, gbad = eng.to_slab(g), eng.to_slab(d), eng.to_slab(bad)
        ok = eng.critic_grad(ds, gs, dev(x), dev(cond), dev(z), 3).cpu().numpy()
        assert ok[eng.n_critic + 4] == 0
        eng.check_numerics()
        s = eng.critic_grad(ds, gbad, dev(x), dev(cond), dev(z), 3).cpu().numpy()
        assert s[eng.n_critic + 4] == 1
        with pytest.raises(_lib.NumericsError):
            eng.check_numerics()
        s = eng.gen_grad(ds, gbad, dev(z), dev(cond), 3).cpu().numpy()
        assert s[eng.n_gen + 4] == 1
        dbad = [a.copy() for a in d]
        dbad[2][0, 0, 0, 0, 0] = np.inf                             # a critic weight: the loss itself goes non-finite
        s = eng.gen_grad(eng.to_slab(dbad), gs, dev(z), dev(cond), 3).cpu().numpy()
        assert s[eng.n_gen + 4] == 1
        s = eng.gen_grad(ds, gs, dev(z), dev(cond), 3).cpu().numpy()  # and the flag clears again
        assert s[eng.n_gen + 4] == 0
    finally:
        eng.close()
