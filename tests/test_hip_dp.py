"""-m gpu: the data-parallel path on the REAL HIP engine -- two ranks with different shards against the global batch,
the self-launching bench.py, and the stream ordering of the overlapped exchange -- plus the non-finite guard."""
import json
import os
import subprocess
import sys

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


def _spawn_ranks(world, out_dir, data_seed, backend="gloo", overlap=-1):
    """world fresh processes (children of this one; none of them replaces a GPU-initialised program), all on cuda:0"""
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, "-m", "tests.dp_case", "--rank", str(r), "--world", str(world), "--port", str(port),
                               "--out", str(out_dir), "--backend", backend, "--data-seed", str(data_seed), "--overlap", str(overlap)],
                              cwd=ROOT, env=env) for r in range(world)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0] * world, rcs
    return [torch.load(os.path.join(out_dir, f"rank{r}.pt")) for r in range(world)]


def test_two_ranks_different_shards_equal_the_global_batch_on_the_hip_engine(tmp_path):
    """SURVEY 8e on the real engine: two processes, each with its own HIP engine and a DIFFERENT shard of 2 samples, the
    gradient slabs summed through torch.distributed (gloo here: one GPU; the production backend is RCCL), against one process
    with the global batch of 4.  Both sides are fp32 with different tile / split choices, so a LeakyReLU input within
    rounding of zero can take the other slope in one of them (tests/test_hip_step.py::_parity_over_batches): every data seed
    must agree loosely, one of at most three tightly.  The overlapped exchange (side stream) is what runs in the ranks."""
    eng = Engine(ndomain=dp_case.NDOMAIN, max_batch=dp_case.N_GLOBAL)
    try:
        history = []
        for data_seed in (21, 22, 23):
            out = tmp_path / f"s{data_seed}"
            out.mkdir()
            r0, r1 = _spawn_ranks(2, out, data_seed)
            assert r0["overlap"] and r1["overlap"]
            for k in ("dgrad", "ggrad", "dl", "gl", "dparams", "gparams"):
                assert torch.equal(r0[k], r1[k]), k                    # replicas bit-identical
            one = dp_case.run_iteration(eng, 1, 0, None, data_seed)
            nd_, ng_ = eng.n_critic, eng.n_gen
            e = dp_case.grad_errors(r0["dgrad"][:nd_], one["dgrad"][:nd_], eng.critic_shapes)
            e.update({"g/" + k: v for k, v in dp_case.grad_errors(r0["ggrad"][:ng_], one["ggrad"][:ng_], eng.gen_shapes).items()})
            worst = max(e.values())
            history.append((data_seed, float(f"{worst:.2e}")))
            assert worst < 5e-2, (data_seed, e)
            assert torch.allclose(r0["dl"][:4], one["dl"][:4], rtol=1e-4, atol=1e-6)
            assert torch.allclose(r0["gl"][:1], one["gl"][:1], rtol=1e-4, atol=1e-6)
            assert float(r0["dl"][4]) == 0 and float(r0["gl"][4]) == 0
            if worst < 2e-5:
                print("DP vs global batch, per-tensor gradient errors:", {k: float(f"{v:.1e}") for k, v in e.items()})
                return
        raise AssertionError(f"no data seed reached the tight tolerance: {history}")
    finally:
        eng.close()


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
