"""One data-parallel training iteration over DIFFERENT shards, shared by the CPU (gloo + oracle-backed FakeEngine) and
the GPU (gloo or RCCL + the real HIP engine) equivalence tests: the mean of the shard gradients must equal the gradient
of the global batch.  Test infrastructure.

Shard invariance needs every per-sample random draw to be independent of the sample's position in the local batch:
dropout is switched off (seed 0, the ``predict`` convention) and RandomWeightedAverage's alpha is keyed by the GLOBAL
sample index through the engine option "sample_offset".

As a script (one rank of a world, real HIP engine; every rank may share cuda:0 -- a rehearsal on a one-GPU box):

    python -m tests.dp_case --rank R --world W --port P --out DIR [--backend gloo|nccl] [--data-seed S] [--overlap 0|1]
                            [--gates 1] [--exchange allreduce|sharded]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_GLOBAL = 4
NDOMAIN = 16


def initial_weights():
    """identical initial weights on every rank (non-zero biases so that every parameter tensor has a gradient of ordinary size)"""
    from pr_disagg_radar_gan_amd import weights as W
    rng = np.random.default_rng(0)
    g, d = W.init_generator(rng, NDOMAIN), W.init_critic(rng, NDOMAIN)
    g = [p if p.ndim > 1 else (0.05 * rng.standard_normal(p.shape)).astype(np.float32) for p in g]
    d = [p if p.ndim > 1 else (0.05 * rng.standard_normal(p.shape)).astype(np.float32) for p in d]
    return g, d


def global_batches(data_seed):
    """(x, c, z) of the critic step and (c2, z2) of the generator step, for the GLOBAL batch"""
    from oracle import rdgan_torch as ot
    x, c, z = ot.synthetic_batch(N_GLOBAL, NDOMAIN, data_seed)
    _, c2, z2 = ot.synthetic_batch(N_GLOBAL, NDOMAIN, data_seed + 1000)
    return (x, c, z), (c2, z2)


def run_iteration(engine, world, rank, pg, data_seed=21, overlap=None, comm_hook=None, collect_gates=False, exchange=None,
                  grad_transport=None):
    """critic step + generator step on this rank's shard of a global batch of N_GLOBAL; returns the all-reduced, 1/world
    scaled gradient slabs (what Adam consumed), the reported losses and the updated weights, all on the CPU.
    collect_gates (real HIP engine only): also the LeakyReLU slope patterns of both steps on this shard ("cgates": critic
    layers over [real; fake; interpolated], "ggates": (generator h0..h3, critic layers)), for the fp64 oracle to differentiate
    the branch this run took.  exchange: WGANGPTrainer's gradient exchange ("allreduce" / "sharded"; None = its default)."""
    from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer, shard_slice
    g, d = initial_weights()
    (x, c, z), (c2, z2) = global_batches(data_seed)
    sl = shard_slice(N_GLOBAL, world, rank)
    per = sl.stop - sl.start
    engine.set_option("sample_offset", sl.start)
    kw = {} if exchange is None else {"exchange": exchange}
    if grad_transport is not None:
        kw["grad_transport"] = grad_transport
    tr = WGANGPTrainer(engine, g, d, n_disc=1, process_group=pg, world_size=world, rank=rank, overlap=overlap,
                       comm_hook=comm_hook, **kw)
    dev = tr.gparams.device
    put = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl])).to(dev)
    res = {}
    if collect_gates:
        from tests.hip_util import hip_gates, hip_critic_gates
        engine.set_option("keep_gates", 1)
    dl = tr.critic_step(put(x), put(c), put(z), seed=0)
    tr.join()
    if collect_gates:
        res["cgates"] = hip_critic_gates(engine, per)
        engine.set_option("keep_gates", 0)
    dgrad = tr.reduced_grad("d").cpu().clone()
    gl = tr.gen_step(put(z2), put(c2), seed=0)
    tr.join()
    if collect_gates:
        res["ggates"] = hip_gates(engine, per)
    ggrad = tr.reduced_grad("g").cpu().clone()
    assert tr.t == 2                                       # one Adam counter shared by both models (reference :385,391,408)
    tr.sync_state()                                        # (sharded exchange: gather the Adam second moments from their owners)
    engine.set_option("sample_offset", 0)
    res.update({"dgrad": dgrad, "ggrad": ggrad, "dl": dl.cpu().clone(), "gl": gl.cpu().clone(),
                "dparams": tr.dparams.cpu().clone(), "gparams": tr.gparams.cpu().clone(),
                "dv": tr.dv.cpu().clone(), "gv": tr.gv.cpu().clone(), "overlap": tr.overlap, "exchange": tr.exchange})
    return res


def grad_errors(got, ref, shapes, skip=()):
    """per-tensor max-abs error relative to the tensor's max-abs gradient"""
    off, errs = 0, {}
    for name, s in shapes:
        n = int(np.prod(s))
        a, b = got[off:off + n].double(), ref[off:off + n].double()
        off += n
        if name == "conv3d_3/bias:0" or name in skip:      # analytically zero (softmax shift invariance; critic step: the last
            assert float(a.abs().max()) < 1e-6             # bias sees mean(-1) + mean(+1)): rounding noise only
            continue
        errs[name] = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    return errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--data-seed", type=int, default=21)
    ap.add_argument("--overlap", type=int, default=-1)
    ap.add_argument("--gates", type=int, default=0)
    ap.add_argument("--exchange", default=None)
    args = ap.parse_args()
    import torch.distributed as dist
    from pr_disagg_radar_gan_amd import Engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(args.port)
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(args.rank % max(ndev, 1))
    dist.init_process_group(args.backend, rank=args.rank, world_size=args.world)
    eng = Engine(ndomain=NDOMAIN, max_batch=N_GLOBAL)
    res = run_iteration(eng, args.world, args.rank, dist.group.WORLD, args.data_seed,
                        overlap=None if args.overlap < 0 else bool(args.overlap), collect_gates=bool(args.gates),
                        exchange=args.exchange)
    torch.save(res, os.path.join(args.out, f"rank{args.rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
