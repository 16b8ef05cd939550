"""A/B of builds of librdgan_hip.so in ONE process: un-profiled iteration time, interleaved rounds.
   python scratch/ab_iter.py libA.so libB.so ... [--bf16 1] [--batch 2048] [--n-critic 5] [--opt k=v]"""
import argparse, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import _lib, weights as W
from pr_disagg_radar_gan_amd.engine import Engine
from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer, synthetic_batch_device
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--bf16", type=int, default=1)
ap.add_argument("--batch", type=int, default=2048)
ap.add_argument("--n-critic", type=int, default=5)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
rng = np.random.default_rng(0)
g, d = W.init_generator(rng, 16), W.init_critic(rng, 16)
trs = []
for spec in a.libs:                       # lib.so[:k=v,k=v]
    p, _, o = spec.partition(":")
    _lib._lib = None; _lib.LIB_PATH = os.path.abspath(p)
    import ctypes
    probe = ctypes.CDLL(_lib.LIB_PATH)
    sigs = dict(_lib.SIGNATURES)
    _lib.SIGNATURES = {k: v for k, v in sigs.items() if hasattr(probe, k)}      # (an older build lacks the newest op entries)
    e = Engine(16, a.batch)
    _lib.SIGNATURES = sigs
    if a.bf16: e.set_option("bf16", 1)
    for kv in a.opt + ([x for x in o.split(",") if x]):
        k, v = kv.split("="); e.set_option(k, int(v))
    trs.append((spec, WGANGPTrainer(e, g, d, n_disc=a.n_critic)))
x, c, z = synthetic_batch_device(a.batch, 16, 1, trs[0][1].eng.device)
def iters(tr, n):
    for _ in range(n):
        tr.iteration_raw([(x, c, z)] * a.n_critic, (z, c))
for _, tr in trs: iters(tr, 2)
torch.cuda.synchronize()
ms = {s: [] for s, _ in trs}
for r in range(a.rounds):
    for s, tr in trs:
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); iters(tr, a.iters); t1.record(); torch.cuda.synchronize()
        ms[s].append(t0.elapsed_time(t1) / a.iters)
for s, v in ms.items():
    print(f"{s}: median {np.median(v):.3f} ms  min {min(v):.3f}  ({a.batch / np.median(v):.2f} k samples/s)", flush=True)
