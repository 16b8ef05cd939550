"""Workload for the profiler passes: a few training iterations (n_critic critic updates + 1 generator update) on
synthetic tiles.  python3 scripts/wl_iteration.py [--bf16 1] [--batch 256] [--n-critic 1] [--iters 2] [--ndomain 16] [--opt name=value]"""
import argparse
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer, synthetic_batch_device

ap = argparse.ArgumentParser()
ap.add_argument("--bf16", type=int, default=0)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--n-critic", type=int, default=1)
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--ndomain", type=int, default=16)
ap.add_argument("--opt", action="append", default=[], help="engine option name=value")
a = ap.parse_args()
eng = Engine(a.ndomain, a.batch)
if a.bf16:
    eng.set_option("bf16", 1)
for kv in a.opt:
    k, v = kv.split("=")
    eng.set_option(k, int(v))
rng = np.random.default_rng(0)
tr = WGANGPTrainer(eng, W.init_generator(rng, a.ndomain), W.init_critic(rng, a.ndomain), n_disc=a.n_critic)
x, c, z = synthetic_batch_device(a.batch, a.ndomain, 1, eng.device)
for i in range(a.iters):
    tr.iteration([(x, c, z)] * a.n_critic, (z, c))
torch.cuda.synchronize()
print("done")
