"""A/B of two builds of librdgan_hip.so in ONE process on one device (guide rule 24): per-launch table of a few training
iterations with each library, interleaved rounds.   python scratch/ab_libs.py libA.so libB.so [--bf16 1] [--batch 256]"""
import argparse, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import _lib, weights as W
from pr_disagg_radar_gan_amd.engine import Engine
from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer, synthetic_batch_device

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs=2)
ap.add_argument("--bf16", type=int, default=0)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--n-critic", type=int, default=1)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
engs = []
for p in a.libs:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(p)
    e = Engine(16, a.batch)
    if a.bf16:
        e.set_option("bf16", 1)
    for kv in a.opt:
        k, v = kv.split("=")
        try:
            e.set_option(k, int(v))
        except Exception as ex:
            print("option", k, "not set on", p, ex)
    engs.append(e)
rng = np.random.default_rng(0)
g, d = W.init_generator(rng, 16), W.init_critic(rng, 16)
trs = [WGANGPTrainer(e, g, d, n_disc=a.n_critic) for e in engs]
x, c, z = synthetic_batch_device(a.batch, 16, 1, engs[0].device)
def iters(tr, n):
    for _ in range(n):
        tr.iteration_raw([(x, c, z)] * a.n_critic, (z, c))
for tr in trs:
    iters(tr, 3)
torch.cuda.synchronize()
acc = [dict(), dict()]
wall = [[], []]
for r in range(a.rounds):
    for i, (e, tr) in enumerate(zip(engs, trs)):
        e.profile_launches(True)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); iters(tr, 3); t1.record(); torch.cuda.synchronize()
        wall[i].append(t0.elapsed_time(t1) / 3)
        for row in e.launch_table():
            k = (row["name"], row["kind"], row["kernel"], row["batch"])
            s = acc[i].setdefault(k, [0.0, 0])
            s[0] += row["ms"]; s[1] += row["launches"]
        e.profile_launches(False)
print("iteration ms (with per-launch events): A", [round(v, 3) for v in wall[0]], "B", [round(v, 3) for v in wall[1]])
keys = sorted(set(acc[0]) | set(acc[1]), key=lambda k: -(acc[0].get(k, [0, 1])[0]))
print(f"{'what':40s} {'kind':6s} {'kernel A':34s} {'ms A':>8s} {'ms B':>8s} {'B/A':>6s}")
for k in keys:
    ma = acc[0].get(k); mb = acc[1].get(k)
    if ma is None or mb is None:
        # kernels named differently in the two builds: match on (name, kind, batch)
        alt = [kk for kk in (acc[1] if mb is None else acc[0]) if kk[0] == k[0] and kk[1] == k[1] and kk[3] == k[3]]
        if mb is None and alt: mb = acc[1][alt[0]]
        if ma is None: continue
    if ma is None or mb is None: continue
    A = ma[0] / ma[1]; Bm = mb[0] / mb[1]
    if A * ma[1] / (a.rounds * 3) < 0.03: continue
    print(f"{k[0][:40]:40s} {k[1]:6s} {k[2][:34]:34s} {A:8.4f} {Bm:8.4f} {Bm / A:6.3f}")
