"""One-process A/B of the fragment GEMM against the streaming GEMM, per launch: python3 scratch/f16_ab.py [nd] [B] [rounds]
(critic step + generator step of the bf16 mode with "conv_f16" = 0 / 2, interleaved rounds, best round per launch)."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import synthetic_batch_device
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
eng = Engine(nd, B)
eng.set_option("bf16", 1)
rng = np.random.default_rng(0)
gs, ds = eng.to_slab(W.init_generator(rng, nd)), eng.to_slab(W.init_critic(rng, nd))
x, c, z = synthetic_batch_device(B, nd, 1, eng.device)
best = {}
for r in range(rounds + 1):
    for v in (0, 2):
        eng.set_option("conv_f16", v)
        eng.critic_grad(ds, gs, x, c, z, 5); eng.gen_grad(ds, gs, z, c, 7)
        torch.cuda.synchronize()
        eng.profile_launches(True)
        for i in range(3):
            eng.critic_grad(ds, gs, x, c, z, 5); eng.gen_grad(ds, gs, z, c, 7)
        rows = eng.launch_table()
        eng.profile_launches(False)
        if r == 0:
            continue
        seen = {}
        for row in rows:
            k = (row["name"], row["batch"], row["kind"], seen.setdefault((row["name"], row["batch"], row["kind"]), 0))
            seen[(row["name"], row["batch"], row["kind"])] += 1
            t = row["ms"] / row["launches"]
            e = best.setdefault(k[:3], {})
            key = (v, "f16" in row["kernel"] or v == 0 and "conv_gemm_ws" in row["kernel"])
            if row["kind"] == "gemm" and ("conv_gemm" in row["kernel"]):
                e[v] = min(e.get(v, 1e9), t)
                e["n"] = row["launches"] // 3
                e["k%d" % v] = row["kernel"]
tot0 = tot2 = 0.0
for k, e in best.items():
    if 0 in e and 2 in e and "f16" in e.get("k2", ""):
        print("%-40s B %5d  x%-2d  ws %.4f  f16 %.4f  (%+5.1f %%)" % (k[0][:40], k[1], e["n"], e[0], e[2], 100 * (e[2] / e[0] - 1)))
        tot0 += e[0] * e["n"]; tot2 += e[2] * e["n"]
print("sum over these launches per step pair: ws %.3f ms  f16 %.3f ms" % (tot0, tot2))
