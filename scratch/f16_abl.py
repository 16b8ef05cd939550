"""Where k_conv_gemm_f16 spends its time: python3 scratch/f16_abl.py [nd] [B]  builds diagnostic variants of the library
(-DRD_F16_ABL_NODMA / NOMFMA / NOEPI and combinations; results are garbage) next to the shipped one and prints, per variant, the
per-launch times of the GEMM launches of one critic step and one generator step (each library in a fresh child process)."""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "f16_abl")
VARIANTS = {"full": [], "noepi": ["-DRD_F16_ABL_NOEPI"], "nodma": ["-DRD_F16_ABL_NODMA"], "nomfma": ["-DRD_F16_ABL_NOMFMA"],
            "nodma_noepi": ["-DRD_F16_ABL_NODMA", "-DRD_F16_ABL_NOEPI"], "nomfma_noepi": ["-DRD_F16_ABL_NOMFMA", "-DRD_F16_ABL_NOEPI"]}
if len(sys.argv) < 2 or not sys.argv[1].endswith(".so"):
    nd = sys.argv[1] if len(sys.argv) > 1 else "16"
    B = sys.argv[2] if len(sys.argv) > 2 else "2048"
    os.makedirs(OUT, exist_ok=True)
    procs = {}
    for name, flags in VARIANTS.items():
        lib = os.path.join(OUT, f"lib_{name}.so")
        procs[name] = subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
                                        "-Wno-pass-failed"] + flags + [os.path.join(ROOT, "pr_disagg_radar_gan_amd", "csrc", "rdgan_api.hip"), "-o", lib],
                                       stderr=subprocess.DEVNULL)
    for name, p in procs.items():
        if p.wait() != 0:
            print("build failed:", name); sys.exit(1)
    print("built", list(VARIANTS), flush=True)
    for name in VARIANTS:
        subprocess.run([sys.executable, os.path.abspath(__file__), os.path.join(OUT, f"lib_{name}.so"), nd, B], check=False)
    sys.exit(0)
lib, nd, B = os.path.abspath(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sys.path.insert(0, ROOT)
from pr_disagg_radar_gan_amd import _lib
_lib.LIB_PATH = lib
import numpy as np
import torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import synthetic_batch_device
eng = Engine(nd, B)
eng.set_option("bf16", 1)
rng = np.random.default_rng(0)
gs, ds = eng.to_slab(W.init_generator(rng, nd)), eng.to_slab(W.init_critic(rng, nd))
x, c, z = synthetic_batch_device(B, nd, 1, eng.device)
for i in range(2):
    eng.critic_grad(ds, gs, x, c, z, 5); eng.gen_grad(ds, gs, z, c, 7)
torch.cuda.synchronize()
eng.profile_launches(True)
for i in range(3):
    eng.critic_grad(ds, gs, x, c, z, 5); eng.gen_grad(ds, gs, z, c, 7)
rows = [r for r in eng.launch_table() if "f16" in r["kernel"]]
print(os.path.basename(lib))
for r in rows:
    print("   %-36s %-30s B %5d  %3d launches  %.4f ms" % (r["name"], r["kernel"], r["batch"], r["launches"], r["ms"] / r["launches"]), flush=True)
