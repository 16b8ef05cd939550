"""A/B of library builds on the bf16 storage mode's dominant launch (generator block 3 forward, collapsed form, bs 256):
python3 scratch/abl16.py LIB.so [LIB2.so ...]   -- each library in a fresh child process (diagnostic builds: -DRD_ABL_NODMA,
-DRD_ABL_NOMFMA, -DRD_ABL_NOEPI take the loads / the MFMAs / the epilogue out of k_conv_gemm_ws<..., BF>; results are garbage)."""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2:
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), lib], check=False)
    sys.exit(0)
lib = os.path.abspath(sys.argv[1])
sys.path.insert(0, ROOT)
from pr_disagg_radar_gan_amd import _lib
_lib.LIB_PATH = lib
import numpy as np
import torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import synthetic_batch_device
B = int(os.environ.get("ABL_BATCH", "256"))
eng = Engine(16, B)
eng.set_option("bf16", 1)
rng = np.random.default_rng(0)
gs = eng.to_slab(W.init_generator(rng, 16))
x, c, z = synthetic_batch_device(B, 16, 1, eng.device)
for i in range(3):
    eng.gen_forward(gs, z, c)
torch.cuda.synchronize()
eng.profile((1 << 5) | (1 << 0))
for i in range(20):
    eng.gen_forward(gs, z, c)
ms5, n5 = eng.profile_read(5)
ms0, n0 = eng.profile_read(0)
print(os.path.basename(lib), "block-3 forward avg ms %.4f" % (ms5 / n5), " blocks 1+2 forward avg ms %.4f" % (ms0 / n0), flush=True)
