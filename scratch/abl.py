import sys, os, shutil, subprocess, json
# run gen_forward timing with a given library file (copied over the in-tree .so inside a temp copy is overkill: use env)
lib = sys.argv[1]
sys.path.insert(0, "/root/repo")
from pr_disagg_radar_gan_amd import _lib
_lib.LIB_PATH = lib
import numpy as np, torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import synthetic_batch_device
eng = Engine(16, 256)
rng = np.random.default_rng(0)
gs = eng.to_slab(W.init_generator(rng, 16))
x, c, z = synthetic_batch_device(256, 16, 1, eng.device)
for i in range(3): eng.gen_forward(gs, z, c)
torch.cuda.synchronize()
eng.profile((1 << 5) | (1 << 0))
for i in range(10): eng.gen_forward(gs, z, c)
ms5, n5 = eng.profile_read(5); ms0, n0 = eng.profile_read(0)
print(os.path.basename(lib), "G3fwd avg ms %.3f" % (ms5 / n5), "G1+G2 fwd avg ms %.3f" % (ms0 / n0))
