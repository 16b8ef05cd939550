import sys, os
lib = sys.argv[1]
sys.path.insert(0, "/root/repo")
from pr_disagg_radar_gan_amd import _lib
_lib.LIB_PATH = lib
import numpy as np, torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import synthetic_batch_device
eng = Engine(16, 256)
rng = np.random.default_rng(0)
gs = eng.to_slab(W.init_generator(rng, 16)); ds = eng.to_slab(W.init_critic(rng, 16))
x, c, z = synthetic_batch_device(256, 16, 1, eng.device)
for i in range(2): eng.gen_grad(ds, gs, z, c, 5)
torch.cuda.synchronize()
eng.profile((1 << 2) | (1 << 1))
for i in range(5): eng.gen_grad(ds, gs, z, c, 5)
w_ms, w_n = eng.profile_read(2); d_ms, d_n = eng.profile_read(1)
print(os.path.basename(lib), "G wgrad total per step ms %.3f (%d launches)" % (w_ms / 5, w_n / 5), "G dgrad per step ms %.3f" % (d_ms / 5))
