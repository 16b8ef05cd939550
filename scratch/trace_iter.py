"""One traced training iteration (bs 256, nd 16) for rocprofv3 --kernel-trace; see scripts/gpu_trace.sh."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer, synthetic_batch_device
import numpy as np
B = int(os.environ.get("TRACE_B", 256)); nd = int(os.environ.get("TRACE_ND", 16))
eng = Engine(ndomain=nd, max_batch=B)
for kv in os.environ.get("TRACE_OPTS", "").split(","):
    if kv:
        k, v = kv.split("="); eng.set_option(k, int(v))
rng = np.random.default_rng(0)
tr = WGANGPTrainer(eng, W.init_generator(rng, nd), W.init_critic(rng, nd), n_disc=1)
x, c, z = synthetic_batch_device(B, nd, 1, eng.device)
for it in range(4):
    tr.iteration([(x, c, z)], (z, c))
torch.cuda.synchronize()
