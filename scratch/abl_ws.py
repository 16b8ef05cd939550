import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import synthetic_batch_device
eng = Engine(16, 256)
rng = np.random.default_rng(0)
gs = eng.to_slab(W.init_generator(rng, 16)); ds = eng.to_slab(W.init_critic(rng, 16))
x, c, z = synthetic_batch_device(256, 16, 1, eng.device)
for ws in (0, 1, 0, 1):
    eng.set_option("wave_specialized", ws)
    for i in range(3): eng.gen_forward(gs, z, c)
    torch.cuda.synchronize()
    eng.profile((1 << 5) | (1 << 0) | (1 << 1))
    for i in range(10): eng.gen_forward(gs, z, c)
    m5, n5 = eng.profile_read(5); m0, n0 = eng.profile_read(0)
    for i in range(3): eng.gen_grad(ds, gs, z, c, 5)
    torch.cuda.synchronize()
    eng.profile((1 << 1) | (1 << 2))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(5): eng.gen_grad(ds, gs, z, c, 5)
    t1.record(); torch.cuda.synchronize()
    d_ms, dn = eng.profile_read(1)
    print(f"ws={ws}: G3 fwd {m5/n5:.3f} ms  G1+G2 fwd avg {m0/n0:.3f} ms  G dgrad/step {d_ms/5:.3f} ms  gen_grad total {t0.elapsed_time(t1)/5:.2f} ms")
