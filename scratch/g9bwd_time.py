"""Time k_g9_bwd_mfma16 against k_g9_bwd_pairs inside the generator step (bf16 storage mode, bs 2048): HIP events around gen_grad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import synthetic_batch_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eng = Engine(16, B); eng.set_option("bf16", 1)
rng = np.random.default_rng(0)
gs, ds = eng.to_slab(W.init_generator(rng, 16)), eng.to_slab(W.init_critic(rng, 16))
x, c, z = synthetic_batch_device(B, 16, 1, eng.device)
for on in (0, 1, 0, 1):
    eng.set_option("g9_bwd_mfma", on)
    for _ in range(3): eng.gen_grad(ds, gs, z, c, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): eng.gen_grad(ds, gs, z, c, 5)
    e1.record(); torch.cuda.synchronize()
    print("g9_bwd_mfma", on, "generator step ms", e0.elapsed_time(e1) / 10)
