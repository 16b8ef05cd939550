import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pr_disagg_radar_gan_amd import gan_train_cwgangp_pixelnorm as T, ensemble
T.configure(ndomain=16)
gen = T.create_generator(seed=2)
rng = np.random.default_rng(0)
real = (rng.gamma(0.3, 2.0, (24, 16, 16)) + 1e-3).astype(np.float32)
for _ in range(2): ensemble.crps_for_day(gen, real, 1000, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
days = 20
for d in range(days): ensemble.crps_for_day(gen, real, 1000, seed=d)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"crps_for_day: {days/dt:.1f} days/s = {1000*days/dt:.0f} scenarios/s incl. CRPS (reference job: 10 000 days x 1000 scenarios, 2-day V100 limit)")
