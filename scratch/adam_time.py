"""k_adam on the generator slab of ndomain 64 (209 M parameters): python3 scratch/adam_time.py [lib.so]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pr_disagg_radar_gan_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from pr_disagg_radar_gan_amd import Engine
eng = Engine(16, 2)
n = 209_300_000
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 1e-3; v = torch.rand(n, device="cuda") * 1e-6
for _ in range(3):
    eng.adam(p, g, v, 5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    eng.adam(p, g, v, 6 + i)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(os.path.basename(_lib.LIB_PATH), "k_adam %.3f ms  %.2f TB/s" % (ms, 20.0 * n / ms / 1e9))
