"""Cumulative s_memtime stamps of the block-3 difference-part launch (ws<256,64,4,1> with addt) from a -DRD_STAMP build."""
import sys, os, ctypes
lib = sys.argv[1]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pr_disagg_radar_gan_amd import _lib
_lib.LIB_PATH = lib
import numpy as np, torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import synthetic_batch_device
eng = Engine(16, 256)
for kv in sys.argv[2:]:
    k, v = kv.split("="); eng.set_option(k, int(v))
rng = np.random.default_rng(0)
gs = eng.to_slab(W.init_generator(rng, 16))
x, c, z = synthetic_batch_device(256, 16, 1, eng.device)
for i in range(3): eng.gen_forward(gs, z, c)
L = eng.lib
buf = (ctypes.c_ulonglong * 8)()
L.rdgan_debug_stamps_ws(buf, 1)
for i in range(5): eng.gen_forward(gs, z, c)
L.rdgan_debug_stamps_ws(buf, 1)
v = list(buf); n = max(v[5], 1)
print("workgroups", v[5])
for name, x_ in zip(("first MFMA at", "loop end at", "tile in LDS at", "kernel end at", "loader: first DMA issued at"), v[:5]):
    print(f"{name:28s} {x_ / n:9.0f} memtime ticks/WG")
