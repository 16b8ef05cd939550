import sys, os, ctypes
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
L = eng.lib
buf = (ctypes.c_ulonglong * 8)()
L.rdgan_debug_stamps(buf, 1)
for i in range(5): eng.gen_forward(gs, z, c)
L.rdgan_debug_stamps(buf, 1)
v = list(buf)
tot = v[4]
print("waves", v[5], "cycles/wave", tot / max(v[5], 1))
for n, x_ in zip(("load-issue", "mfma+ldsread", "wait+store", "barrier"), v[:4]):
    print(f"{n:14s} {100 * x_ / tot:5.1f}%  {x_ / max(v[5],1):9.0f} cyc/wave")
print("prologue cyc/wave %.0f  epilogue cyc/wave %.0f  (loop %.0f)" % (v[6] / max(v[5],1), v[7] / max(v[5],1), tot / max(v[5],1)))
