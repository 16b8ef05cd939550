"""Helpers for the -m gpu parity tests: call the C ABI with torch CUDA tensors."""
import ctypes

import numpy as np
import torch

from pr_disagg_radar_gan_amd import _lib


def dev(a, dtype=torch.float32):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    return a.to(dtype).contiguous().cuda()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def lib():
    return _lib.load()     # raises loudly if the HIP library is missing


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def hip_gates(eng, B):
    """LeakyReLU slope pattern (True = slope 1) of the engine's last generator-step forward: generator h0..h3 and critic
    layers 1..4 over B samples, from the activations left in the workspace (rdgan_debug_activation).  A dropped critic
    activation reads 0 -> False, which is immaterial: its gradient is multiplied by the zero mask anyway."""
    from oracle import rdgan_np as onp
    nd = eng.ndomain
    s = nd // 8
    gshape = [(B, 3, s, s, 256), (B, 6, 2 * s, 2 * s, 256), (B, 12, 4 * s, 4 * s, 128), (B, 24, 8 * s, 8 * s, 64)]
    geo = onp.critic_geometry(nd)
    chans = (64, 128, 256, 256)
    g = [eng.debug_activation(i, shp).cpu() > 0 for i, shp in enumerate(gshape)]
    d = [eng.debug_activation(4 + li, (B,) + tuple(geo[li][1]) + (chans[li],)).cpu() > 0 for li in range(4)]
    return g, d
