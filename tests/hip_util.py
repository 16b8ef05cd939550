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
