"""Python object over the C ABI of include/rdgan.h: one Engine per GPU.

PyTorch is plumbing only (device memory, streams, torch.distributed); every FLOP and byte
of the step goes through librdgan_hip.so.  No CPU fallback.
"""
import ctypes
import itertools

import numpy as np
import torch

from . import _lib
from . import weights as W

LOSS_SLOTS = 8

_versions = itertools.count(1)


def new_version():
    """A process-wide fresh content version for a weight slab (rdgan_set_weight_versions): whoever writes a slab takes a
    new one, so two slabs -- or two states of one slab -- never share a number, whichever object owns them."""
    return next(_versions)


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.RdganError("no MI355X/ROCm device visible: the cWGAN-GP hot path runs only on the HIP "
                              "library (there is deliberately no CPU fallback)")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _chk_tensor(t, shape, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous float32 CUDA tensor")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


class Engine:
    def __init__(self, ndomain=16, max_batch=256, n_cond_channels=1, device=None):
        require_gpu()
        self.lib = _lib.load()
        self.ndomain = int(ndomain)
        self.max_batch = int(max_batch)
        self.n_cond_channels = int(n_cond_channels)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.rdgan_create(ctypes.byref(self._h), self.ndomain, self.n_cond_channels, self.max_batch)
        if rc != 0:
            raise _lib.RdganError(f"rdgan_create(ndomain={ndomain}, n_cond_channels={n_cond_channels}, max_batch={max_batch}) "
                                  f"failed with code {rc}")
        self.n_gen = int(self.lib.rdgan_gen_param_count(self._h))
        self.n_critic = int(self.lib.rdgan_critic_param_count(self._h))
        self.gen_shapes = W.gen_param_shapes(self.ndomain, self.n_cond_channels)
        self.critic_shapes = W.critic_param_shapes(self.ndomain, self.n_cond_channels)
        assert self.n_gen == W.param_count(self.gen_shapes) and self.n_critic == W.param_count(self.critic_shapes)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.rdgan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace_bytes(self):
        return int(self.lib.rdgan_workspace_bytes(self._h))

    def to_slab(self, arrays):
        return torch.from_numpy(W.flatten(arrays)).to(self.device)

    def _check_batch(self, B, cap):
        if not (1 <= B <= cap):
            raise ValueError(f"batch {B} outside [1, {cap}] (engine created with max_batch={self.max_batch})")

    def _versions(self, gen_version, critic_version):
        """content versions of the slabs of the next call (0 = unknown: the weight forms are rebuilt)"""
        _lib.check(self.lib.rdgan_set_weight_versions(self._h, ctypes.c_uint64(int(gen_version)), ctypes.c_uint64(int(critic_version))),
                   self._h, "rdgan_set_weight_versions")

    def form_builds(self):
        """(generator, critic): how many times this engine has built each network's weight forms (rdgan_form_builds)"""
        g, c = ctypes.c_long(), ctypes.c_long()
        _lib.check(self.lib.rdgan_form_builds(self._h, ctypes.byref(g), ctypes.byref(c)), self._h, "rdgan_form_builds")
        return g.value, c.value

    # ---- entry points.  gen_version / critic_version: the caller's content version of the slab it passes (new_version()
    # after every write); with it the engine skips rebuilding the weight forms of a network that has not changed since the
    # last call.  0 (default) = unknown: always rebuild.
    def gen_forward(self, gen_params, z, cond, out=None, gen_version=0):
        B = z.shape[0]
        nd = self.ndomain
        self._check_batch(B, self.max_batch)
        _chk_tensor(gen_params, (self.n_gen,), "gen_params")
        _chk_tensor(z, (B, W.LATENT_DIM), "z")
        _chk_tensor(cond, (B, nd, nd, self.n_cond_channels), "cond")
        if out is None:
            out = torch.empty((B, W.NHOURS, nd, nd, 1), dtype=torch.float32, device=self.device)
        _chk_tensor(out, (B, W.NHOURS, nd, nd, 1), "out")
        self._versions(gen_version, 0)
        _lib.check(self.lib.rdgan_gen_forward(self._h, _ptr(gen_params), _ptr(z), _ptr(cond), _ptr(out), B, self._stream()),
                   self._h, "rdgan_gen_forward")
        return out

    def check_numerics(self):
        """The reference's ``tf.debugging.check_numerics(x, 'found nan in output of per_gridpoint_softmax')`` (T:349-350):
        waits for the current stream and raises NumericsError if the generator output of the last call held NaN/Inf."""
        rc = self.lib.rdgan_check_numerics(self._h, self._stream())
        if rc == -1:
            raise _lib.NumericsError("found nan in output of per_gridpoint_softmax")
        _lib.check(rc, self._h, "rdgan_check_numerics")

    def critic_forward(self, critic_params, sample, cond, seed=0, critic_version=0):
        B = sample.shape[0]
        nd = self.ndomain
        self._check_batch(B, 3 * self.max_batch)
        _chk_tensor(critic_params, (self.n_critic,), "critic_params")
        _chk_tensor(sample, (B, W.NHOURS, nd, nd, 1), "sample")
        _chk_tensor(cond, (B, nd, nd, self.n_cond_channels), "cond")
        out = torch.empty((B, 1), dtype=torch.float32, device=self.device)
        self._versions(0, critic_version)
        _lib.check(self.lib.rdgan_critic_forward(self._h, _ptr(critic_params), _ptr(sample), _ptr(cond), _ptr(out), B,
                                                 ctypes.c_uint64(seed), self._stream()), self._h, "rdgan_critic_forward")
        return out

    @staticmethod
    def _event_handle(ev):
        """hipEvent_t of a recorded torch.cuda.Event (None -> NULL)"""
        return ctypes.c_void_p(ev.cuda_event if ev is not None else 0)

    def critic_grad(self, critic_params, gen_params, x_real, cond, z, seed, grad_out=None, critic_ready=None,
                    gen_version=0, critic_version=0):
        """critic_ready: torch.cuda.Event recorded (on another stream) behind the last update of critic_params; the
        current stream waits for it after the generator forward (rdgan_critic_grad_after)."""
        B = x_real.shape[0]
        nd = self.ndomain
        self._check_batch(B, self.max_batch)
        _chk_tensor(critic_params, (self.n_critic,), "critic_params")
        _chk_tensor(gen_params, (self.n_gen,), "gen_params")
        _chk_tensor(x_real, (B, W.NHOURS, nd, nd, 1), "x_real")
        _chk_tensor(cond, (B, nd, nd, self.n_cond_channels), "cond")
        _chk_tensor(z, (B, W.LATENT_DIM), "z")
        if grad_out is None:
            grad_out = torch.empty(self.n_critic + LOSS_SLOTS, dtype=torch.float32, device=self.device)
        _chk_tensor(grad_out, (self.n_critic + LOSS_SLOTS,), "grad_out")
        self._versions(gen_version, critic_version)
        _lib.check(self.lib.rdgan_critic_grad_after(self._h, _ptr(critic_params), _ptr(gen_params), _ptr(x_real), _ptr(cond),
                                                    _ptr(z), ctypes.c_uint64(seed), _ptr(grad_out), B,
                                                    self._event_handle(critic_ready), self._stream()),
                   self._h, "rdgan_critic_grad")
        return grad_out

    def gen_grad(self, critic_params, gen_params, z, cond, seed, grad_out=None, critic_ready=None,
                 gen_version=0, critic_version=0):
        B = z.shape[0]
        nd = self.ndomain
        self._check_batch(B, self.max_batch)
        _chk_tensor(critic_params, (self.n_critic,), "critic_params")
        _chk_tensor(gen_params, (self.n_gen,), "gen_params")
        _chk_tensor(z, (B, W.LATENT_DIM), "z")
        _chk_tensor(cond, (B, nd, nd, self.n_cond_channels), "cond")
        if grad_out is None:
            grad_out = torch.empty(self.n_gen + LOSS_SLOTS, dtype=torch.float32, device=self.device)
        _chk_tensor(grad_out, (self.n_gen + LOSS_SLOTS,), "grad_out")
        self._versions(gen_version, critic_version)
        _lib.check(self.lib.rdgan_gen_grad_after(self._h, _ptr(critic_params), _ptr(gen_params), _ptr(z), _ptr(cond),
                                                 ctypes.c_uint64(seed), _ptr(grad_out), B,
                                                 self._event_handle(critic_ready), self._stream()),
                   self._h, "rdgan_gen_grad")
        return grad_out

    def adam(self, params, grad, v, t, lr=1e-4, beta2=0.9, eps=1e-7, grad_scale=1.0):
        n = params.numel()
        _chk_tensor(params, (n,), "params")
        _chk_tensor(v, (n,), "v")
        if grad.numel() < n:
            raise ValueError("grad slab shorter than params")
        _chk_tensor(grad, None, "grad")
        _lib.check(self.lib.rdgan_adam(_ptr(params), _ptr(grad), _ptr(v), n, int(t), float(lr), float(beta2), float(eps),
                                       float(grad_scale), self._stream()), self._h, "rdgan_adam")

    def set_option(self, name, value):
        """rdgan_set_option (include/rdgan.h): "collapse", "fast_fwd", "fast_bwd" (exact algebraic forms), "bf16" (bf16 storage
        mode), "wave_specialized", "ws_ksplit", "tapgather", "g9_direct" (kernel variants), "sample_offset" (data-parallel
        tests)."""
        _lib.check(self.lib.rdgan_set_option(self._h, name.encode(), int(value)), self._h, "rdgan_set_option")

    def profile(self, tag_mask):
        _lib.check(self.lib.rdgan_profile(self._h, int(tag_mask)), self._h, "rdgan_profile")

    def debug_activation(self, which, shape):
        """test hook (rdgan_debug_activation): activation tensor `which` of the last forward (0..3 generator h0..h3,
        4..7 critic layers 1..4) as a float32 CUDA tensor of `shape`"""
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.rdgan_debug_activation(self._h, int(which), _ptr(out), out.numel(), self._stream()), self._h,
                   "rdgan_debug_activation")
        return out

    def flop_count(self, reset=False):
        """algorithmic FLOPs of every GEMM launched since the last reset (rdgan_flop_count)"""
        f = ctypes.c_double()
        _lib.check(self.lib.rdgan_flop_count(self._h, ctypes.byref(f), int(bool(reset))), self._h, "rdgan_flop_count")
        return f.value

    def profile_launches(self, on):
        _lib.check(self.lib.rdgan_profile_launches(self._h, int(bool(on))), self._h, "rdgan_profile_launches")

    def launch_table(self):
        """rows of rdgan_launch_table as dicts: name, kernel, kind, batch, launches, gflop (summed), ms (summed)"""
        buf = (_lib.LaunchStat * 256)()
        n = ctypes.c_int()
        _lib.check(self.lib.rdgan_launch_table(self._h, buf, 256, ctypes.byref(n)), self._h, "rdgan_launch_table")
        kinds = {0: "gemm", 1: "wgrad", 2: "edge"}
        return [dict(name=r.name.decode(), kernel=r.kernel.decode(), kind=kinds.get(r.kind, "?"), batch=r.batch,
                     launches=r.launches, gflop=r.gflop, ms=r.ms) for r in buf[:n.value]]

    def profile_read(self, tag):
        ms, n = ctypes.c_double(), ctypes.c_long()
        _lib.check(self.lib.rdgan_profile_read(self._h, int(tag), ctypes.byref(ms), ctypes.byref(n)), self._h,
                   "rdgan_profile_read")
        return ms.value, n.value
