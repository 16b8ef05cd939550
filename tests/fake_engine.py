"""Oracle-backed stand-in for the HIP engine so that the data-parallel logic of
pr_disagg_radar_gan_amd.trainer (sharding, one all-reduce per update, 1/world in Adam, shared
iteration counter) can be exercised on CPU with gloo.  Test infrastructure only."""
import numpy as np
import torch

from oracle import rdgan_torch as ot
from pr_disagg_radar_gan_amd import weights as W

LOSS_SLOTS = 8


class FakeEngine:
    def __init__(self, ndomain=16, dtype=torch.float32):
        self.ndomain = ndomain
        self.dtype = dtype              # arithmetic of the oracle behind the fp32 slabs (float64: rounding-free comparisons)
        self.sample_offset = 0          # the HIP engine's "sample_offset" option
        self.device = torch.device("cpu")
        self.gen_shapes = W.gen_param_shapes(ndomain)
        self.critic_shapes = W.critic_param_shapes(ndomain)
        self.n_gen = W.param_count(self.gen_shapes)
        self.n_critic = W.param_count(self.critic_shapes)

    def to_slab(self, arrays):
        return torch.from_numpy(W.flatten(arrays)).clone()

    def set_option(self, name, value):
        if name != "sample_offset":
            raise ValueError(f"FakeEngine: unknown option {name}")
        self.sample_offset = int(value)

    def _unpack(self, slab, shapes):
        out, off = [], 0
        for _, s in shapes:
            n = int(np.prod(s))
            out.append(slab[off:off + n].reshape(s).to(self.dtype))
            off += n
        return out

    def critic_grad(self, dparams, gparams, x_real, cond, z, seed, grad_out=None):
        losses, grads = ot.critic_step_grads(self._unpack(dparams, self.critic_shapes), self._unpack(gparams, self.gen_shapes),
                                             x_real.to(self.dtype), cond.to(self.dtype), z.to(self.dtype), seed,
                                             alpha_offset=self.sample_offset)
        grad_out[:self.n_critic] = torch.cat([g.reshape(-1) for g in grads]).float()
        grad_out[self.n_critic:] = 0
        grad_out[self.n_critic:self.n_critic + 4] = losses
        return grad_out

    def gen_grad(self, dparams, gparams, z, cond, seed, grad_out=None):
        loss, grads = ot.gen_step_grads(self._unpack(dparams, self.critic_shapes), self._unpack(gparams, self.gen_shapes),
                                        z.to(self.dtype), cond.to(self.dtype), seed)
        grad_out[:self.n_gen] = torch.cat([g.reshape(-1) for g in grads]).float()
        grad_out[self.n_gen:] = 0
        grad_out[self.n_gen] = loss
        return grad_out

    def adam(self, params, grad, v, t, lr=1e-4, beta2=0.9, eps=1e-7, grad_scale=1.0):
        n = params.numel()
        ot.adam_update([params], [grad[:n] * grad_scale], [v], t, lr, beta2, eps)
