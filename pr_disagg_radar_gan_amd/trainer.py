"""The training iteration of the reference's ``train()`` (gan_train_cwgangp_pixelnorm.py:466-482)
on the HIP engine, data-parallel over ranks.

One iteration = ``n_disc`` critic ``train_on_batch`` calls then one generator
``train_on_batch`` (reference :468-482).  Each rank computes the gradient slab of its
minibatch shard with librdgan_hip.so, the slabs are summed with ONE RCCL all-reduce per
optimizer update (torch.distributed backend "nccl" is RCCL on ROCm; the four loss scalars
ride in the slab's tail), and the fused Adam kernel applies 1/world.  Weights and Adam state
are replicated and stay bit-identical across ranks.  Both models share one Adam iteration
counter, as the single ``tf.optimizers.Adam`` object of the reference does (:385,:391,:408).
"""
import numpy as np
import torch

LOSS_SLOTS = 8


class WGANGPTrainer:
    """overlap (default: world_size > 1 on a GPU engine): the gradient all-reduce and the Adam update of an optimizer step
    run on a separate communication stream.  The next gradient call starts right away on the compute stream: its generator
    forward reads no critic weight, so it runs beside the critic's exchange + update, and the compute stream only waits for
    the `critic ready` event in front of the first kernel that reads critic weights (rdgan_*_grad_after).  Per iteration the
    n_disc critic exchanges hide behind generator forwards; the generator's own exchange is exposed (the next critic step's
    first kernel needs the new generator weights).  comm_hook(slab): called on the communication stream in place of / in
    front of the all-reduce (tests: a delay kernel that would expose a missing dependency)."""

    def __init__(self, engine, gen_arrays, critic_arrays, n_disc=5, lr=1e-4, beta2=0.9, eps=1e-7,
                 process_group=None, world_size=1, rank=0, base_seed=1234, overlap=None, comm_hook=None):
        self.eng = engine
        self.n_disc = int(n_disc)
        self.lr, self.beta2, self.eps = lr, beta2, eps
        self.pg, self.world, self.rank = process_group, int(world_size), int(rank)
        self.gparams = engine.to_slab(gen_arrays)
        self.dparams = engine.to_slab(critic_arrays)
        self.gv = torch.zeros_like(self.gparams)
        self.dv = torch.zeros_like(self.dparams)
        self.ggrad = torch.zeros(self.gparams.numel() + LOSS_SLOTS, dtype=torch.float32, device=self.gparams.device)
        self.dgrad = torch.zeros(self.dparams.numel() + LOSS_SLOTS, dtype=torch.float32, device=self.dparams.device)
        self.t = 0                      # shared optimizer.iterations
        self.base_seed = int(base_seed)
        self.calls = 0
        on_gpu = self.gparams.is_cuda
        self.overlap = bool(on_gpu and (self.world > 1 if overlap is None else overlap))
        self.comm_hook = comm_hook
        self.comm = torch.cuda.Stream(device=self.gparams.device) if self.overlap else None
        self.d_ready = None             # events recorded on the communication stream behind the last critic / generator update
        self.g_ready = None

    # every stochastic draw inside a step (dropout masks, alpha) is keyed by (base_seed, call index, rank)
    def _next_seed(self):
        self.calls += 1
        s = (self.base_seed * 0x9E3779B97F4A7C15 + self.calls * 0xD1B54A32D192ED03 + self.rank * 0x94D049BB133111EB)
        s &= 0xFFFFFFFFFFFFFFFF
        return s or 1

    def _allreduce(self, slab):
        if self.comm_hook is not None:
            self.comm_hook(slab)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(slab, op=dist.ReduceOp.SUM, group=self.pg)

    def _update(self, params, grad, v):
        """exchange + optimizer half of a train_on_batch call: ONE all-reduce of the flat gradient slab (losses in its
        tail), then the fused Adam kernel with 1/world folded in.  Returns [total, valid, fake, gp, nonfinite] averaged
        over ranks.  Runs on the current stream."""
        self._allreduce(grad)
        self.t += 1
        self.eng.adam(params, grad, v, self.t, self.lr, self.beta2, self.eps, 1.0 / self.world)
        tail = grad[-LOSS_SLOTS:-LOSS_SLOTS + 5]
        return tail if self.world == 1 else tail / self.world      # (a view at world 1: no extra kernel on the step path)

    def _update_overlapped(self, params, grad, v, which):
        cur = torch.cuda.current_stream(params.device)
        done = torch.cuda.Event()
        done.record(cur)                               # the gradient slab is complete
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(done)
            losses = self._update(params, grad, v)
            ready = torch.cuda.Event()
            ready.record(self.comm)
        losses.record_stream(cur)
        setattr(self, which, ready)
        return losses

    def critic_step(self, x_real, cond, z, seed=None):
        """critic_model.train_on_batch([X_real, cond_real, latent], [valid, fake, dummy]) (reference :472).
        Returns the device tensor [total, valid, fake, gp, nonfinite] averaged over ranks."""
        seed = self._next_seed() if seed is None else seed
        if not self.overlap:
            self.eng.critic_grad(self.dparams, self.gparams, x_real, cond, z, seed, grad_out=self.dgrad)
            return self._update(self.dparams, self.dgrad, self.dv)
        cur = torch.cuda.current_stream(self.dparams.device)
        if self.g_ready is not None:
            cur.wait_event(self.g_ready)               # the generator forward reads the generator weights at once
        # critic weights, their Adam state and the gradient slab are touched only behind the wait for d_ready
        self.eng.critic_grad(self.dparams, self.gparams, x_real, cond, z, seed, grad_out=self.dgrad, critic_ready=self.d_ready)
        return self._update_overlapped(self.dparams, self.dgrad, self.dv, "d_ready")

    def gen_step(self, z, cond, seed=None):
        """generator_model.train_on_batch([latent, cond], valid) (reference :482)."""
        seed = self._next_seed() if seed is None else seed
        if not self.overlap:
            self.eng.gen_grad(self.dparams, self.gparams, z, cond, seed, grad_out=self.ggrad)
            return self._update(self.gparams, self.ggrad, self.gv)
        cur = torch.cuda.current_stream(self.gparams.device)
        if self.g_ready is not None:
            cur.wait_event(self.g_ready)
        self.eng.gen_grad(self.dparams, self.gparams, z, cond, seed, grad_out=self.ggrad, critic_ready=self.d_ready)
        return self._update_overlapped(self.gparams, self.ggrad, self.gv, "g_ready")

    def join(self):
        """Make the current stream wait for the updates still running on the communication stream (no host sync)."""
        if self.overlap:
            cur = torch.cuda.current_stream(self.gparams.device)
            for ev in (self.d_ready, self.g_ready):
                if ev is not None:
                    cur.wait_event(ev)

    def iteration_raw(self, critic_batches, gen_batch):
        """The same iteration without any arithmetic on the loss tails: returns (critic tail, generator tail), each
        [total, valid, fake, gp, nonfinite] as left by the LAST critic step / the generator step (views into the gradient
        slabs at world 1: read them before the next iteration).  For loops that look at the losses only now and then."""
        assert len(critic_batches) == self.n_disc
        for (x, c, z) in critic_batches:
            dl = self.critic_step(x, c, z)
        gl = self.gen_step(*gen_batch)
        self.join()
        return dl, gl

    def iteration(self, critic_batches, gen_batch):
        """critic_batches: n_disc tuples (x_real, cond, z); gen_batch: (z, cond).  Returns (d_loss, g_loss)
        device scalars with the reference's reporting: d_loss = mean(valid_loss, fake_loss) of the LAST
        critic step (reference :475), g_loss = generator loss."""
        assert len(critic_batches) == self.n_disc
        for (x, c, z) in critic_batches:
            dl = self.critic_step(x, c, z)
        gl = self.gen_step(*gen_batch)
        self.join()         # the returned scalars (and the weights) are safe to read on the current stream; the next
        #                     iteration's first kernel needs the new generator weights anyway
        return 0.5 * (dl[1] + dl[2]), gl[0], torch.maximum(dl[4], gl[4])

    def state_arrays(self):
        """(generator arrays, critic arrays) in Keras weight order, as numpy."""
        from . import weights as W
        self.join()
        return (W.unflatten(self.gparams.cpu().numpy(), self.eng.gen_shapes),
                W.unflatten(self.dparams.cpu().numpy(), self.eng.critic_shapes))


    # ---- resume checkpoint (the reference saves weights only, T:520-521, and cannot resume)
    def save_checkpoint(self, path, extra=None):
        """Everything a bit-identical continuation needs: both weight slabs, both Adam second-moment slabs, the
        shared Adam iteration counter, the step-RNG position (base_seed, calls) and numpy's global RNG state (the
        reference draws batches and latents from it, T:150,179).  One .npz; rank 0 writes (replicas are identical)."""
        st = np.random.get_state()
        self.join()
        np.savez(path, format=np.array("rdgan-checkpoint-1"), ndomain=self.eng.ndomain,
                 n_cond_channels=getattr(self.eng, "n_cond_channels", 1),
                 gparams=self.gparams.cpu().numpy(), dparams=self.dparams.cpu().numpy(),
                 gv=self.gv.cpu().numpy(), dv=self.dv.cpu().numpy(),
                 t=self.t, calls=self.calls, base_seed=self.base_seed, n_disc=self.n_disc,
                 hyper=np.array([self.lr, self.beta2, self.eps], np.float64),
                 np_rng_keys=st[1], np_rng_pos=np.array([st[2], st[3]], np.int64), np_rng_gauss=np.float64(st[4]),
                 extra=np.array(repr(extra) if extra is not None else ""))

    def load_checkpoint(self, path, restore_numpy_rng=True):
        """Inverse of save_checkpoint, in place (device slabs keep their addresses, so models that adopted them
        keep tracking).  Raises ValueError when the file belongs to another configuration."""
        self.join()
        with np.load(path, allow_pickle=False) as f:
            if str(f["format"]) != "rdgan-checkpoint-1":
                raise ValueError(f"{path}: not an rdgan checkpoint")
            if int(f["ndomain"]) != self.eng.ndomain or int(f["n_cond_channels"]) != getattr(self.eng, "n_cond_channels", 1):
                raise ValueError(f"{path}: checkpoint of ndomain {int(f['ndomain'])} / {int(f['n_cond_channels'])} condition "
                                 f"channels does not fit this engine")
            for name in ("gparams", "dparams", "gv", "dv"):
                dst, src = getattr(self, name), f[name]
                if src.shape != tuple(dst.shape) or src.dtype != np.float32:
                    raise ValueError(f"{path}: {name} has shape {src.shape}, expected {tuple(dst.shape)}")
                dst.copy_(torch.from_numpy(src))
            self.t, self.calls, self.base_seed = int(f["t"]), int(f["calls"]), int(f["base_seed"])
            if restore_numpy_rng:
                pos = f["np_rng_pos"]
                np.random.set_state(("MT19937", f["np_rng_keys"], int(pos[0]), int(pos[1]), float(f["np_rng_gauss"])))


def shard_slice(global_batch, world, rank):
    """rank r takes samples [r*B/W, (r+1)*B/W) -- equal shards so the global mean is the mean of local means"""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} not divisible by world size {world}")
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)


def synthetic_batch_device(batch, ndomain, seed, device):
    """Synthetic inputs of SURVEY 8(d), generated on the device by torch (plumbing): real tiles =
    softmax over hours of 2*N(0,1) (values in [0,1], sum over hours 1, as the reference asserts at
    :167-172), cond = Gamma(2, 5 mm)/127.4, z ~ N(0,1)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    x = torch.softmax(2.0 * torch.randn((batch, 24, ndomain, ndomain, 1), generator=g, device=device), dim=1).contiguous()
    shape = (batch, ndomain, ndomain, 1)
    # Gamma(k=2, theta=5) = -5*(log u1 + log u2)
    u = torch.rand((2,) + shape, generator=g, device=device).clamp_min(1e-12)
    cond = (-5.0 * (u[0].log() + u[1].log()) / 127.4).contiguous()
    z = torch.randn((batch, 100), generator=g, device=device)
    return x, cond, z
