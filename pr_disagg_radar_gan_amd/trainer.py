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


SHARD_THRESHOLD_BYTES = 128 << 20      # slabs above this take the sharded exchange by default (ndomain 64 generator: 837 MB)


class WGANGPTrainer:
    """overlap (default: world_size > 1 on a GPU engine): the gradient exchange and the Adam update of an optimizer step
    run on a separate communication stream.  The next gradient call starts right away on the compute stream: its generator
    forward reads no critic weight, so it runs beside the critic's exchange + update, and the compute stream only waits for
    the `critic ready` event in front of the first kernel that reads critic weights (rdgan_*_grad_after).  Per iteration the
    n_disc critic exchanges hide behind generator forwards; the generator's own exchange is exposed (the next critic step's
    first kernel needs the new generator weights).  comm_hook(slab): called on the communication stream in place of / in
    front of the exchange (tests: a delay kernel that would expose a missing dependency).

    exchange: how a gradient slab is exchanged, per network.
      "allreduce"  ONE all-reduce(sum) of the flat slab (losses in its tail), then Adam on the whole slab on every rank.
      "sharded"    reduce-scatter(sum) of the slab, Adam on the 1/world this rank owns (its share of the weights and of the
                   second-moment slab), all-gather of the updated weights; the loss sums ride in 8 spare floats behind the
                   weights.  Same bytes on the wire as a ring all-reduce, Adam over n/world instead of n parameters, and the
                   second-moment slab is only kept current where it is owned (sync_state() gathers it for checkpoints).
      None/"auto"  "sharded" for a slab of at least SHARD_THRESHOLD_BYTES (the 837 MB generator of ndomain 64, whose Dense
                   kernel is 99.6 % of it), else "allreduce" (the 11-16 MB slabs of ndomain 16 are latency-bound: one
                   collective beats two).
    Replicas stay bit-identical either way (every rank receives the same bytes).

    grad_transport ("fp32" | "bf16", sharded exchange only; SURVEY 8e "optional bf16 gradient transport"): "bf16" rounds the
    gradient slab to bfloat16 for the reduce-scatter -- half the bytes of that half of the exchange (the all-gather carries fp32
    master weights either way), i.e. a quarter off the exposed ring time of ndomain 64's 837 MB generator slab.  The sum is then
    formed in bf16 by the collective (relative error ~2^-8 per addend), the loss tail travels in a separate 8-float fp32
    all-reduce; replicas still receive identical bytes.  Off by default: it changes the arithmetic of the update."""

    def __init__(self, engine, gen_arrays, critic_arrays, n_disc=5, lr=1e-4, beta2=0.9, eps=1e-7,
                 process_group=None, world_size=1, rank=0, base_seed=1234, overlap=None, comm_hook=None, exchange=None,
                 grad_transport="fp32"):
        self.eng = engine
        self.n_disc = int(n_disc)
        self.lr, self.beta2, self.eps = lr, beta2, eps
        self.pg, self.world, self.rank = process_group, int(world_size), int(rank)
        if exchange not in (None, "auto", "allreduce", "sharded"):
            raise ValueError(f"exchange must be 'allreduce', 'sharded' or None, not {exchange!r}")
        if grad_transport not in ("fp32", "bf16"):
            raise ValueError(f"grad_transport must be 'fp32' or 'bf16', not {grad_transport!r}")
        self.grad_transport = grad_transport
        g0, d0 = engine.to_slab(gen_arrays), engine.to_slab(critic_arrays)
        self.exchange = {}
        self._pad = {}
        for which, p0 in (("g", g0), ("d", d0)):
            n = p0.numel()
            sharded = self.world > 1 and (exchange == "sharded" or
                                          (exchange in (None, "auto") and 4 * n >= SHARD_THRESHOLD_BYTES))
            self.exchange[which] = "sharded" if sharded else "allreduce"
            # sharded: every slab is padded to `world` equal shards of a multiple of 4 floats that also cover the loss tail
            per = -(-(n + LOSS_SLOTS) // (4 * self.world)) * 4 if sharded else 0
            P = per * self.world if sharded else n
            self._pad[which] = (n, per, P)
            pbuf = torch.zeros(max(P, n), dtype=torch.float32, device=p0.device)
            pbuf[:n].copy_(p0)
            vbuf = torch.zeros_like(pbuf)
            grad = torch.zeros(max(P, n + LOSS_SLOTS), dtype=torch.float32, device=p0.device)
            setattr(self, which + "_pbuf", pbuf)
            setattr(self, which + "_vbuf", vbuf)
            setattr(self, which + "params", pbuf[:n])           # what the engine reads: the first n floats
            setattr(self, which + "v", vbuf[:n])
            setattr(self, which + "grad", grad[:n + LOSS_SLOTS] if not sharded else grad)
            if sharded:
                setattr(self, which + "_gshard", torch.zeros(per, dtype=torch.float32, device=p0.device))
                setattr(self, which + "_pshard", torch.zeros(per, dtype=torch.float32, device=p0.device))
        self.t = 0                      # shared optimizer.iterations
        # content versions of the two weight slabs (engine.new_version(): process-wide unique): a fresh one after every write,
        # so the engine rebuilds a network's weight forms once per update instead of once per call (rdgan_set_weight_versions)
        self.weight_cache = hasattr(engine, "form_builds")
        self.gver = self.dver = 0
        self.weights_changed()
        self.base_seed = int(base_seed)
        self.calls = 0
        on_gpu = self.gparams.is_cuda
        self.overlap = bool(on_gpu and (self.world > 1 if overlap is None else overlap))
        self.comm_hook = comm_hook
        self.comm = torch.cuda.Stream(device=self.gparams.device) if self.overlap else None
        self.d_ready = None             # events recorded on the communication stream behind the last critic / generator update
        self.g_ready = None
        self._native_collectives = None
        self._native_verified = set()   # slabs whose first native sharded exchange has been checked against an all-reduce

    def weights_changed(self, which="gd"):
        """Call after writing a weight slab from outside (set_weights, a checkpoint load): its forms are rebuilt on next use."""
        if self.weight_cache:
            from .engine import new_version
            for w in which:
                setattr(self, w + "ver", new_version())

    def _ver(self):
        return {"gen_version": self.gver, "critic_version": self.dver} if self.weight_cache else {}

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

    def _allreduce_plain(self, t):
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)

    # ---- collectives of the sharded exchange.  RCCL (backend "nccl") has both natively; gloo -- only used to rehearse the
    # data-parallel path on CPU or on one GPU -- has neither for device tensors, so there the same data movement is spelled
    # with what it has: all-reduce + keep the own shard, and one broadcast per shard.
    def _native(self):
        if self._native_collectives is None:
            import torch.distributed as dist
            self._native_collectives = dist.get_backend(self.pg) == "nccl"
        return self._native_collectives

    def _global_rank(self, r):
        import torch.distributed as dist
        return r if self.pg is None or self.pg is dist.group.WORLD else dist.get_global_rank(self.pg, r)

    def _reduce_scatter(self, out, inp):
        import torch.distributed as dist
        if self.comm_hook is not None:
            self.comm_hook(inp)
        if self._native():
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.pg)
        else:
            dist.all_reduce(inp, op=dist.ReduceOp.SUM, group=self.pg)
            per = out.numel()
            out.copy_(inp[self.rank * per:(self.rank + 1) * per])

    def _reduce_scatter_bf16(self, gsh, grad, n, per, P):
        """the reduce-scatter of the sharded exchange with the gradients rounded to bf16 on the wire; the 8 loss sums behind the
        gradients go through their own fp32 all-reduce and are put back where the fp32 path leaves them"""
        import torch.distributed as dist
        if self.comm_hook is not None:
            self.comm_hook(grad[:P])
        tail = grad[n:n + LOSS_SLOTS].clone()
        self._allreduce_plain(tail)
        g16 = grad[:P].to(torch.bfloat16)
        if self._native():
            sh16 = torch.empty(per, dtype=torch.bfloat16, device=grad.device)
            dist.reduce_scatter_tensor(sh16, g16, op=dist.ReduceOp.SUM, group=self.pg)
            gsh.copy_(sh16)
        else:                      # gloo rehearsal: the same rounding of the addends, the sum in fp32 and rounded once
            r32 = g16.float()
            dist.all_reduce(r32, op=dist.ReduceOp.SUM, group=self.pg)
            gsh.copy_(r32[self.rank * per:(self.rank + 1) * per].to(torch.bfloat16))
        lo = self.rank * per
        a, b = max(lo, n), min(lo + per, n + LOSS_SLOTS)
        if b > a:
            gsh[a - lo:b - lo].copy_(tail[a - n:b - n])

    def _all_gather(self, out, shard):
        import torch.distributed as dist
        if self._native():
            dist.all_gather_into_tensor(out, shard, group=self.pg)
        else:
            per = shard.numel()
            out[self.rank * per:(self.rank + 1) * per].copy_(shard)
            for r in range(self.world):
                dist.broadcast(out[r * per:(r + 1) * per], src=self._global_rank(r), group=self.pg)

    def _update(self, which):
        """exchange + optimizer half of a train_on_batch call for network `which` ("d" / "g").  Returns [total, valid, fake,
        gp, nonfinite] averaged over ranks.  Runs on the current stream."""
        params, grad, v = getattr(self, which + "params"), getattr(self, which + "grad"), getattr(self, which + "v")
        n, per, P = self._pad[which]
        if self.exchange[which] == "allreduce":
            # ONE all-reduce of the flat gradient slab (losses in its tail), then the fused Adam kernel with 1/world folded in
            self._allreduce(grad)
            self.t += 1
            self.eng.adam(params, grad, v, self.t, self.lr, self.beta2, self.eps, 1.0 / self.world)
            self.weights_changed(which)
            tail = grad[n:n + 5]
            return tail if self.world == 1 else tail / self.world      # (a view at world 1: no extra kernel on the step path)
        pbuf, vbuf = getattr(self, which + "_pbuf"), getattr(self, which + "_vbuf")
        gsh, psh = getattr(self, which + "_gshard"), getattr(self, which + "_pshard")
        first_native = self._native() and which not in self._native_verified
        if first_native:
            # RCCL's reduce_scatter_tensor / all_gather_into_tensor (in place on views of the padded buffers) had never run
            # on hardware when this was written: the FIRST native sharded exchange of a slab is checked against a plain
            # all-reduce of a copy (one extra collective, once) and fails loudly instead of training on a mixed-up shard
            check = grad[:P].clone()
            self._allreduce_plain(check)
        if self.grad_transport == "bf16":
            self._reduce_scatter_bf16(gsh, grad, n, per, P)
        else:
            self._reduce_scatter(gsh, grad[:P])            # this rank's 1/world of the summed slab (loss tail included)
        if first_native and self.grad_transport == "bf16":
            want = check[self.rank * per:(self.rank + 1) * per]
            tol = 0.05 * float(want.abs().max()) + 1e-30       # (a sum of `world` bf16-rounded addends formed in bf16)
            bad = (~((gsh - want).abs() <= tol).all()).float().reshape(1)
            self._allreduce_plain(bad)
            if float(bad) != 0:
                raise RuntimeError(f"sharded exchange (bf16 transport): reduce_scatter_tensor of slab '{which}' is off by more than "
                                   f"bf16 rounding on rank {self.rank}; use grad_transport='fp32'")
            del check, want
        elif first_native:
            want = check[self.rank * per:(self.rank + 1) * per]
            tol = 1e-5 * float(want.abs().max()) + 1e-30       # (ring order of the two collectives may differ: not bitwise)
            bad = (~((gsh - want).abs() <= tol).all()).float().reshape(1)
            self._allreduce_plain(bad)                         # every rank learns of it: nobody is left alone in the all-gather
            if float(bad) != 0:
                raise RuntimeError(f"sharded exchange: reduce_scatter_tensor of slab '{which}' disagrees with all_reduce on rank "
                                   f"{self.rank} (max diff {float((gsh - want).abs().max()):.3e}); use exchange='allreduce'")
            del check, want
        self.t += 1
        lo = self.rank * per
        hi = min(lo + per, n)
        if hi > lo:                                        # Adam on the owned parameters only
            self.eng.adam(pbuf[lo:hi], gsh[:hi - lo], vbuf[lo:hi], self.t, self.lr, self.beta2, self.eps, 1.0 / self.world)
        a, b = max(lo, n), min(lo + per, n + LOSS_SLOTS)   # the part of the loss tail that fell into this shard ...
        if b > a:
            pbuf[a:b].copy_(gsh[a - lo:b - lo])            # ... rides behind the weights in the all-gather
        psh.copy_(pbuf[lo:lo + per])
        self._all_gather(pbuf[:P], psh)                    # every rank: the updated weights + the loss sums
        if first_native:
            import torch.distributed as dist
            misplaced = 0.0 if torch.equal(pbuf[lo:lo + per], psh) else 1.0
            cs = torch.stack([pbuf[:P].double().sum(), torch.tensor(misplaced, dtype=torch.float64, device=pbuf.device)])
            lo_, hi_ = cs.clone(), cs.clone()
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN, group=self.pg)
            dist.all_reduce(hi_, op=dist.ReduceOp.MAX, group=self.pg)
            if float(hi_[1]) != 0:
                raise RuntimeError("sharded exchange: all_gather_into_tensor put another shard where a rank's own belongs; "
                                   "use exchange='allreduce'")
            if not torch.equal(lo_[:1], hi_[:1]):
                raise RuntimeError("sharded exchange: replicas differ after all_gather_into_tensor; use exchange='allreduce'")
            self._native_verified.add(which)
        self.weights_changed(which)
        return pbuf[n:n + 5] / self.world

    def _update_overlapped(self, which):
        params = getattr(self, which + "params")
        cur = torch.cuda.current_stream(params.device)
        done = torch.cuda.Event()
        done.record(cur)                               # the gradient slab is complete
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(done)
            losses = self._update(which)
            ready = torch.cuda.Event()
            ready.record(self.comm)
        losses.record_stream(cur)
        setattr(self, which + "_ready", ready)
        return losses

    def reduced_grad(self, which):
        """The summed gradient slab of the last update of network `which` times 1/world -- what Adam consumed -- as one
        tensor of at least n floats on every rank (tests).  Sharded exchange: gathered from the owners (a collective)."""
        n, per, P = self._pad[which]
        grad = getattr(self, which + "grad")
        self.join()
        if self.exchange[which] == "allreduce":
            return grad / self.world
        full = torch.empty(P, dtype=torch.float32, device=grad.device)
        self._all_gather(full, getattr(self, which + "_gshard"))
        return full / self.world

    def sync_state(self):
        """Sharded exchange: a rank keeps the Adam second moments current only for the parameters it owns; this gathers the
        slabs so that every rank holds all of them (checkpoints, tests).  A collective: call it on every rank."""
        self.join()
        for which in ("g", "d"):
            if self.exchange[which] == "sharded":
                n, per, P = self._pad[which]
                vbuf, psh = getattr(self, which + "_vbuf"), getattr(self, which + "_pshard")
                psh.copy_(vbuf[self.rank * per:(self.rank + 1) * per])
                self._all_gather(vbuf[:P], psh)

    def _slab(self, which):
        """the gradient slab the engine writes: n gradients + 8 loss slots (the first n + 8 floats of the padded buffer)"""
        return getattr(self, which + "grad")[:self._pad[which][0] + LOSS_SLOTS]

    def critic_step(self, x_real, cond, z, seed=None):
        """critic_model.train_on_batch([X_real, cond_real, latent], [valid, fake, dummy]) (reference :472).
        Returns the device tensor [total, valid, fake, gp, nonfinite] averaged over ranks."""
        seed = self._next_seed() if seed is None else seed
        if not self.overlap:
            self.eng.critic_grad(self.dparams, self.gparams, x_real, cond, z, seed, grad_out=self._slab("d"), **self._ver())
            return self._update("d")
        cur = torch.cuda.current_stream(self.dparams.device)
        if self.g_ready is not None:
            cur.wait_event(self.g_ready)               # the generator forward reads the generator weights at once
        # critic weights, their Adam state and the gradient slab are touched only behind the wait for d_ready
        self.eng.critic_grad(self.dparams, self.gparams, x_real, cond, z, seed, grad_out=self._slab("d"), critic_ready=self.d_ready,
                             **self._ver())
        return self._update_overlapped("d")

    def gen_step(self, z, cond, seed=None):
        """generator_model.train_on_batch([latent, cond], valid) (reference :482)."""
        seed = self._next_seed() if seed is None else seed
        if not self.overlap:
            self.eng.gen_grad(self.dparams, self.gparams, z, cond, seed, grad_out=self._slab("g"), **self._ver())
            return self._update("g")
        cur = torch.cuda.current_stream(self.gparams.device)
        if self.g_ready is not None:
            cur.wait_event(self.g_ready)
        self.eng.gen_grad(self.dparams, self.gparams, z, cond, seed, grad_out=self._slab("g"), critic_ready=self.d_ready,
                          **self._ver())
        return self._update_overlapped("g")

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
    def save_checkpoint(self, path, extra=None, synced=False, write=True):
        """Everything a bit-identical continuation needs: both weight slabs, both Adam second-moment slabs, the
        shared Adam iteration counter, the step-RNG position (base_seed, calls) and numpy's global RNG state (the
        reference draws batches and latents from it, T:150,179).  One .npz.

        With the sharded exchange the second-moment slabs have to be gathered first, and that is a COLLECTIVE: either every
        rank calls save_checkpoint (write=(rank == 0) on all but one keeps the file single), or every rank calls
        sync_state() and the writer passes synced=True (what train() does).  Never call it under `if rank == 0` with
        synced=False: rank 0 would wait in the all-gather alone."""
        st = np.random.get_state()
        if not synced:
            self.sync_state()               # (sharded exchange: gather the second-moment slabs; a collective)
        if not write:
            return
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
            self.weights_changed()
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
