#!/usr/bin/env python3
"""Benchmark of the cWGAN-GP training iteration on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

A "step" is one training iteration of the reference's loop body (gan_train_cwgangp_pixelnorm.py:468-482) on synthetic
24 x nd x nd tiles: n_critic critic updates + 1 generator update.  Configurations (SURVEY 8d numbering = BASELINE.json
configs[] index + 1):

    --config 2 (default)  ndomain 16, bs 256 per GPU, fp32, 1 critic + 1 generator update      (the BASELINE metric; weak scaling)
    --config 3            ndomain 16, bs 2048, bf16 storage, n_critic 5, one GPU               (GP double-backward stress)
    --config 4            ndomain 16, GLOBAL bs 8192 sharded over the GPUs, bf16, n_critic 5   (strong scaling)
    --config 5            ndomain 64, GLOBAL bs 512 sharded over the GPUs, bf16, n_critic 5    (large domain; strong scaling)

For N > 1 there is one process per GPU.  Either the launcher creates them (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE come from the environment), or, when WORLD_SIZE
is unset, this script starts N fresh child processes itself BEFORE it touches the GPU and relays rank 0's line.  The
minibatch dimension is sharded; gradient slabs are summed by RCCL (torch.distributed backend "nccl"), one all-reduce per
optimizer update, on a communication stream beside the next generator forward.  Inputs are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
BF16_MFMA_PEAK_TFLOPS = 2516.6     # dense bf16 (v_mfma_f32_32x32x16_bf16, 32 cycles): 16x the fp32 rate
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable with a float4 copy)

CONFIGS = {
    2: dict(nd=16, batch=256, n_critic=1, bf16=0, scaling="weak",
            name="ndomain=16, 24h, bs=256 fp32, 1 critic step + 1 gen step (BASELINE configs[1])"),
    3: dict(nd=16, batch=2048, n_critic=5, bf16=1, scaling="weak",
            name="ndomain=16, 24h, bs=2048 bf16, n_critic=5 (BASELINE configs[2])"),
    4: dict(nd=16, batch=8192, n_critic=5, bf16=1, scaling="strong",
            name="ndomain=16, 24h, global bs=8192 bf16, n_critic=5, batch-sharded (BASELINE configs[3])"),
    5: dict(nd=64, batch=512, n_critic=5, bf16=1, scaling="strong",
            name="ndomain=64 (largedomain), 24h, global bs=512 bf16, n_critic=5, batch-sharded (BASELINE configs[4])"),
}


def gconv3_flops(batch, nd, taps=4):
    """Algorithmic (executed) FLOPs of ONE launch of the dominant kernel: the difference-part GEMM of the generator's
    third UpSampling3D+Conv3D block (128 -> 64 channels onto the 24 x nd x nd grid) in the shared-centre form
    (DESIGN.md 4.2): 8 output-parity phases x 4 taps on the un-upsampled grid, 2 * B * (24*nd*nd) * 4*128 * 64
    = 402.7 MFLOP per sample at nd=16.  (The shared part T = S x of the same block is a separate, smaller launch.)"""
    return 2.0 * batch * 24 * nd * nd * taps * 128 * 64


def dominant_traffic(bf16, nd, batch):
    """HBM bytes of ONE launch of the dominant kernel from the PMC passes kept under profiles/ (FETCH_SIZE x 2 -- gfx950 counts
    half the bytes of wide streaming reads, MI355X_MICROARCH.md -- plus WRITE_SIZE, separate `rocprofv3 --pmc` passes): read from
    profiles/hbm_traffic_dominant.json, which scripts/summarize_profile.py writes from the counter CSVs of the newest profile
    run (not measurable inside this process).  None when no entry matches this configuration."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic_dominant.json")) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, None
    for e in table.get("entries", []):
        if (bool(e.get("bf16")), e.get("ndomain"), e.get("batch")) == (bool(bf16), nd, batch):
            return e.get("bytes_per_launch"), e.get("source")
    return None, None


# FLOPs of one critic / generator step per sample in the reference's direct 27-tap form, by ndomain; prices
# the measured iteration as "direct-equivalent" TFLOP/s
DIRECT_GF_PER_SAMPLE = {16: {"critic_step": 5.41, "gen_step": 13.54}, 64: {"critic_step": 87.1, "gen_step": 217.9}}


def _host_cores():
    # the GPU box gives one GPU's job a 16-core share; oversubscribing it (torch defaults to every visible core) makes the
    # baseline 20x slower, so use at most 16 threads
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(16, avail))


def cpu_baseline(nd=16, n_critic=1, batch=256, reps=3, small_batch=32):
    """The oracle's torch-CPU restatement of the same arithmetic (kind "port": the reference's own runtime, TensorFlow
    2.1, is not installed and no reference code travels to the GPU box), timed on the host cores on a bounded sample:
    BASELINE configs[0] (generate_scenarios, 10 scenarios for one condition: generator forward only), and the training
    iteration at the reference's default batch (32, T:70) and at the GPU line's batch.  Medians."""
    import numpy as np
    import torch
    from oracle import rdgan_torch as ot
    cores = _host_cores()
    torch.set_num_threads(cores)
    tr = ot.Trainer(ndomain=nd, seed=0)

    def median_time(fn, n, warm=1):
        for _ in range(warm):
            fn(0)
        ts = []
        for k in range(n):
            t0 = time.perf_counter()
            fn(k + 1)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    # configs[0]: example.py -- cond = 10 mm/day everywhere, 10 scenarios
    cond = torch.full((10, nd, nd, 1), 10.0 / 127.4)
    z = torch.randn((10, 100))
    with torch.no_grad():
        t_fwd = median_time(lambda k: ot.generator_forward(tr.gp, z, cond), 5)

    def iteration_at(bs):
        batches = []
        for i in range(n_critic + 1):
            x, c, zz = ot.synthetic_batch(bs, nd, 50 + i)
            batches.append((torch.from_numpy(x), torch.from_numpy(c), torch.from_numpy(zz)))

        def it(k):
            for j in range(n_critic):
                x, c, zz = batches[j]
                tr.critic_step(x, c, zz, seed=1000 + k * 7 + j)
            x, c, zz = batches[n_critic]
            tr.gen_step(zz, c, seed=2000 + k)
        return it

    t_small = median_time(iteration_at(small_batch), reps)
    t_big = median_time(iteration_at(batch), reps) if batch != small_batch else t_small
    return {"value": round(batch / t_big, 3), "unit": "samples/s", "cores": int(cores), "kind": "port",
            "sample": f"median of {reps} iterations (n_critic={n_critic} critic + 1 generator update) at bs={batch} after 1 "
                      f"warm-up, torch-CPU fp32 restatement of the reference arithmetic (oracle/rdgan_torch.py)",
            "bs32_samples_per_s": round(small_batch / t_small, 3),
            "config0_generate_scenarios_n10": {"value": round(10 / t_fwd, 2), "unit": "scenarios/s",
                                               "sample": "generator forward, 10 scenarios of one 16x16 condition "
                                                         "(example.py), median of 5 after 1 warm-up"}}


def launch_children(args, argv):
    """--gpus N without a launcher: N fresh processes, one per GPU, started before this process makes any GPU call (it
    never does); rank 0's JSON line is relayed, the others' output goes to stderr."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    # HSA_ENABLE_IPC_MODE_LEGACY=0: RCCL maps its peers' buffers through hipIpcGetMemHandle, and this image's host driver only
    # supports the dmabuf flavour of IPC -- with the legacy mode the first collective fails with `hipIpcGetMemHandle: invalid
    # argument`.  The image exports the variable already; it is repeated here so that a caller with a scrubbed environment gets
    # the same ranks (an explicit setting of the caller wins).
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's stdout is drained by a thread while every rank is polled: as soon as one rank exits non-zero (a crash at start-up,
    # a failed rendezvous) the others are terminated and that code is returned -- rank 0 must not sit in a collective until
    # the RCCL watchdog fires, holding the GPUs
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = 0
    while True:
        rcs = [p.poll() for p in procs]
        bad = [rc for rc in rcs if rc not in (None, 0)]
        if bad:
            failed = bad[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(rc == 0 for rc in rcs):
            break
        time.sleep(0.2)
    reader.join(timeout=30)
    out0 = (chunks[0] if chunks else b"").decode()
    for line in out0.splitlines():             # (the gloo rehearsal backend prints a connection banner on stdout)
        (sys.stdout if line.startswith("{") and not failed else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return abs(failed) if failed else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="SURVEY 8d configuration number")
    ap.add_argument("--batch", type=int, default=None, help="override: samples per GPU per iteration")
    ap.add_argument("--n-critic", type=int, default=None)
    ap.add_argument("--ndomain", type=int, default=None)
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="weak: --batch per GPU; strong: the configuration's global batch divided over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="exchange + Adam on the compute stream (A/B)")
    ap.add_argument("--grad-transport", choices=("fp32", "bf16"), default="fp32",
                    help="sharded exchange: gradients on the wire as fp32 (default) or rounded to bf16 for the reduce-scatter (optional "
                         "data point: changes the arithmetic of the update)")
    ap.add_argument("--exchange", choices=("auto", "allreduce", "sharded"), default="auto",
                    help="gradient exchange per slab: one all-reduce, or reduce-scatter + Adam on the owned 1/N + all-gather; "
                         "auto = sharded for slabs of at least 128 MB (the ndomain-64 generator)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="rdgan_set_option override for A/B runs (e.g. --opt fast_bwd=0); the default run sets none")
    args = ap.parse_args()

    if args.steps < 1:
        raise SystemExit("bench.py: --steps must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_children(args, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist
    from pr_disagg_radar_gan_amd import Engine, weights as W, _lib
    from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer, synthetic_batch_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("RDGAN_BENCH_FAIL_RANK") == str(rank) and world > 1:      # test hook (tests/test_hip_dp.py): a rank that dies at start-up
        raise SystemExit(3)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        pg = dist.group.WORLD
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                          # every rank really takes part in the collective
        ranks_seen = int(ones.item())
        if ranks_seen != world:
            raise SystemExit(f"bench.py: the first all-reduce saw {ranks_seen} of {world} ranks")

    cfg = dict(CONFIGS[args.config])
    ND = args.ndomain or cfg["nd"]
    n_critic = args.n_critic or cfg["n_critic"]
    scaling = args.scaling or cfg["scaling"]
    if args.batch is not None:
        B, scaling = args.batch, "weak"
    elif scaling == "strong":
        if cfg["batch"] % world:
            raise SystemExit(f"global batch {cfg['batch']} is not divisible by {world} GPUs")
        B = cfg["batch"] // world
    else:
        B = cfg["batch"]
    opts = dict(kv.split("=") for kv in args.opt)
    bf16 = int(opts.get("bf16", cfg["bf16"])) != 0

    eng = Engine(ndomain=ND, max_batch=B, device=dev)
    if cfg["bf16"] and "bf16" not in opts:
        eng.set_option("bf16", 1)
    for name, value in opts.items():
        eng.set_option(name, int(value))
    rng = np.random.default_rng(0)                  # identical initial weights on every rank
    trainer = WGANGPTrainer(eng, W.init_generator(rng, ND), W.init_critic(rng, ND), n_disc=n_critic,
                            process_group=pg, world_size=world, rank=rank, base_seed=1234 + 1000 * args.config,
                            overlap=False if args.no_overlap else None, exchange=args.exchange,
                            grad_transport=args.grad_transport)
    # synthetic inputs resident in HBM; per-rank seeds 1234 + 1000*config + rank
    nbuf = 4 if B * ND * ND <= 256 * 16 * 16 * 8 else 2
    data = []
    base = 1234 + 1000 * args.config + rank
    for i in range(nbuf):
        crit = [synthetic_batch_device(B, ND, base + 97 * (i * (n_critic + 1) + j), dev) for j in range(n_critic)]
        _, c, z = synthetic_batch_device(B, ND, base + 97 * (i * (n_critic + 1) + n_critic) + 13, dev)
        data.append((crit, (z, c)))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def check_replicas(dl_, gl_, when):
        """Data parallel keeps the replicas bit-identical: every rank must hold the same loss tails (they come out of the
        exchange) and the same weights after an update.  Cheap (two all-reduces of 12 doubles), loud: a mis-exchange -- a
        collective that silently mixed up shards, a rank that missed an update -- ends the run instead of producing a number."""
        if world == 1:
            return
        trainer.join()
        v = torch.cat([dl_[:5].double(), gl_[:5].double(), trainer.gparams.double().sum().reshape(1),
                       trainer.dparams.double().sum().reshape(1)])
        lo, hi = v.clone(), v.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit(f"bench.py: replicas differ {when} (rank {rank}): loss tails / weight checksums min {lo.tolist()} max {hi.tolist()}")

    for k in range(args.warmup):
        crit, gen = data[k % nbuf]
        dl_w, gl_w = trainer.iteration_raw(crit, gen)
        if k == 0:
            check_replicas(dl_w, gl_w, "after the first iteration")
    sync()
    eng.profile(1 << _lib.TAG_GCONV3_FWD)            # HIP events around the dominant kernel, on the launch stream
    eng.flop_count(reset=True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        crit, gen = data[k % nbuf]
        ev[k][0].record()
        dl, gl = trainer.iteration_raw(crit, gen)         # loss tails stay on the device; looked at once, below
        ev[k][1].record()
    sync()
    dt = time.perf_counter() - t0
    flops_iter = eng.flop_count() / max(args.steps, 1)
    kern_ms, kern_n = eng.profile_read(_lib.TAG_GCONV3_FWD)
    eng.profile(0)
    it_ms = np.array([a.elapsed_time(b) for a, b in ev]) if args.steps else np.zeros(1)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    check_replicas(dl, gl, "after the timed iterations")
    # (a non-finite value anywhere poisons the weights for good, so the last iteration's flags and losses tell)
    dl, gl = dl.cpu().numpy(), gl.cpu().numpy()
    nonfinite = float(max(dl[4], gl[4]))
    d_loss, g_loss = float(0.5 * (dl[1] + dl[2])), float(gl[0])
    if nonfinite != 0 or not (np.isfinite(d_loss) and np.isfinite(g_loss)):
        raise SystemExit(f"non-finite loss encountered (d_loss={d_loss}, g_loss={g_loss})")   # reference :487-488

    # time per kernel class over a few extra iterations (HIP events around every launch of the class; outside the timed
    # region because the extra events perturb it)
    classes = None
    if rank == 0:
        names = {_lib.TAG_GCONV_FWD: "gen_conv_fwd", _lib.TAG_GCONV_DGRAD: "gen_conv_dgrad", _lib.TAG_GCONV_WGRAD: "gen_conv_wgrad",
                 _lib.TAG_CRITIC_GEMM: "critic_gemm", _lib.TAG_ELEMENTWISE: "elementwise"}
    sync()
    nprof = 3
    eng.profile(sum(1 << t for t in (_lib.TAG_GCONV_FWD, _lib.TAG_GCONV_DGRAD, _lib.TAG_GCONV_WGRAD, _lib.TAG_CRITIC_GEMM,
                                     _lib.TAG_ELEMENTWISE)))
    for k in range(nprof):
        crit, gen = data[k % nbuf]
        trainer.iteration_raw(crit, gen)
    sync()
    if rank == 0:
        classes = {}
        for t, nm in names.items():
            ms, n = eng.profile_read(t)
            classes[nm] = {"ms_per_iteration": round(ms / nprof, 4), "launches_per_iteration": n // nprof}
    eng.profile(0)

    # per-launch table: HIP events around every GEMM launch of a few more iterations (outside the timed region)
    launches = None
    sync()
    eng.profile_launches(True)
    for k in range(nprof):
        crit, gen = data[k % nbuf]
        trainer.iteration_raw(crit, gen)
    sync()
    if rank == 0:
        rows = eng.launch_table()
        pk = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
        launches = []
        for r in sorted(rows, key=lambda r: -r["ms"]):
            n = max(r["launches"], 1)
            # the fp32-pipe GEMMs of the bf16 storage mode (Dense, the 64 -> 1 conv's weight gradient) are priced against the fp32 peak
            on_bf16_pipe = bf16 and ("bf16" in r["kernel"] or "ws16" in r["kernel"]) and "g9_wgrad" not in r["kernel"]
            peak_r = pk if on_bf16_pipe else FP32_MFMA_PEAK_TFLOPS
            tf = r["gflop"] / max(r["ms"], 1e-9)
            launches.append({"what": r["name"], "kind": r["kind"], "kernel": r["kernel"], "samples": r["batch"],
                             "launches_per_iteration": round(r["launches"] / nprof, 2), "gflop_per_launch": round(r["gflop"] / n, 3),
                             "ms_per_launch": round(r["ms"] / n, 4), "ms_per_iteration": round(r["ms"] / nprof, 4),
                             "tflops": round(tf, 1), "peak": peak_r, "frac": round(tf / peak_r, 4)})
    eng.profile_launches(False)

    if rank == 0:
        value = world * B * args.steps / dt
        avg_ms = kern_ms / max(kern_n, 1)
        # tap products per output position of the tagged launch (generator block 3 forward): direct 27, collapsed 8, or the
        # difference part of the shared-centre form 4 (fp32 default); the bf16 storage mode defaults to the collapsed form
        fast_fwd = opts.get("fast_fwd", "0" if bf16 else "1")
        taps = 27 if opts.get("collapse") == "0" else (8 if fast_fwd == "0" else 4)
        # "split3" (optional data point, never the metric): fp32 storage, the conv GEMMs' products on the bf16 matrix pipe from
        # operands split three ways in registers, six partial products per fp32 product -> priced against a sixth of the bf16 peak
        split3 = int(opts.get("split3", 0)) != 0 and not bf16
        peak = BF16_MFMA_PEAK_TFLOPS if bf16 else (BF16_MFMA_PEAK_TFLOPS / 6 if split3 else FP32_MFMA_PEAK_TFLOPS)
        achieved = gconv3_flops(B, ND, taps) / (avg_ms * 1e-3) / 1e12 if kern_n else None
        med = float(np.median(it_ms))
        direct_equiv = None
        if ND in DIRECT_GF_PER_SAMPLE:
            gf = n_critic * DIRECT_GF_PER_SAMPLE[ND]["critic_step"] + DIRECT_GF_PER_SAMPLE[ND]["gen_step"]
            direct_equiv = gf * 1e9 * B / (med * 1e-3) / 1e12          # per GPU
        it_tflops = flops_iter / (med * 1e-3) / 1e12
        traffic, traffic_src = dominant_traffic(bf16, ND, B) if taps == (8 if bf16 else 4) else (None, None)
        is_metric = (args.config, ND, B, n_critic, bf16, split3) == (2, 16, 256, 1, False, False)
        dtype = ("bf16 activations / gradients in HBM and bf16 MFMA operands, f32 accumulation, f32 master weights, "
                 "PixelNorm / softmax / penalty / Adam in f32") if bf16 else "f32"
        if split3:
            dtype = ("f32 storage and accumulation; forward / input-gradient conv GEMMs multiply f32 operands split into three bf16 "
                     "parts on the bf16 matrix pipe (6 partial products per product; weight gradients on the f32 pipe) -- "
                     "optional data point, not the BASELINE metric")
        out = {
            "metric": "cWGAN-GP train samples/sec, 24x16x16 tiles, bs=256" if is_metric
                      else f"cWGAN-GP train samples/sec, 24x{ND}x{ND} tiles, bs={B} per GPU (config {args.config})",
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": cfg["name"] if (ND, n_critic, bf16) == (cfg["nd"], cfg["n_critic"], bool(cfg["bf16"])) and args.batch is None and not split3
                                   else f"ndomain={ND}, 24h, bs={B} per GPU, {'bf16 storage' if bf16 else ('fp32, split3 GEMMs' if split3 else 'fp32')}, n_critic={n_critic}",
                       "batch_per_gpu": B, "global_batch": world * B, "n_critic": n_critic, "parallelism": f"dp{world}",
                       "world": world, "rccl_ranks_seen": ranks_seen, "backend": args.backend if world > 1 else None,
                       "exchange": None if world == 1 else
                                   ("per optimizer update: " + "; ".join(
                                       ("generator" if k == "g" else "critic") + " slab: "
                                       + ("one all-reduce of the flat gradient slab" if v == "allreduce" else
                                          "reduce-scatter, Adam on the owned 1/%d, all-gather of the updated weights" % world)
                                       for k, v in sorted(trainer.exchange.items()))
                                    + (", on a side stream beside the next generator forward" if trainer.overlap else ", on the compute stream")),
                       "exchange_by_slab": None if world == 1 else dict(trainer.exchange),
                       "grad_transport": None if world == 1 else trainer.grad_transport,
                       "rccl_verified": None if world == 1 else (args.backend == "nccl"),
                       "weights": "random init (RandomNormal 0.02 / glorot_uniform), dropout 0.25 active"},
            "iteration_ms": {"median": round(med, 4), "p10": round(float(np.percentile(it_ms, 10)), 4),
                             "p90": round(float(np.percentile(it_ms, 90)), 4), "n": int(args.steps),
                             "clock": "HIP events on the compute stream around each iteration"},
            "roofline": {"bound": "mfma", "kernel": ((("k_upconv_slab16<1, true, *> (last conv's tap products in its epilogue, their FLOPs not counted; critic steps do not store the output)"
                                                      if opts.get("g9_fused", "1") != "0" and opts.get("tapgather", "1") != "0" else "k_upconv_slab16<1>")
                                                     if (bf16 and taps == 8 and ND == 16 and opts.get("upconv_slab", "1") != "0") else
                                                     ("k_upconv_slab_t16<*, *> (8 x 8 tiles with their halo resident; last conv's tap products in its epilogue, "
                                                      "their FLOPs not counted)"
                                                      if (bf16 and taps == 8 and ND > 16 and ND % 16 == 0 and opts.get("upconv_slab_t", "1") != "0") else
                                                      "k_conv_gemm_ws<256, 64, 4, 1, %d, %s, 1>" % (8 if taps == 8 else 4, "true" if bf16 else "false"))))
                                                    + " (own symbol: this launch only), generator block 3 forward, "
                                                    + ("collapsed form (8 parity phases x 8 taps + bias + PixelNorm + LeakyReLU in the epilogue)"
                                                       if taps == 8 else
                                                       "difference part (E x U over 8 parity phases x 4 taps + shared part T + bias + "
                                                       "PixelNorm + LeakyReLU in the epilogue)"),
                         "achieved": None if achieved is None else round(achieved, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": None if achieved is None else round(achieved / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "launches": int(kern_n), "avg_launch_ms": round(avg_ms, 4),
                         "flops_per_launch": gconv3_flops(B, ND, taps),
                         "iteration": {"executed_gflop": round(flops_iter / 1e9, 2), "tflops": round(it_tflops, 2),
                                       "frac": round(it_tflops / peak, 4),
                                       "note": "algorithmic FLOPs of every GEMM of one iteration in the forms actually run "
                                               "(rdgan_flop_count) / median iteration time / the same MFMA peak; includes all "
                                               "elementwise kernels' time"},
                         "iteration_direct_equiv_tflops": None if direct_equiv is None else round(direct_equiv, 2),
                         "kernel_classes": classes,
                         "launches": launches,
                         "launches_note": "every GEMM launch of an iteration (HIP events on the launch stream, %d extra iterations outside "
                                          "the timed region): algorithmic GFLOP of the form run / mean duration / the dense MFMA peak of "
                                          "the launch's operand type; a split-K launch includes its finish kernel, a weight gradient "
                                          "its partial-slab fold" % nprof},
        }
        out.update({
            "final_losses": {"d_loss": round(d_loss, 5), "g_loss": round(g_loss, 5)},
        })
        if world == 1 and not args.no_cpu_baseline and ND == 16:
            # bounded sample: the default configuration's iteration at the GPU line's batch (~6 s per CPU iteration);
            # n_critic = 5 configurations at bs 64 so that the default run still ends within minutes
            out["cpu_baseline"] = cpu_baseline(ND, n_critic, batch=256 if n_critic == 1 else 64)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
