#!/usr/bin/env python3
"""Benchmark of the cWGAN-GP training iteration on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one training iteration of the reference's loop body
(gan_train_cwgangp_pixelnorm.py:468-482) on synthetic 24x16x16 tiles: n_critic critic
updates + 1 generator update, fp32, batch 256 per GPU (BASELINE.json configs[1]: "ndomain=16,
24h, bs=256 fp32, 1 critic step + 1 gen step, single MI355X").  For N > 1 (one process per GPU,
launched by torch.distributed.run) the minibatch dimension is sharded: every rank processes its
own 256 samples and the gradient slabs are summed by RCCL (weak scaling).  Inputs are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NDOMAIN = 16
BATCH_PER_GPU = 256
N_CRITIC = 1
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
BF16_MFMA_PEAK_TFLOPS = 2516.6     # dense bf16 (v_mfma_f32_32x32x16_bf16, 32 cycles): 16x the fp32 rate; only with --opt mfma_bf16=1


def gconv3_flops(batch, nd=NDOMAIN, taps=4):
    """Algorithmic (executed) FLOPs of ONE launch of the dominant kernel: the difference-part GEMM of the generator's
    third UpSampling3D+Conv3D block (128 -> 64 channels onto the 24 x nd x nd grid) in the shared-centre form
    (DESIGN.md 4.2): 8 output-parity phases x 4 taps on the un-upsampled grid, 2 * B * (24*nd*nd) * 4*128 * 64
    = 402.7 MFLOP per sample at nd=16.  (The shared part T = S x of the same block is a separate, smaller launch.)"""
    return 2.0 * batch * 24 * nd * nd * taps * 128 * 64


# HBM traffic of ONE launch of the dominant kernel at the default configuration, from separate rocprofv3 --pmc passes
# (scripts/gpu_pmc_traffic.sh -> profiles/r01_e_pmc_hbm_traffic_gen_forward.json): FETCH_SIZE 241.6 MB x 2 (gfx950 counts
# half the bytes of wide streaming reads, MI355X_MICROARCH.md) + WRITE_SIZE 399.4 MB.  Algorithmic: E 109 MB + T 2 x 201 MB
# read, 403 MB output + 6 MB 1/l2 written.  Not measurable inside this process, hence a recorded constant.
DOMINANT_TRAFFIC_BYTES = 2 * 241.6e6 + 399.4e6

# FLOPs of one iteration (n_critic critic steps + 1 generator step) per sample in the reference's direct
# 27-tap form, nd = 16; used to price the measured iteration time as "direct-equivalent" TFLOP/s
DIRECT_GF_PER_SAMPLE = {"critic_step": 5.41, "gen_step": 13.54}


def cpu_baseline(iters=3, batch=32):
    """The oracle's torch-CPU restatement of the same iteration (kind "port": the reference's own
    runtime, TensorFlow 2.1, is not installed and no reference code travels to the GPU box),
    timed on the host cores on a bounded sample."""
    import torch
    from oracle import rdgan_torch as ot
    # the GPU box gives one GPU's job a 16-core share; oversubscribing it (torch defaults to every visible
    # core) makes the baseline 20x slower, so use at most 16 threads
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    tr = ot.Trainer(ndomain=NDOMAIN, seed=0)
    batches = []
    for i in range(N_CRITIC + 1):
        x, c, z = ot.synthetic_batch(batch, NDOMAIN, 50 + i)
        batches.append((torch.from_numpy(x), torch.from_numpy(c), torch.from_numpy(z)))

    def iteration(k):
        for j in range(N_CRITIC):
            x, c, z = batches[j]
            tr.critic_step(x, c, z, seed=1000 + k * 7 + j)
        x, c, z = batches[N_CRITIC]
        tr.gen_step(z, c, seed=2000 + k)

    iteration(0)                                   # warm-up
    t0 = time.perf_counter()
    for k in range(iters):
        iteration(k + 1)
    dt = time.perf_counter() - t0
    return {"value": round(batch * iters / dt, 3), "unit": "samples/s", "cores": int(cores), "kind": "port",
            "sample": f"{iters} iterations (n_critic={N_CRITIC} critic + 1 generator update) at bs={batch}, "
                      f"torch-CPU fp32 restatement of the reference arithmetic, after 1 warm-up iteration"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="samples per GPU per iteration")
    ap.add_argument("--n-critic", type=int, default=N_CRITIC)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ndomain", type=int, default=NDOMAIN, help="16 = BASELINE metric; 64 = large-domain variant (extra data point)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="rdgan_set_option override for A/B runs (e.g. --opt fast_bwd=0); the default run sets none")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pr_disagg_radar_gan_amd import Engine, weights as W, _lib
    from pr_disagg_radar_gan_amd.trainer import WGANGPTrainer, synthetic_batch_device
    import numpy as np

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        pg = dist.group.WORLD

    B = args.batch
    ND = args.ndomain
    eng = Engine(ndomain=ND, max_batch=B, device=dev)
    for kv in args.opt:
        name, value = kv.split("=")
        eng.set_option(name, int(value))
    rng = np.random.default_rng(0)                  # identical initial weights on every rank
    trainer = WGANGPTrainer(eng, W.init_generator(rng, ND), W.init_critic(rng, ND), n_disc=args.n_critic,
                            process_group=pg, world_size=world, rank=rank, base_seed=1234 + 1000 * 2)
    # synthetic inputs resident in HBM; per-rank seeds 1234 + 1000*config + rank
    nbuf = 4
    data = []
    for i in range(nbuf):
        crit = [synthetic_batch_device(B, ND, 1234 + 2000 + rank + 97 * (i * (args.n_critic + 1) + j), dev)
                for j in range(args.n_critic)]
        _, c, z = synthetic_batch_device(B, ND, 1234 + 2000 + rank + 97 * (i * (args.n_critic + 1) + args.n_critic) + 13, dev)
        data.append((crit, (z, c)))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    flags = []
    for k in range(args.warmup):
        crit, gen = data[k % nbuf]
        trainer.iteration(crit, gen)
    sync()
    eng.profile(1 << _lib.TAG_GCONV3_FWD)            # HIP events around the dominant kernel, on the launch stream
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        crit, gen = data[k % nbuf]
        d_loss, g_loss, bad = trainer.iteration(crit, gen)
        flags.append(bad)
    sync()
    dt = time.perf_counter() - t0
    kern_ms, kern_n = eng.profile_read(_lib.TAG_GCONV3_FWD)
    eng.profile(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    nonfinite = float(torch.stack(flags).max().item()) if flags else 0.0
    d_loss, g_loss = float(d_loss.item()), float(g_loss.item())
    if nonfinite != 0 or not (np.isfinite(d_loss) and np.isfinite(g_loss)):
        raise SystemExit(f"non-finite loss encountered (d_loss={d_loss}, g_loss={g_loss})")   # reference :487-488

    if rank == 0:
        value = world * B * args.steps / dt
        avg_ms = kern_ms / max(kern_n, 1)
        opts = dict(kv.split("=") for kv in args.opt)     # A/B runs: the tagged launch is the whole block in the other forms
        taps = 27 if opts.get("collapse") == "0" else (8 if opts.get("fast_fwd") == "0" else 4)
        bf16 = opts.get("mfma_bf16") == "1" and taps == 4     # mixed mode
        peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
        achieved = gconv3_flops(B, ND, taps) / (avg_ms * 1e-3) / 1e12 if kern_n else None
        direct_equiv = None
        if ND == 16:
            gf = args.n_critic * DIRECT_GF_PER_SAMPLE["critic_step"] + DIRECT_GF_PER_SAMPLE["gen_step"]
            direct_equiv = gf * 1e9 * B * args.steps / dt / 1e12          # per GPU
        out = {
            "metric": "cWGAN-GP train samples/sec, 24x16x16 tiles, bs=256" if (ND, B) == (16, 256)
                      else f"cWGAN-GP train samples/sec, 24x{ND}x{ND} tiles, bs={B} (extra data point)",
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16 MFMA operands in every heavy GEMM, f32 accumulation / tensors / optimizer (mixed mode)" if bf16 else "f32",
            "data": "synthetic",
            "config": {"workload": f"ndomain={ND}, 24h, bs={B} fp32 per GPU, {args.n_critic} critic step + 1 gen step"
                                   + (" (BASELINE configs[1])" if (ND, B, args.n_critic) == (16, 256, 1) else ""),
                       "global_batch": world * B, "n_critic": args.n_critic, "parallelism": f"dp{world}",
                       "weights": "random init (RandomNormal 0.02 / glorot_uniform), dropout 0.25 active"},
            "roofline": {"bound": "mfma", "kernel": "k_conv_gemm_ws<256, 64, 4, 1, 4, false, 1> (own symbol: this launch only), generator block 3 forward, "
                                                    "difference part (E x U over 8 parity phases x 4 taps + shared part T + bias + "
                                                    "PixelNorm + LeakyReLU in the epilogue)",
                         "achieved": None if achieved is None else round(achieved, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": None if achieved is None else round(achieved / peak, 4),
                         "traffic": DOMINANT_TRAFFIC_BYTES if (ND, B, taps, bf16) == (16, 256, 4, False) else None,
                         "traffic_source": "profiles/r01_e_pmc_hbm_traffic_gen_forward.json (separate --pmc passes, bytes per launch)",
                         "launches": int(kern_n), "avg_launch_ms": round(avg_ms, 4),
                         "flops_per_launch": gconv3_flops(B, ND, taps),
                         "iteration_direct_equiv_tflops": None if direct_equiv is None else round(direct_equiv, 2)},
            "final_losses": {"d_loss": round(d_loss, 5), "g_loss": round(g_loss, 5)},
        }
        if world == 1 and not args.no_cpu_baseline and ND == 16:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
