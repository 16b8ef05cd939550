"""Slab kernel of generator block 3 (k_upconv_slab16) against the streaming bf16 GEMM it replaces: h3, 1/l2 and the generator
output at several batches, and the time of the launch."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import Engine, weights as W
from oracle import rdgan_torch as ot
for B in (3, 256, 2048):
    eng = Engine(16, B)
    rng = np.random.default_rng(5)
    g = W.init_generator(rng, 16)
    g = [p if p.ndim > 1 else (0.05 * rng.standard_normal(p.shape)).astype(np.float32) for p in g]
    gs = eng.to_slab(g)
    x, cond, z = ot.synthetic_batch(min(B, 64), 16, 3)
    reps = (B + 63) // 64
    z = np.concatenate([z] * reps)[:B]; cond = np.concatenate([cond] * reps)[:B]
    zd, cd = torch.from_numpy(z).cuda(), torch.from_numpy(cond).cuda()
    eng.set_option("bf16", 1)
    res = {}
    for opt in (0, 1):
        eng.set_option("upconv_slab", opt)
        out = eng.gen_forward(gs, zd, cd).clone()
        h3 = eng.debug_activation(3, (B, 24, 16, 16, 64)).clone()
        eng.profile_launches(True)
        for _ in range(5):
            eng.gen_forward(gs, zd, cd)
        rows = [r for r in eng.launch_table() if "block3" in r["name"] and r["kind"] == "gemm"]
        eng.profile_launches(False)
        res[opt] = (out, h3, rows)
    o0, h0, r0 = res[0]; o1, h1, r1 = res[1]
    d = (h1 - h0).abs()
    rel = d / h0.abs().clamp_min(1e-3)
    print(f"B {B}: h3 max abs diff {float(d.max()):.3e} (max |h3| {float(h0.abs().max()):.3f}), elements differing by more than one bf16 ulp "
          f"(rel > 2^-7): {int((rel > 2**-7).sum())} of {h0.numel()}, identical {float((d == 0).float().mean()):.4f}; "
          f"output max abs diff {float((o1 - o0).abs().max()):.3e}; finite {bool(torch.isfinite(h1).all())}")
    for tag, rr in (("stream", r0), ("slab", r1)):
        for r in rr:
            print(f"   {tag}: {r['kernel']} {r['ms'] / r['launches']:.4f} ms  {r['gflop'] / r['ms']:.0f} TFLOP/s = {r['gflop'] / r['ms'] / 2516.6:.3f} of the bf16 roof")
    if B <= 3:
        ref = ot.generator_forward([torch.from_numpy(a).double() for a in g], torch.from_numpy(z).double(), torch.from_numpy(cond).double()).numpy()
        for tag, o in (("stream", o0), ("slab", o1)):
            print(f"   {tag}: forward vs fp64 oracle: {np.abs(o.cpu().numpy() - ref).max() / np.abs(ref).max():.3e}")
    eng.close()
