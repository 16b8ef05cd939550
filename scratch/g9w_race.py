"""Run-to-run differences of k_g9_wgrad_mfma through the op-level entry: which taps differ, at which sizes."""
import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pr_disagg_radar_gan_amd import _lib
lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for nd, B, bf16 in [(64, 8, 1), (64, 8, 0), (64, 2, 1), (64, 1, 1), (32, 16, 1), (16, 128, 1), (128, 2, 1), (64, 32, 1)]:
    g = torch.Generator(device="cuda"); g.manual_seed(nd * 100 + B)
    dl = torch.randn((B, 24, nd, nd), device="cuda", generator=g)
    h3 = torch.randn((B, 24, nd, nd, 64), device="cuda", generator=g)
    outs = []
    for k in range(6):
        dW = torch.full((27 * 64,), float("nan"), device="cuda")
        rc = lib.rdgan_op_g9_wgrad(p(dl), p(h3), p(dW), B, nd, bf16, 1, st)
        assert rc == 0, rc
        outs.append(dW.clone())
    ref = torch.full((27 * 64,), float("nan"), device="cuda")
    rc = lib.rdgan_op_g9_wgrad(p(dl), p(h3), p(ref), B, nd, bf16, 0, st) if nd <= 72 else -1
    taps = set()
    for o in outs[1:]:
        d = (o - outs[0]).abs().reshape(27, 64)
        taps |= set(torch.nonzero(d.max(dim=1).values > 0).flatten().tolist())
    err = float((outs[0] - ref).abs().max() / ref.abs().max()) if rc == 0 else float("nan")
    worst = max(float((o - ref).abs().max() / ref.abs().max()) for o in outs) if rc == 0 else float("nan")
    print(f"nd {nd} B {B} bf16 {bf16}: taps differing between runs {sorted(taps)}; rel err vs scalar kernel: first {err:.2e} worst {worst:.2e}", flush=True)
