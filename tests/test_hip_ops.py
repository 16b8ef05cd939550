"""-m gpu: op-level parity of the HIP kernels (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import rdgan_torch as ot
from oracle import rng as orng
from tests.hip_util import dev, lib, ptr, rel_err, stream

pytestmark = pytest.mark.gpu

# (name, B, D,H,W (conv input before upsample), Cin, Cout, out dims, stride, pad, upsample)
GEOMS = [
    ("G1", 2, (3, 2, 2), 256, 256, (6, 4, 4), 1, (1, 1, 1), 1),
    ("G2", 2, (6, 4, 4), 256, 128, (12, 8, 8), 1, (1, 1, 1), 1),
    ("G3", 3, (12, 8, 8), 128, 64, (24, 16, 16), 1, (1, 1, 1), 1),
    ("D2", 5, (11, 7, 7), 64, 128, (6, 4, 4), 2, (1, 1, 1), 0),
    ("D3", 5, (6, 4, 4), 128, 256, (3, 2, 2), 2, (0, 0, 0), 0),
    ("D4", 7, (3, 2, 2), 256, 256, (2, 1, 1), 2, (1, 0, 0), 0),
    ("D2_nd64", 1, (11, 31, 31), 64, 128, (6, 16, 16), 2, (1, 1, 1), 0),
    ("plain_s1", 2, (5, 6, 7), 64, 64, (5, 6, 7), 1, (1, 1, 1), 0),
    ("plain_128_64", 2, (5, 6, 7), 128, 64, (5, 6, 7), 1, (1, 1, 1), 0),
]


def _oracle_conv(x, w, b, g):
    _, B, dims, cin, cout, od, stride, pad, up = g
    xt = torch.from_numpy(x).double()
    if up:
        xt = ot.upsample3d(xt)
    return ot._conv3d_tf(xt, torch.from_numpy(w).double(), None if b is None else torch.from_numpy(b).double(),
                         stride, pad, od)


@pytest.mark.parametrize("g", GEOMS, ids=[g[0] for g in GEOMS])
def test_conv3d_forward(g):
    name, B, dims, cin, cout, od, stride, pad, up = g
    rng = np.random.default_rng(hash(name) % 1000)
    x = rng.standard_normal((B,) + dims + (cin,)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = _oracle_conv(x, w, b, g).numpy()
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = torch.full((B,) + od + (cout,), float("nan"), device="cuda")
    rc = lib().rdgan_op_conv3d(ptr(xd), ptr(wd), ptr(bd), ptr(y), B, *dims, cin, cout, *od, stride, *pad, up, stream())
    assert rc == 0
    assert rel_err(y.cpu().numpy(), ref) < 1e-5


def _bf16_round(a):
    """round to nearest-even bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 does)"""
    return torch.from_numpy(a).bfloat16().float().numpy()


@pytest.mark.parametrize("g", [g for g in GEOMS if not g[8]], ids=[g[0] for g in GEOMS if not g[8]])
def test_conv3d_bf16_operands(g):
    """bf16-operand conv GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulation): against the oracle evaluated on the
    bf16-rounded inputs the result must be as tight as the fp32 kernel's (only the summation order differs)."""
    name, B, dims, cin, cout, od, stride, pad, up = g
    rng = np.random.default_rng(hash(name) % 1000 + 7)
    x = rng.standard_normal((B,) + dims + (cin,)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = _oracle_conv(_bf16_round(x), _bf16_round(w), b, g).numpy()
    ref32 = _oracle_conv(x, w, b, g).numpy()
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = torch.full((B,) + od + (cout,), float("nan"), device="cuda")
    rc = lib().rdgan_op_conv3d_bf16(ptr(xd), ptr(wd), ptr(bd), ptr(y), B, *dims, cin, cout, *od, stride, *pad, 0, stream())
    assert rc == 0
    assert rel_err(y.cpu().numpy(), ref) < 1e-5
    assert 1e-4 < rel_err(y.cpu().numpy(), ref32) < 3e-2          # and it really is bf16: ~2^-9 per operand
    # bf16 destination (the storage mode's epilogue): the same fp32 accumulator, rounded to nearest-even bf16 once
    y16 = torch.full((B,) + od + (cout,), float("nan"), device="cuda", dtype=torch.bfloat16)
    rc = lib().rdgan_op_conv3d_bf16(ptr(xd), ptr(wd), ptr(bd), ptr(y16), B, *dims, cin, cout, *od, stride, *pad, 1, stream())
    assert rc == 0
    assert torch.equal(y16, y.bfloat16())
    # the same launch through the fragment kernel (k_conv_gemm_f16: weights global -> VGPR in fragment order, 256 x 128 tiles,
    # epilogue in registers): same chunk and k order, so the same bits
    if cout % 128 == 0:
        y16f = torch.full((B,) + od + (cout,), float("nan"), device="cuda", dtype=torch.bfloat16)
        rc = lib().rdgan_op_conv3d_bf16(ptr(xd), ptr(wd), ptr(bd), ptr(y16f), B, *dims, cin, cout, *od, stride, *pad, 2, stream())
        assert rc == 0
        assert torch.equal(y16f, y16)


@pytest.mark.parametrize("g", GEOMS, ids=[g[0] for g in GEOMS])
def test_conv3d_dgrad(g):
    name, B, dims, cin, cout, od, stride, pad, up = g
    rng = np.random.default_rng(hash(name) % 1000 + 1)
    in_dims = tuple(2 * d for d in dims) if up else dims       # the conv's own input grid
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cout)).astype(np.float32)
    gy = rng.standard_normal((B,) + od + (cout,)).astype(np.float32)
    xt = torch.zeros((B,) + in_dims + (cin,), dtype=torch.float64, requires_grad=True)
    yt = ot._conv3d_tf(xt, torch.from_numpy(w).double(), None, stride, pad, od)
    ref, = torch.autograd.grad(yt, xt, torch.from_numpy(gy).double())
    gx = torch.full((B,) + in_dims + (cin,), float("nan"), device="cuda")
    gyd, wd = dev(gy), dev(w)            # keep the device tensors alive across the call
    rc = lib().rdgan_op_conv3d_dgrad(ptr(gyd), ptr(wd), ptr(gx), B, *in_dims, cin, cout, *od, stride, *pad, stream())
    assert rc == 0
    assert rel_err(gx.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("g", [g for g in GEOMS if not g[8]], ids=[g[0] for g in GEOMS if not g[8]])
def test_conv3d_dgrad_bf16_operands(g):
    """input gradient through the bf16-operand GEMM (parity-phase plans, weights un-transposed): exact against the oracle
    on bf16-rounded gy and w up to fp32 summation order"""
    name, B, dims, cin, cout, od, stride, pad, up = g
    rng = np.random.default_rng(hash(name) % 1000 + 9)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cout)).astype(np.float32)
    gy = rng.standard_normal((B,) + od + (cout,)).astype(np.float32)
    xt = torch.zeros((B,) + dims + (cin,), dtype=torch.float64, requires_grad=True)
    yt = ot._conv3d_tf(xt, torch.from_numpy(_bf16_round(w)).double(), None, stride, pad, od)
    ref, = torch.autograd.grad(yt, xt, torch.from_numpy(_bf16_round(gy)).double())
    gx = torch.full((B,) + dims + (cin,), float("nan"), device="cuda")
    gyd, wd = dev(gy), dev(w)
    rc = lib().rdgan_op_conv3d_dgrad_bf16(ptr(gyd), ptr(wd), ptr(gx), B, *dims, cin, cout, *od, stride, *pad, stream())
    assert rc == 0
    assert rel_err(gx.cpu().numpy(), ref.numpy()) < 1e-5


_WG16 = [g for g in GEOMS if not g[8] and (g[3] % 128 == 0 or (g[3] == 64 and g[4] % 128 == 0))]   # tiles of 128+ rows
# >= 32768 gathered rows and N % 128 == 0: the 256 x 128 tiles on three LDS stages (k_wgrad_gemm_ws16<256, 128>, "wgrad_wide"):
# 64 input channels (four taps per row tile, 27 -> 28 tap slots), 128 (two per tile, a partial last split), 256 (one tile per tap)
_WG16 += [("D2_nd64_wide", 22, (11, 31, 31), 64, 128, (6, 16, 16), 2, (1, 1, 1), 0),
          ("D3_nd64_wide", 171, (6, 16, 16), 128, 256, (3, 8, 8), 2, (0, 0, 0), 0),
          ("plain_256_wide", 157, (5, 6, 7), 256, 128, (5, 6, 7), 1, (1, 1, 1), 0)]


@pytest.mark.parametrize("g", _WG16, ids=[g[0] for g in _WG16])
def test_conv3d_wgrad_bf16_operands(g):
    """weight gradient through the bf16-operand kernel (transposed LDS reads, ds_read_b64_tr_b16): exact against the oracle
    on bf16-rounded x and gy up to fp32 summation order"""
    name, B, dims, cin, cout, od, stride, pad, up = g
    rng = np.random.default_rng(hash(name) % 1000 + 11)
    x = rng.standard_normal((B,) + dims + (cin,)).astype(np.float32)
    gy = rng.standard_normal((B,) + od + (cout,)).astype(np.float32)
    wt = torch.zeros((3, 3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    yt = ot._conv3d_tf(torch.from_numpy(_bf16_round(x)).double(), wt, None, stride, pad, od)
    ref, = torch.autograd.grad(yt, wt, torch.from_numpy(_bf16_round(gy)).double())
    dw = torch.full((3, 3, 3, cin, cout), float("nan"), device="cuda")
    xd, gyd = dev(x), dev(gy)
    rc = lib().rdgan_op_conv3d_wgrad_bf16(ptr(xd), ptr(gyd), ptr(dw), B, *dims, cin, cout, *od, stride, *pad, stream())
    assert rc == 0
    assert rel_err(dw.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("g", GEOMS, ids=[g[0] for g in GEOMS])
def test_conv3d_wgrad(g):
    name, B, dims, cin, cout, od, stride, pad, up = g
    rng = np.random.default_rng(hash(name) % 1000 + 2)
    x = rng.standard_normal((B,) + dims + (cin,)).astype(np.float32)
    gy = rng.standard_normal((B,) + od + (cout,)).astype(np.float32)
    wt = torch.zeros((3, 3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    xt = torch.from_numpy(x).double()
    if up:
        xt = ot.upsample3d(xt)
    yt = ot._conv3d_tf(xt, wt, None, stride, pad, od)
    ref, = torch.autograd.grad(yt, wt, torch.from_numpy(gy).double())
    dw = torch.full((3, 3, 3, cin, cout), float("nan"), device="cuda")
    xd, gyd = dev(x), dev(gy)
    rc = lib().rdgan_op_conv3d_wgrad(ptr(xd), ptr(gyd), ptr(dw), B, *dims, cin, cout, *od, stride, *pad, up, stream())
    assert rc == 0
    assert rel_err(dw.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("C", [64, 128, 256])
def test_pixelnorm_lrelu_fwd_bwd(C):
    rng = np.random.default_rng(C)
    npix = 1001
    y = rng.standard_normal((npix, C)).astype(np.float32)
    y[7] = 0.0                                           # KAT: zero vector -> 0, no NaN
    gh = rng.standard_normal((npix, C)).astype(np.float32)
    yt = torch.from_numpy(y).double().requires_grad_(True)
    ht = torch.nn.functional.leaky_relu(ot.pixel_norm(yt), 0.2)
    ref_dy, = torch.autograd.grad(ht, yt, torch.from_numpy(gh).double())
    yd = dev(y); h = torch.empty_like(yd); rinv = torch.empty(npix, device="cuda")
    assert lib().rdgan_op_pixelnorm_lrelu(ptr(yd), ptr(h), ptr(rinv), npix, C, stream()) == 0
    hn = h.cpu().numpy()
    assert np.all(np.isfinite(hn)) and np.all(hn[7] == 0)
    assert rel_err(hn, ht.detach().numpy()) < 1e-6
    dy = torch.empty_like(yd)
    ghd = dev(gh)
    assert lib().rdgan_op_pixelnorm_lrelu_bwd(ptr(ghd), ptr(h), ptr(rinv), ptr(dy), npix, C, stream()) == 0
    mask = np.ones(npix, bool); mask[7] = False          # the zero row has an eps-dominated (1e4) slope
    assert rel_err(dy.cpu().numpy()[mask], ref_dy.numpy()[mask]) < 2e-5


def test_rng_bit_exact():
    n = 100003
    for seed, sid in ((1, 1), (0x1234567890ABCDEF, 3), (2 ** 63 + 5, 5)):
        m = torch.empty(n, device="cuda"); u = torch.empty(n, device="cuda")
        assert lib().rdgan_op_rng(seed, sid, ptr(m), ptr(u), n, stream()) == 0
        assert np.array_equal(m.cpu().numpy(), orng.dropout_scale_mask(seed, sid, (n,)))
        assert np.array_equal(u.cpu().numpy(), orng.uniform(seed, sid, n))


@pytest.mark.parametrize("bf16", [0, 1])
@pytest.mark.parametrize("g", [0, 1, 2])
def test_shared_centre_wgrad_256_row_tile(g, bf16):
    """The weight-gradient kernels that carry the bs = 256 step -- k_wgrad_gemm_ws<256, 64> and its bf16 twin
    k_wgrad_gemm_ws16<256, 64> -- through the production plan of generator block 3 (shared-centre form: 4 parity phases x
    4 taps; the launcher takes the 256-row tile from B * D*H*W >= 65536, i.e. B >= 86), against the definition
        dU[g][(ph,th),(pw,tw)] = sum_{b,s,h,w} src[b, s + (g==2), h+ph-1+th, w+pw-1+tw, :]^T  dy[b, od(s), 2h+ph, 2w+pw, :]
    evaluated in float64 (on the bf16-rounded operands for the bf16 kernel): 1e-5."""
    B, D, H, W, Cin, Cout = 86, 12, 8, 8, 128, 64
    rng = np.random.default_rng(40 + g)
    SD = D if g == 1 else D + 1
    DD = D if g == 1 else 2 * D
    src = rng.standard_normal((B, SD, H, W, Cin)).astype(np.float32)
    dy = rng.standard_normal((B, DD, 2 * H, 2 * W, Cout)).astype(np.float32)
    s64 = (_bf16_round(src) if bf16 else src).astype(np.float64)
    d64 = (_bf16_round(dy) if bf16 else dy).astype(np.float64)
    sp = np.zeros((B, SD, H + 2, W + 2, Cin))                       # zero halo of one pixel in h and w
    sp[:, :, 1:-1, 1:-1] = s64
    td = 1 if g == 2 else 0
    ref = np.zeros((16, Cin, Cout))
    for ph in range(2):
        for pw in range(2):
            dsel = d64[:, (slice(None) if g == 1 else slice(1 if g == 2 else 0, None, 2)), ph::2, pw::2]   # [B, D, H, W, Cout]
            for th in range(2):
                for tw in range(2):
                    a = sp[:, td:td + D, ph + th:ph + th + H, pw + tw:pw + tw + W]                          # offsets ph-1+th (+1 halo)
                    ref[(ph * 2 + th) * 4 + (pw * 2 + tw)] = np.tensordot(a.reshape(-1, Cin), dsel.reshape(-1, Cout), axes=(0, 0))
    dU = torch.full((48, Cin, Cout), float("nan"), device="cuda")
    sd, dd = dev(src), dev(dy)                     # (named: the device copies must outlive the call)
    rc = lib().rdgan_op_fastd_wgrad(ptr(sd), ptr(dd), ptr(dU), B, D, H, W, Cin, Cout, g, bf16, stream())
    assert rc == 0
    got = dU[g * 16:(g + 1) * 16].cpu().numpy()
    assert rel_err(got, ref) < 1e-5


@pytest.mark.parametrize("kernel", [1, 0])
@pytest.mark.parametrize("nd,B,bf16", [(16, 3, 0), (16, 3, 1), (8, 5, 0), (8, 3, 1), (24, 1, 0), (32, 1, 0), (64, 1, 1), (128, 1, 0), (16, 40, 0)])
def test_last_conv_weight_gradient_kernels(nd, B, bf16, kernel):
    """Backward of the 64 -> 1 conv (T:345) w.r.t. its kernel, through the production kernels alone: the matrix-pipe kernel
    k_g9_wgrad_mfma (kernel = 1: 128-pixel tiles of whole w rows -- 16, 8, 2 rows at ndomain 8, 16, 64; tiles that straddle
    hour planes and samples at ndomain 8; persistent workgroups that walk several tiles at B = 40) and the scalar kernel it replaced (0), against the definition
        dW[kd,kh,kw,c] = sum_{b,d,h,w} dl[b, d+1-kd, h+1-kh, w+1-kw] * h3[b, d, h, w, c]
    evaluated in float64 (on the bf16-rounded h3 for the storage mode): 1e-5."""
    rng = np.random.default_rng(70 + nd + B)
    D = 24
    dl = rng.standard_normal((B, D, nd, nd)).astype(np.float32)
    h3 = rng.standard_normal((B, D, nd, nd, 64)).astype(np.float32)
    h64 = (_bf16_round(h3) if bf16 else h3).astype(np.float64)
    dlp = np.zeros((B, D + 2, nd + 2, nd + 2))
    dlp[:, 1:-1, 1:-1, 1:-1] = dl
    ref = np.zeros((3, 3, 3, 64))
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                sh = dlp[:, 2 - kd:2 - kd + D, 2 - kh:2 - kh + nd, 2 - kw:2 - kw + nd]       # dl[d+1-kd, ...] with the +1 halo
                ref[kd, kh, kw] = np.tensordot(sh.reshape(-1), h64.reshape(-1, 64), axes=(0, 0))
    d_dl, d_h3 = dev(dl), dev(h3)
    dW = torch.full((27 * 64,), float("nan"), device="cuda")
    rc = lib().rdgan_op_g9_wgrad(ptr(d_dl), ptr(d_h3), ptr(dW), B, nd, bf16, kernel, stream())
    if (kernel == 1 and nd & (nd - 1)) or (kernel == 0 and nd > 72):
        assert rc == -2          # the matrix-pipe kernel takes power-of-two widths (the engine falls back to the scalar kernel,
        return                   # whose four dlogits planes must fit in LDS: ndomain <= 72)
    assert rc == 0
    assert rel_err(dW.cpu().numpy().reshape(3, 3, 3, 64), ref) < 1e-5


@pytest.mark.parametrize("B", [1, 3])
def test_upconv_slab_kernel_vs_oracle(B):
    """k_upconv_slab16 alone (rdgan_op_upconv_slab16): generator block 3 of the bf16 storage mode -- UpSampling3D(2) + Conv3D(128 ->
    64, 3x3x3, 'same') + bias + PixelNorm + LeakyReLU(0.2) (T:340-343) on a 12 x 8 x 8 x 128 input -- against the fp64 oracle on the
    bf16-rounded input.  The kernel rounds the COLLAPSED weights (sums of up to 8 taps) to bf16 and its output to bf16: 2^-8 each,
    so 2e-2 of the largest output (observed 6e-3); the per-pixel 1/l2 it hands to the backward pass at 1e-2."""
    g = torch.Generator(); g.manual_seed(100 + B)
    x = torch.randn((B, 12, 8, 8, 128), generator=g)
    w = 0.02 * torch.randn((3, 3, 3, 128, 64), generator=g)
    bias = 0.05 * torch.randn((64,), generator=g)
    u = ot.upsample3d(x.to(torch.bfloat16).double())
    pre = ot._conv3d_tf(u, w.double(), bias.double(), 1, (1, 1, 1), u.shape[1:4])
    ref = ot._lrelu(ot.pixel_norm(pre)).numpy()
    rinv_ref = (1.0 / torch.sqrt((pre * pre).mean(-1) + 1e-8)).numpy()
    xd, wd, bd = dev(x.numpy()), dev(w.numpy()), dev(bias.numpy())
    y = torch.full((B, 24, 16, 16, 64), float("nan"), device="cuda")
    rinv = torch.full((B, 24, 16, 16), float("nan"), device="cuda")
    rc = lib().rdgan_op_upconv_slab16(ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(rinv), ptr(None), B, stream())
    assert rc == 0
    assert rel_err(y.cpu().numpy(), ref) < 2e-2
    assert rel_err(rinv.cpu().numpy(), rinv_ref) < 1e-2
    # one-hot probe: a single kernel tap and channel pair must move exactly the source voxel the definition names into the output
    x1 = torch.zeros((1, 12, 8, 8, 128)); x1[..., 77] = (torch.arange(12 * 64, dtype=torch.float32).reshape(1, 12, 8, 8) % 251) + 1
    for tap in (0, 13, 26, 5):
        w1 = torch.zeros((3, 3, 3, 128, 64)); w1[tap // 9, (tap // 3) % 3, tap % 3, 77, 5] = 1.0
        u1 = ot.upsample3d(x1.double())
        want = ot._conv3d_tf(u1, w1.double(), torch.zeros(64).double(), 1, (1, 1, 1), u1.shape[1:4])[..., 5].numpy()
        xd, wd, bd = dev(x1.numpy()), dev(w1.numpy()), dev(np.zeros(64, np.float32))
        dbg = torch.zeros((24 * 256, 4), device="cuda")
        y1 = torch.empty((1, 24, 16, 16, 64), device="cuda"); r1 = torch.empty((1, 24, 16, 16), device="cuda")
        assert lib().rdgan_op_upconv_slab16(ptr(xd), ptr(wd), ptr(bd), ptr(y1), ptr(r1), ptr(dbg), 1, stream()) == 0
        got = dbg.cpu().numpy().reshape(1, 24, 16, 16, 4)
        np.testing.assert_array_equal(got[..., 0], got[..., 1])                  # both lane halves of a row hold the same sum
        np.testing.assert_allclose(np.sqrt(got[..., 0]), want, rtol=0, atol=1e-3), tap


@pytest.mark.parametrize("B,H,W", [(2, 16, 16), (1, 8, 24), (3, 24, 8)])
def test_upconv_slab_tiled_kernel_vs_oracle(B, H, W):
    """k_upconv_slab_t16 alone (rdgan_op_upconv_slab_t16): generator block 3 of the bf16 storage mode on source planes larger than
    8 x 8 (the large-domain variant, L:355-358: 32 x 32 at ndomain 64; here 16 x 16 and two rectangular grids that would expose a
    swapped axis) -- (h, w) tiles of 8 x 8 positions with their halo resident, the K loop in two channel halves -- against the fp64
    oracle on the bf16-rounded input: 2e-2 of the largest output as for the one-tile kernel (observed 6e-3), 1/l2 at 1e-2.  One-hot
    probes: a single kernel tap and channel pair must move exactly the source voxel the definition names, at EVERY position --
    tile borders (halo columns fetched from the neighbouring tile's positions) and picture borders (zeros) included."""
    g = torch.Generator(); g.manual_seed(300 + B + H)
    x = torch.randn((B, 12, H, W, 128), generator=g)
    w = 0.02 * torch.randn((3, 3, 3, 128, 64), generator=g)
    bias = 0.05 * torch.randn((64,), generator=g)
    u = ot.upsample3d(x.to(torch.bfloat16).double())
    pre = ot._conv3d_tf(u, w.double(), bias.double(), 1, (1, 1, 1), u.shape[1:4])
    ref = ot._lrelu(ot.pixel_norm(pre)).numpy()
    rinv_ref = (1.0 / torch.sqrt((pre * pre).mean(-1) + 1e-8)).numpy()
    xd, wd, bd = dev(x.numpy()), dev(w.numpy()), dev(bias.numpy())
    y = torch.full((B, 24, 2 * H, 2 * W, 64), float("nan"), device="cuda")
    rinv = torch.full((B, 24, 2 * H, 2 * W), float("nan"), device="cuda")
    rc = lib().rdgan_op_upconv_slab_t16(ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(rinv), ptr(None), B, H, W, stream())
    assert rc == 0
    assert rel_err(y.cpu().numpy(), ref) < 2e-2
    assert rel_err(rinv.cpu().numpy(), rinv_ref) < 1e-2
    x1 = torch.zeros((1, 12, H, W, 128)); x1[..., 77] = (torch.arange(12 * H * W, dtype=torch.float32).reshape(1, 12, H, W) % 251) + 1
    for tap, ch in ((0, 77), (13, 77), (26, 77), (5, 77), (21, 3), (9, 120)):      # (channels of both K halves)
        xs = torch.zeros_like(x1); xs[..., ch] = x1[..., 77]
        w1 = torch.zeros((3, 3, 3, 128, 64)); w1[tap // 9, (tap // 3) % 3, tap % 3, ch, 5] = 1.0
        u1 = ot.upsample3d(xs.double())
        want = ot._conv3d_tf(u1, w1.double(), torch.zeros(64).double(), 1, (1, 1, 1), u1.shape[1:4])[..., 5].numpy()
        xd, wd, bd = dev(xs.numpy()), dev(w1.numpy()), dev(np.zeros(64, np.float32))
        dbg = torch.zeros((24 * 4 * H * W, 4), device="cuda")
        y1 = torch.empty((1, 24, 2 * H, 2 * W, 64), device="cuda"); r1 = torch.empty((1, 24, 2 * H, 2 * W), device="cuda")
        assert lib().rdgan_op_upconv_slab_t16(ptr(xd), ptr(wd), ptr(bd), ptr(y1), ptr(r1), ptr(dbg), 1, H, W, stream()) == 0
        got = dbg.cpu().numpy().reshape(1, 24, 2 * H, 2 * W, 4)
        np.testing.assert_array_equal(got[..., 0], got[..., 1])                  # both lane halves of a row hold the same sum
        np.testing.assert_allclose(np.sqrt(got[..., 0]), want, rtol=0, atol=1e-3), (tap, ch)
    assert lib().rdgan_op_upconv_slab_t16(ptr(xd), ptr(wd), ptr(bd), ptr(y1), ptr(r1), ptr(None), 1, 12, 8, stream()) == -2    # H % 8


@pytest.mark.parametrize("B", [1, 2, 3, 7, 130, 1100])
def test_d2_dgrad_slab_kernel_vs_oracle(B):
    """k_d2_dgrad_slab16 alone (rdgan_op_d2_dgrad_slab16): input gradient of the critic's second layer (backward of T:291,
    Conv3D(128, 3x3x3, stride 2, 'same') on 11 x 7 x 7 x 64 -> 6 x 4 x 4 x 128) times the LeakyReLU'/dropout gate of layer 1
    (T:287-289), against the definition (every tap scattered, oracle/rdgan_np.py conv3d_input_grad) in fp64 on the bf16-rounded
    operands.  The kernel accumulates in fp32 and rounds its output to bf16 once: 2^-8 of each element (checked per element), odd
    B = a last work item of one sample, B = 1100 = persistent workgroups walk two items.  With dropout the kernel reads the mask
    from the stored activation (+0.0 = dropped, rd_drop_apply / rd_gate_from_out), a kept exact zero is stored as -0.0."""
    from oracle import rdgan_np as onp
    rng = np.random.default_rng(200 + B)
    gy = rng.standard_normal((B, 6, 4, 4, 128)).astype(np.float32)
    w = (0.05 * rng.standard_normal((3, 3, 3, 64, 128))).astype(np.float32)
    aux = rng.standard_normal((B, 11, 7, 7, 64)).astype(np.float32)
    gx_ref = onp.conv3d_input_grad(_bf16_round(gy).astype(np.float64), _bf16_round(w).astype(np.float64), (11, 7, 7), 2, (1, 1, 1))
    slope = np.where(_bf16_round(aux) > 0, 1.0, 0.2)
    gyd, wd = dev(gy), dev(w)
    for seed in (0, 0x5DEECE66D):
        want, a = gx_ref * slope, aux.copy()
        if seed:
            m = orng.dropout_scale_mask(seed, orng.STREAM_D1, want.shape)
            a[m == 0] = 0.0                                   # dropped: +0.0
            kept0 = (m > 0) & (rng.random(want.shape) < 0.01)
            a[kept0] = -0.0                                   # kept, exactly zero: LeakyReLU' = alpha there (TF: features > 0 ? 1 : alpha)
            want = gx_ref * np.where(_bf16_round(a) > 0, 1.0, 0.2) * m
        gx = torch.full((B, 11, 7, 7, 64), float("nan"), device="cuda")
        ad = dev(a)
        assert lib().rdgan_op_d2_dgrad_slab16(ptr(gyd), ptr(wd), ptr(ad), ptr(gx), B, int(seed != 0), stream()) == 0
        got = gx.cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(got))
        np.testing.assert_allclose(got, want, rtol=2.0 ** -8 + 1e-5, atol=1e-5 * np.abs(want).max())
    # one-hot probe: one kernel tap and channel pair moves exactly the voxels the definition names
    gy1 = np.zeros((1, 6, 4, 4, 128), np.float32); gy1[..., 77] = (np.arange(96).reshape(1, 6, 4, 4) % 61) + 1
    a1 = np.ones((1, 11, 7, 7, 64), np.float32)
    for tap in (0, 13, 26, 5, 21):
        w1 = np.zeros((3, 3, 3, 64, 128), np.float32); w1[tap // 9, (tap // 3) % 3, tap % 3, 9, 77] = 1.0
        want = onp.conv3d_input_grad(gy1.astype(np.float64), w1.astype(np.float64), (11, 7, 7), 2, (1, 1, 1))
        gx = torch.full((1, 11, 7, 7, 64), float("nan"), device="cuda")
        g1d, w1d, a1d = dev(gy1), dev(w1), dev(a1)
        assert lib().rdgan_op_d2_dgrad_slab16(ptr(g1d), ptr(w1d), ptr(a1d), ptr(gx), 1, 0, stream()) == 0
        np.testing.assert_array_equal(gx.cpu().numpy(), want.astype(np.float32)), tap


@pytest.mark.parametrize("B,OH,OW", [(1, 8, 8), (3, 4, 12), (2, 16, 16), (2, 12, 4)])
def test_d2_dgrad_slab_tiled_kernel_vs_oracle(B, OH, OW):
    """k_d2_dgrad_slab_t16 alone (rdgan_op_d2_dgrad_slab_t16): the same input gradient on the larger domains (L:291-293: 6 x 16 x 16 x
    128 -> 11 x 31 x 31 x 64 at ndomain 64), tiles of 8 x 8 destination positions, against the definition in fp64 on the bf16-rounded
    operands; every destination written (the buffer starts as NaN), 2^-8 per element; rectangular grids expose a swapped axis, odd B a
    last item of one sample; with and without dropout; one-hot probes of five taps at every position (tile and picture borders)."""
    from oracle import rdgan_np as onp
    rng = np.random.default_rng(400 + B + OH)
    IH, IW = 2 * OH - 1, 2 * OW - 1
    gy = rng.standard_normal((B, 6, OH, OW, 128)).astype(np.float32)
    w = (0.05 * rng.standard_normal((3, 3, 3, 64, 128))).astype(np.float32)
    aux = rng.standard_normal((B, 11, IH, IW, 64)).astype(np.float32)
    gx_ref = onp.conv3d_input_grad(_bf16_round(gy).astype(np.float64), _bf16_round(w).astype(np.float64), (11, IH, IW), 2, (1, 1, 1))
    slope = np.where(_bf16_round(aux) > 0, 1.0, 0.2)
    gyd, wd = dev(gy), dev(w)
    for seed in (0, 0x5DEECE66D):
        want, a = gx_ref * slope, aux.copy()
        if seed:
            m = orng.dropout_scale_mask(seed, orng.STREAM_D1, want.shape)
            a[m == 0] = 0.0
            want = gx_ref * np.where(_bf16_round(a) > 0, 1.0, 0.2) * m
        gx = torch.full((B, 11, IH, IW, 64), float("nan"), device="cuda")
        ad = dev(a)
        assert lib().rdgan_op_d2_dgrad_slab_t16(ptr(gyd), ptr(wd), ptr(ad), ptr(gx), B, OH, OW, int(seed != 0), stream()) == 0
        got = gx.cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(got))
        np.testing.assert_allclose(got, want, rtol=2.0 ** -8 + 1e-5, atol=1e-5 * np.abs(want).max())
    gy1 = np.zeros((1, 6, OH, OW, 128), np.float32); gy1[..., 77] = (np.arange(6 * OH * OW).reshape(1, 6, OH, OW) % 61) + 1
    a1 = np.ones((1, 11, IH, IW, 64), np.float32)
    for tap in (0, 13, 26, 5, 21):
        w1 = np.zeros((3, 3, 3, 64, 128), np.float32); w1[tap // 9, (tap // 3) % 3, tap % 3, 9, 77] = 1.0
        want = onp.conv3d_input_grad(gy1.astype(np.float64), w1.astype(np.float64), (11, IH, IW), 2, (1, 1, 1))
        gx = torch.full((1, 11, IH, IW, 64), float("nan"), device="cuda")
        g1d, w1d, a1d = dev(gy1), dev(w1), dev(a1)
        assert lib().rdgan_op_d2_dgrad_slab_t16(ptr(g1d), ptr(w1d), ptr(a1d), ptr(gx), 1, OH, OW, 0, stream()) == 0
        np.testing.assert_array_equal(gx.cpu().numpy(), want.astype(np.float32)), tap
    assert lib().rdgan_op_d2_dgrad_slab_t16(ptr(g1d), ptr(w1d), ptr(a1d), ptr(gx), 1, 6, 8, 0, stream()) == -2       # OH % 4


@pytest.mark.parametrize("B", [1, 3, 12])
def test_upconv_wgrad_slab_kernel_vs_definition(B):
    """k_upconv_wgrad_slab16 alone (rdgan_op_upconv_wgrad_slab16): the collapsed weight gradient of generator block 3 (backward of
    T:340-341) -- entry (phase, tap) = sum over samples and source positions r of x[r + phase - 1 + tap] (outer) dy[2 r + phase],
    zero outside the picture -- against the definition in fp64 on the bf16-rounded operands.  bf16 products are exact in fp32 and
    the kernel accumulates in fp32: 2e-5 of the largest entry.  B = 12: 32 groups of workgroups (two stages in flight), B = 1, 3:
    8 groups, some without an item."""
    rng = np.random.default_rng(300 + B)
    x = rng.standard_normal((B, 12, 8, 8, 128)).astype(np.float32)
    dy = rng.standard_normal((B, 24, 16, 16, 64)).astype(np.float32)
    xr, dyr = _bf16_round(x).astype(np.float64), _bf16_round(dy).astype(np.float64)
    xp = np.zeros((B, 14, 10, 10, 128)); xp[:, 1:13, 1:9, 1:9] = xr
    ref = np.zeros((64, 128, 64))
    for p in range(8):
        par = (p >> 2, (p >> 1) & 1, p & 1)
        dyp = dyr[:, par[0]::2, par[1]::2, par[2]::2]
        for t in range(8):
            o = [par[a] - 1 + ((t >> (2 - a)) & 1) for a in range(3)]
            xs = xp[:, 1 + o[0]:13 + o[0], 1 + o[1]:9 + o[1], 1 + o[2]:9 + o[2]]
            ref[p * 8 + t] = np.einsum("bdhwi,bdhwo->io", xs, dyp, optimize=True)
    xd, dyd = dev(x), dev(dy)
    out = torch.full((64, 128, 64), float("nan"), device="cuda")
    assert lib().rdgan_op_upconv_wgrad_slab16(ptr(xd), ptr(dyd), ptr(out), B, stream()) == 0
    got = out.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(got))
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max(), (np.abs(got - ref).max(), np.abs(ref).max())
    out2 = torch.empty_like(out)
    assert lib().rdgan_op_upconv_wgrad_slab16(ptr(xd), ptr(dyd), ptr(out2), B, stream()) == 0
    assert torch.equal(out, out2)                                   # deterministic


@pytest.mark.parametrize("B", [1, 5, 70])
def test_d2_wgrad_slab_kernel_vs_definition(B):
    """k_d2_wgrad_slab16 alone (rdgan_op_d2_wgrad_slab16): weight gradient of the critic's second layer (backward of T:291,
    stride 2, 'same': dW[tap] = sum_o x[2 o + tap - 1] (outer) dy[o]) against torch autograd of the definition in fp64 on the
    bf16-rounded operands; fp32 accumulation of exact bf16 products: 2e-5 of the largest entry.  B = 70: 64 groups of workgroups,
    B = 1, 5: 8 groups, some without a sample."""
    g = torch.Generator(); g.manual_seed(400 + B)
    x = torch.randn((B, 11, 7, 7, 64), generator=g)
    dy = torch.randn((B, 6, 4, 4, 128), generator=g)
    xr, dyr = x.bfloat16().double(), dy.bfloat16().double()
    w = torch.zeros((3, 3, 3, 64, 128), dtype=torch.float64, requires_grad=True)
    y = ot._conv3d_tf(xr, w, torch.zeros(128, dtype=torch.float64), 2, (1, 1, 1), (6, 4, 4))
    (ref,) = torch.autograd.grad((y * dyr).sum(), w)
    ref = ref.numpy()
    xd, dyd = dev(x.numpy()), dev(dy.numpy())
    out = torch.full((3, 3, 3, 64, 128), float("nan"), device="cuda")
    assert lib().rdgan_op_d2_wgrad_slab16(ptr(xd), ptr(dyd), ptr(out), B, stream()) == 0
    got = out.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(got))
    err = np.abs(got - ref).max(axis=(3, 4)) / np.abs(ref).max()
    assert err.max() < 2e-5, err
    out2 = torch.empty_like(out)
    assert lib().rdgan_op_d2_wgrad_slab16(ptr(xd), ptr(dyd), ptr(out2), B, stream()) == 0
    assert torch.equal(out, out2)                                   # deterministic


@pytest.mark.parametrize("B,OH,OW", [(3, 8, 8), (2, 16, 16), (5, 12, 8), (70, 4, 4)])
def test_d2_wgrad_slab_tiled_kernel_vs_definition(B, OH, OW):
    """k_d2_wgrad_slab_t16 alone (rdgan_op_d2_wgrad_slab_t16): the same weight gradient with work items = 4 x 4 tiles of output
    positions (ndomain 32 / 48 / 64: 8, 12, 16 output positions a side; 4 x 4 = one tile whose halo lies wholly outside the picture)
    against torch autograd of the definition in fp64 on the bf16-rounded operands, 2e-5 of the largest entry; then a one-hot probe:
    ONE output-gradient position at a tile corner and one layer-1 position at a time -- every tap that joins them across the tile
    border must pick it up, no other tap may."""
    IH, IW = 2 * OH - 1, 2 * OW - 1
    g = torch.Generator(); g.manual_seed(500 + B + OH)
    x = torch.randn((B, 11, IH, IW, 64), generator=g)
    dy = torch.randn((B, 6, OH, OW, 128), generator=g)
    xr, dyr = x.bfloat16().double(), dy.bfloat16().double()
    w = torch.zeros((3, 3, 3, 64, 128), dtype=torch.float64, requires_grad=True)
    y = ot._conv3d_tf(xr, w, torch.zeros(128, dtype=torch.float64), 2, (1, 1, 1), (6, OH, OW))
    (ref,) = torch.autograd.grad((y * dyr).sum(), w)
    ref = ref.numpy()
    xd, dyd = dev(x.numpy()), dev(dy.numpy())
    out = torch.full((3, 3, 3, 64, 128), float("nan"), device="cuda")
    assert lib().rdgan_op_d2_wgrad_slab_t16(ptr(xd), ptr(dyd), ptr(out), B, OH, OW, stream()) == 0
    got = out.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(got))
    err = np.abs(got - ref).max(axis=(3, 4)) / np.abs(ref).max()
    assert err.max() < 2e-5, err
    out2 = torch.empty_like(out)
    assert lib().rdgan_op_d2_wgrad_slab_t16(ptr(xd), ptr(dyd), ptr(out2), B, OH, OW, stream()) == 0
    assert torch.equal(out, out2)                                   # deterministic
    if OH < 8:
        return
    # one-hot: output position (od, oh, ow) = (2, 4, 3) sits at a tile border along h (tiles of 4) and at a tile's last column
    dy1 = torch.zeros((1, 6, OH, OW, 128)); dy1[0, 2, 4, 3, 5] = 1.0
    for (id_, ih, iw) in ((3, 7, 5), (4, 8, 6), (5, 9, 7), (3, 7, 7), (4, 9, 5), (3, 6, 6)):
        x1 = torch.zeros((1, 11, IH, IW, 64)); x1[0, id_, ih, iw, 9] = 1.0
        o1 = torch.full((3, 3, 3, 64, 128), float("nan"), device="cuda")
        x1d, dy1d = dev(x1.numpy()), dev(dy1.numpy())
        assert lib().rdgan_op_d2_wgrad_slab_t16(ptr(x1d), ptr(dy1d), ptr(o1), 1, OH, OW, stream()) == 0
        want = np.zeros((3, 3, 3, 64, 128), np.float32)
        td, th, tw = id_ - 2 * 2 + 1, ih - 2 * 4 + 1, iw - 2 * 3 + 1          # input = 2 o + tap - 1
        if 0 <= td < 3 and 0 <= th < 3 and 0 <= tw < 3:
            want[td, th, tw, 9, 5] = 1.0
        assert np.array_equal(o1.cpu().numpy(), want), (id_, ih, iw)


@pytest.mark.parametrize("B", [1, 3, 520])
def test_upconv2_slab_kernel_vs_oracle(B):
    """k_upconv2_slab16 alone (rdgan_op_upconv2_slab16): generator block 2 of the bf16 storage mode -- UpSampling3D(2) + Conv3D(256
    -> 128, 3x3x3, 'same') + bias + PixelNorm + LeakyReLU(0.2) (T:335-338) on a 6 x 4 x 4 x 256 input -- against the fp64 oracle on
    the bf16-rounded input.  The kernel rounds the COLLAPSED weights (sums of up to 8 taps) and its output to bf16: 2^-8 each, so
    2e-2 of the largest output; the per-pixel 1/l2 at 1e-2.  B = 520: persistent workgroups walk two samples."""
    g = torch.Generator(); g.manual_seed(500 + B)
    nref = min(B, 3)
    x = torch.randn((B, 6, 4, 4, 256), generator=g)
    w = 0.02 * torch.randn((3, 3, 3, 256, 128), generator=g)
    bias = 0.05 * torch.randn((128,), generator=g)
    sel = [0, B // 2, B - 1][:nref]
    u = ot.upsample3d(x[sel].to(torch.bfloat16).double())
    pre = ot._conv3d_tf(u, w.double(), bias.double(), 1, (1, 1, 1), u.shape[1:4])
    ref = ot._lrelu(ot.pixel_norm(pre)).numpy()
    rinv_ref = (1.0 / torch.sqrt((pre * pre).mean(-1) + 1e-8)).numpy()
    xd, wd, bd = dev(x.numpy()), dev(w.numpy()), dev(bias.numpy())
    y = torch.full((B, 12, 8, 8, 128), float("nan"), device="cuda")
    rinv = torch.full((B, 12, 8, 8), float("nan"), device="cuda")
    assert lib().rdgan_op_upconv2_slab16(ptr(xd), ptr(wd), ptr(bd), ptr(y), ptr(rinv), B, stream()) == 0
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(rinv).all())
    assert rel_err(y[sel].cpu().numpy(), ref) < 2e-2
    assert rel_err(rinv[sel].cpu().numpy(), rinv_ref) < 1e-2
    # one-hot probe: a single kernel tap and channel pair moves exactly the source voxel the definition names (PixelNorm of a
    # one-channel row = sign * sqrt(128) up to eps, so compare supports and signs through rinv * pre)
    x1 = torch.zeros((1, 6, 4, 4, 256)); x1[..., 77] = (torch.arange(96, dtype=torch.float32).reshape(1, 6, 4, 4) % 61) + 1
    for tap in (0, 13, 26, 5):
        w1 = torch.zeros((3, 3, 3, 256, 128)); w1[tap // 9, (tap // 3) % 3, tap % 3, 77, 5] = 1.0
        u1 = ot.upsample3d(x1.double())
        want = ot._conv3d_tf(u1, w1.double(), torch.zeros(128).double(), 1, (1, 1, 1), u1.shape[1:4])[..., 5].numpy()
        x1d, w1d, b1d = dev(x1.numpy()), dev(w1.numpy()), dev(np.zeros(128, np.float32))
        y1 = torch.empty((1, 12, 8, 8, 128), device="cuda"); r1 = torch.empty((1, 12, 8, 8), device="cuda")
        assert lib().rdgan_op_upconv2_slab16(ptr(x1d), ptr(w1d), ptr(b1d), ptr(y1), ptr(r1), 1, stream()) == 0
        got = (y1[..., 5] / r1.clamp_max(1e3)).cpu().numpy()          # = pre (bf16-rounded) where the row is not empty
        np.testing.assert_allclose(got * (want != 0), want, rtol=2.0 ** -7, atol=1e-3), tap
        assert float(y1[..., :5].abs().max()) == 0 and float(y1[..., 6:].abs().max()) == 0


@pytest.mark.parametrize("B", [1, 3, 520])
def test_d2_fwd_slab_kernel_vs_oracle(B):
    """k_d2_fwd_slab16 alone (rdgan_op_d2_fwd_slab16): forward of the critic's second layer (T:291-293: Conv3D(128, 3x3x3, stride 2,
    'same') + bias + LeakyReLU + dropout) on an 11 x 7 x 7 x 64 input against the fp64 oracle on bf16-rounded operands: fp32
    accumulation, one rounding of the output to bf16 (2^-8 per element); with a seed the oracle's layer-2 mask (oracle/rng.py) times
    1/0.75.  B = 520: persistent workgroups walk two samples."""
    g = torch.Generator(); g.manual_seed(600 + B)
    x = torch.randn((B, 11, 7, 7, 64), generator=g)
    w = 0.05 * torch.randn((3, 3, 3, 64, 128), generator=g)
    bias = 0.1 * torch.randn((128,), generator=g)
    pre = ot._conv3d_tf(x.bfloat16().double(), w.bfloat16().double(), bias.double(), 2, (1, 1, 1), (6, 4, 4))
    ref = ot._lrelu(pre).numpy()
    xd, wd, bd = dev(x.numpy()), dev(w.numpy()), dev(bias.numpy())
    for seed in (0, 0x1234ABCD5):
        want = ref * (orng.dropout_scale_mask(seed, orng.STREAM_D2, ref.shape).astype(np.float64) if seed else 1.0)
        y = torch.full((B, 6, 4, 4, 128), float("nan"), device="cuda")
        assert lib().rdgan_op_d2_fwd_slab16(ptr(xd), ptr(wd), ptr(bd), ptr(y), B, seed, stream()) == 0
        got = y.cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(got))
        np.testing.assert_allclose(got, want, rtol=2.0 ** -8 + 1e-5, atol=2e-5 * np.abs(want).max())


@pytest.mark.parametrize("B", [1, 6, 300])
def test_d3_wgrad_slab_kernel_vs_definition(B):
    """k_d3_wgrad_slab16 alone (rdgan_op_d3_wgrad_slab16): weight gradient of the critic's third layer (backward of T:295, stride 2,
    no padding in front: dW[tap] = sum_o x[2 o + tap] (outer) dy[o]) against torch autograd of the definition in fp64 on the
    bf16-rounded operands: 2e-5 of the largest entry.  B = 1, 6: items with fewer than four samples; B = 300: 16 groups."""
    g = torch.Generator(); g.manual_seed(700 + B)
    x = torch.randn((B, 6, 4, 4, 128), generator=g)
    dy = torch.randn((B, 3, 2, 2, 256), generator=g)
    xr, dyr = x.bfloat16().double(), dy.bfloat16().double()
    w = torch.zeros((3, 3, 3, 128, 256), dtype=torch.float64, requires_grad=True)
    y = ot._conv3d_tf(xr, w, torch.zeros(256, dtype=torch.float64), 2, (0, 0, 0), (3, 2, 2))
    (ref,) = torch.autograd.grad((y * dyr).sum(), w)
    ref = ref.numpy()
    xd, dyd = dev(x.numpy()), dev(dy.numpy())
    out = torch.full((3, 3, 3, 128, 256), float("nan"), device="cuda")
    assert lib().rdgan_op_d3_wgrad_slab16(ptr(xd), ptr(dyd), ptr(out), B, stream()) == 0
    got = out.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(got))
    err = np.abs(got - ref).max(axis=(3, 4)) / np.abs(ref).max()
    assert err.max() < 2e-5, err
    out2 = torch.empty_like(out)
    assert lib().rdgan_op_d3_wgrad_slab16(ptr(xd), ptr(dyd), ptr(out2), B, stream()) == 0
    assert torch.equal(out, out2)
