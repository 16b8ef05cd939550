"""numpy restatement of the cWGAN-GP forward path, written from the op definitions.

TEST INFRASTRUCTURE (see oracle/__init__.py; parity unpinned).  No convolution library is
used: Conv3D is an explicit loop over the 27 taps.  Every function cites the reference
line it follows (T = gan_train_cwgangp_pixelnorm.py, P = raindisagg_gan_pretrained.py,
L = alternative_domains/gan_train_cwgangp_pixelnorm_largedomain.py).

Layouts are the reference's (Keras/TF): activations NDHWC, Conv3D kernels
(kd,kh,kw,Cin,Cout), Dense kernels (in,out).  dtype follows the inputs (use float64 arrays
for the fp64 oracle).
"""
import math
import numpy as np

NHOURS = 24
LATENT_DIM = 100          # T:69
NORM_SCALE = 127.4        # T:64, P:13
LRELU_ALPHA = 0.2         # T:288,327
PIXELNORM_EPS = 1.0e-8    # T:261
DROPOUT_RATE = 0.25       # T:289
GP_WEIGHT = 10.0          # T:392 (the literal, not GRADIENT_PENALTY_WEIGHT)


# ----------------------------------------------------------------------------------
# shapes / parameter layout
# ----------------------------------------------------------------------------------
def gen_param_shapes(ndomain=16, n_cond_channels=1):
    """Keras weight order of the generator's Sequential (T:325-345, L:323-335)."""
    s = ndomain // 8
    n_in = LATENT_DIM + ndomain * ndomain * n_cond_channels      # T:322-323
    n_nodes = 256 * s * s * 3                                     # T:318, L:325
    return [
        ("dense/kernel", (n_in, n_nodes)), ("dense/bias", (n_nodes,)),
        ("conv3d/kernel", (3, 3, 3, 256, 256)), ("conv3d/bias", (256,)),
        ("conv3d_1/kernel", (3, 3, 3, 256, 128)), ("conv3d_1/bias", (128,)),
        ("conv3d_2/kernel", (3, 3, 3, 128, 64)), ("conv3d_2/bias", (64,)),
        ("conv3d_3/kernel", (3, 3, 3, 64, 1)), ("conv3d_3/bias", (1,)),
    ]


def critic_geometry(ndomain=16):
    """Per-layer (in_dims, out_dims, pad_before) of the four strided convs (T:286-299).
    TF rule: 'valid' out=floor((n-k)/s)+1; 'same' out=ceil(n/s),
    pad_total=max((out-1)*s+k-n,0), before=pad_total//2 (extra pad at the END)."""
    dims = (NHOURS, ndomain, ndomain)
    geo = []
    for li in range(4):
        if li == 0:
            out = tuple((n - 3) // 2 + 1 for n in dims)
            pad = (0, 0, 0)
        else:
            out = tuple(-(-n // 2) for n in dims)
            pad = tuple(max((o - 1) * 2 + 3 - n, 0) // 2 for n, o in zip(dims, out))
        geo.append((dims, out, pad))
        dims = out
    return geo


def critic_param_shapes(ndomain=16, n_cond_channels=1):
    """Keras weight order of the critic's Sequential (T:284-305)."""
    geo = critic_geometry(ndomain)
    d, h, w = geo[-1][1]
    cin = 1 + n_cond_channels
    return [
        ("conv3d_4/kernel", (3, 3, 3, cin, 64)), ("conv3d_4/bias", (64,)),
        ("conv3d_5/kernel", (3, 3, 3, 64, 128)), ("conv3d_5/bias", (128,)),
        ("conv3d_6/kernel", (3, 3, 3, 128, 256)), ("conv3d_6/bias", (256,)),
        ("conv3d_7/kernel", (3, 3, 3, 256, 256)), ("conv3d_7/bias", (256,)),
        ("dense_1/kernel", (d * h * w * 256, 1)), ("dense_1/bias", (1,)),
    ]


def param_count(shapes):
    return int(sum(int(np.prod(s)) for _, s in shapes))


def unflatten(flat, shapes):
    out, off = [], 0
    for _, s in shapes:
        n = int(np.prod(s))
        out.append(np.asarray(flat[off:off + n]).reshape(s))
        off += n
    assert off == len(flat), (off, len(flat))
    return out


def flatten(params):
    return np.concatenate([np.asarray(p).ravel() for p in params])


def init_generator(rng, ndomain=16, n_cond_channels=1, dtype=np.float32):
    """RandomNormal(stddev=0.02) kernels, zero biases (T:315,326-345)."""
    out = []
    for name, s in gen_param_shapes(ndomain, n_cond_channels):
        out.append((rng.standard_normal(s) * 0.02).astype(dtype) if name.endswith("kernel")
                   else np.zeros(s, dtype))
    return out


def init_critic(rng, ndomain=16, n_cond_channels=1, dtype=np.float32):
    """Keras defaults: glorot_uniform kernels, zero biases (T:286-304 pass no initializer).
    fan_in = receptive*Cin, fan_out = receptive*Cout, limit = sqrt(6/(fan_in+fan_out))."""
    out = []
    for name, s in critic_param_shapes(ndomain, n_cond_channels):
        if name.endswith("kernel"):
            rec = int(np.prod(s[:-2]))
            limit = math.sqrt(6.0 / (rec * s[-2] + rec * s[-1]))
            out.append(rng.uniform(-limit, limit, s).astype(dtype))
        else:
            out.append(np.zeros(s, dtype))
    return out


# ----------------------------------------------------------------------------------
# ops
# ----------------------------------------------------------------------------------
def leaky_relu(x, alpha=LRELU_ALPHA):
    """tf.keras.layers.LeakyReLU(alpha=0.2) (T:288): x if x > 0 else alpha*x."""
    return np.where(x > 0, x, alpha * x)


def upsample3d(x):
    """UpSampling3D(size=(2,2,2)) (T:330): nearest repeat on D,H,W."""
    return x.repeat(2, axis=1).repeat(2, axis=2).repeat(2, axis=3)


def conv3d(x, w, b=None, stride=1, pad=(1, 1, 1), out_dims=None):
    """Conv3D cross-correlation (T:286-299,331-345).  x (B,D,H,W,Cin), w (3,3,3,Cin,Cout).
    ``pad`` is the zero padding BEFORE each axis; the padding after is whatever the output
    extent ``out_dims`` needs (TF 'same' puts the extra pad at the end)."""
    B, D, H, W, Cin = x.shape
    kd, kh, kw, _, Cout = w.shape
    if out_dims is None:
        out_dims = tuple((n + 2 * p - k) // stride + 1 for n, p, k in zip((D, H, W), pad, (kd, kh, kw)))
    need = [(o - 1) * stride + k for o, k in zip(out_dims, (kd, kh, kw))]
    after = [max(nd - n - p, 0) for nd, n, p in zip(need, (D, H, W), pad)]
    xp = np.pad(x, ((0, 0), (pad[0], after[0]), (pad[1], after[1]), (pad[2], after[2]), (0, 0)))
    Do, Ho, Wo = out_dims
    y = np.zeros((B, Do, Ho, Wo, Cout), dtype=np.result_type(x, w))
    for a in range(kd):
        for bb in range(kh):
            for c in range(kw):
                patch = xp[:, a:a + (Do - 1) * stride + 1:stride,
                           bb:bb + (Ho - 1) * stride + 1:stride,
                           c:c + (Wo - 1) * stride + 1:stride, :]
                y += patch @ w[a, bb, c]
    if b is not None:
        y = y + b
    return y


def pixel_norm(x):
    """PixelNormalization.call (T:255-266): x / sqrt(mean(x**2, -1, keepdims) + 1e-8)."""
    m = np.mean(x ** 2.0, axis=-1, keepdims=True) + x.dtype.type(PIXELNORM_EPS)
    return x / np.sqrt(m)


def softmax_hours(x):
    """Softmax(axis=1) (T:347): over the 24-hour axis, per grid point."""
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


# ----------------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------------
def generator_forward(params, z, cond, return_intermediates=False):
    """create_generator (T:312-357; L:317-364 for ndomain=64).  z (B,100), cond
    (B,nd,nd,nc) normalised daily sums -> (B,24,nd,nd,1) hourly fractions."""
    Wd, bd, W1, b1, W2, b2, W3, b3, W4, b4 = params
    B = z.shape[0]
    nd = cond.shape[1]
    s = nd // 8
    x = np.concatenate([z, cond.reshape(B, -1)], axis=1)              # T:322-323
    h0 = leaky_relu(x @ Wd + bd).reshape(B, 3, s, s, 256)            # T:326-328
    h1 = leaky_relu(pixel_norm(conv3d(upsample3d(h0), W1, b1)))       # T:330-333
    h2 = leaky_relu(pixel_norm(conv3d(upsample3d(h1), W2, b2)))       # T:335-338
    h3 = leaky_relu(pixel_norm(conv3d(upsample3d(h2), W3, b3)))       # T:340-343
    logits = conv3d(h3, W4, b4)                                       # T:345
    out = softmax_hours(logits)                                       # T:347
    if not np.all(np.isfinite(out)):                                  # T:349-350
        raise FloatingPointError("found nan in output of per_gridpoint_softmax")
    if return_intermediates:
        return out, dict(h0=h0, h1=h1, h2=h2, h3=h3, logits=logits)
    return out


def critic_input(sample, cond):
    """T:275-282: cond (B,nd,nd,nc) repeated 24x on the hour axis, concatenated as extra
    channel(s) behind the sample (B,24,nd,nd,1)."""
    cond_rep = np.repeat(cond[:, None], NHOURS, axis=1)
    return np.concatenate([sample, cond_rep], axis=-1)


def critic_forward(params, sample, cond, masks=None, return_intermediates=False):
    """create_discriminator (T:272-309).  ``masks``: four arrays (0 or 1/0.75) shaped like
    the conv outputs -- inverted dropout as applied by train_on_batch (T:289-301); None =
    inference (``predict``)."""
    x = critic_input(sample, cond)
    nd = cond.shape[1]
    geo = critic_geometry(nd)
    hs, acts = [], []
    for li in range(4):
        W, b = params[2 * li], params[2 * li + 1]
        _, out_dims, pad = geo[li]
        a = conv3d(x, W, b, stride=2, pad=pad, out_dims=out_dims)
        x = leaky_relu(a)
        if masks is not None:
            x = x * masks[li]
        acts.append(a)
        hs.append(x)
    Wl, bl = params[8], params[9]
    v = x.reshape(x.shape[0], -1) @ Wl + bl                            # T:303-304
    if return_intermediates:
        return v, dict(a=acts, h=hs)
    return v


def random_weighted_average(real, fake, alpha):
    """RandomWeightedAverage.call (T:221-224) with alpha (B,) supplied by the caller."""
    a = alpha.reshape(-1, 1, 1, 1, 1)
    return a * real + (1 - a) * fake


def wasserstein_loss(y_true, y_pred):
    """T:215-216."""
    return np.mean(y_true * y_pred)


def generate_scenarios(gen_params, cond, n_scenarios, latent=None):
    """raindisagg_gan_pretrained.generate_scenarios (P:52-65).  cond (nd,nd,1) in mm/day."""
    cond_n = cond / NORM_SCALE                                         # P:54
    if latent is None:
        latent = np.random.normal(size=(n_scenarios, LATENT_DIM))     # P:56
    cond_batch = np.repeat(cond_n[np.newaxis], repeats=n_scenarios, axis=0)   # P:59
    generated = generator_forward(gen_params, latent.astype(gen_params[0].dtype),
                                  cond_batch.astype(gen_params[0].dtype))
    generated = generated.squeeze()                                    # P:62
    return generated * cond_n.squeeze() * NORM_SCALE                   # P:64


# ----------------------------------------------------------------------------------
# gradient-penalty path from the definitions (used to pin the autograd restatement)
# ----------------------------------------------------------------------------------
def conv3d_input_grad(gy, w, in_dims, stride, pad):
    """Adjoint of conv3d w.r.t. its input, by scattering every tap (definition of the
    transposed cross-correlation).  gy (B,Do,Ho,Wo,Cout) -> (B,D,H,W,Cin)."""
    B, Do, Ho, Wo, Cout = gy.shape
    D, H, W = in_dims
    kd, kh, kw, Cin, _ = w.shape
    need = [(o - 1) * stride + k for o, k in zip((Do, Ho, Wo), (kd, kh, kw))]
    ext = [max(nd, n + p) for nd, n, p in zip(need, (D, H, W), pad)]
    gxp = np.zeros((B, ext[0], ext[1], ext[2], Cin), dtype=gy.dtype)
    for a in range(kd):
        for bb in range(kh):
            for c in range(kw):
                gxp[:, a:a + (Do - 1) * stride + 1:stride,
                    bb:bb + (Ho - 1) * stride + 1:stride,
                    c:c + (Wo - 1) * stride + 1:stride, :] += gy @ w[a, bb, c].T
    return gxp[:, pad[0]:pad[0] + D, pad[1]:pad[1] + H, pad[2]:pad[2] + W, :]


def critic_input_gradient(params, sample, cond, masks=None):
    """d D(sample,cond) / d sample, by the explicit chain the HIP path uses: the critic is
    piecewise linear in its input given the LeakyReLU signs and dropout masks, so the
    gradient is a chain of transposed convs gated by slope*mask (what K.gradients
    evaluates at T:240)."""
    nd = cond.shape[1]
    geo = critic_geometry(nd)
    v, inter = critic_forward(params, sample, cond, masks, True)
    B = sample.shape[0]
    g = np.broadcast_to(params[8].reshape((1,) + inter["h"][3].shape[1:]), inter["h"][3].shape)
    for li in (3, 2, 1, 0):
        slope = np.where(inter["a"][li] > 0, 1.0, LRELU_ALPHA)
        if masks is not None:
            slope = slope * masks[li]
        u = g * slope
        g = conv3d_input_grad(u, params[2 * li], geo[li][0], 2, geo[li][2])
    return v, g[..., :1]


def critic_loss(dparams, x_real, fake, cond, alpha, masks3):
    """Critic objective of T:388-392 from the definitions: returns (total, valid, fake, gp).
    masks3: per layer masks for the 3B batch [real; fake; interpolated] or None."""
    B = x_real.shape[0]
    xhat = random_weighted_average(x_real, fake, alpha)
    cut = lambda lo, hi: None if masks3 is None else [m[lo:hi] for m in masks3]
    v_real = critic_forward(dparams, x_real, cond, cut(0, B))
    v_fake = critic_forward(dparams, fake, cond, cut(B, 2 * B))
    _, g = critic_input_gradient(dparams, xhat, cond, cut(2 * B, 3 * B))
    gp = np.sqrt((g ** 2).reshape(B, -1).sum(1, keepdims=True)) - 1          # T:241
    l_valid = wasserstein_loss(-np.ones((B, 1)), v_real)
    l_fake = wasserstein_loss(np.ones((B, 1)), v_fake)
    l_gp = np.mean(gp ** 2)
    return l_valid + l_fake + GP_WEIGHT * l_gp, l_valid, l_fake, l_gp
