"""Parameter layout, initialisers and (de)serialisation of the two networks.

Keras weight order and shapes of the reference's models (gan_train_cwgangp_pixelnorm.py
:284-305 critic Sequential, :325-345 generator Sequential; largedomain variant :323-335).
A model's parameters live in ONE flat fp32 slab in this order (kernel, bias per layer):
that makes Adam a single kernel and the data-parallel gradient exchange a single RCCL call.
"""
import math
import os

import numpy as np

NHOURS = 24
LATENT_DIM = 100
NORM_SCALE = 127.4


def gen_param_shapes(ndomain=16, n_cond_channels=1):
    s = ndomain // 8
    n_in = LATENT_DIM + ndomain * ndomain * n_cond_channels
    n_nodes = 256 * s * s * 3
    return [
        ("dense/kernel:0", (n_in, n_nodes)), ("dense/bias:0", (n_nodes,)),
        ("conv3d/kernel:0", (3, 3, 3, 256, 256)), ("conv3d/bias:0", (256,)),
        ("conv3d_1/kernel:0", (3, 3, 3, 256, 128)), ("conv3d_1/bias:0", (128,)),
        ("conv3d_2/kernel:0", (3, 3, 3, 128, 64)), ("conv3d_2/bias:0", (64,)),
        ("conv3d_3/kernel:0", (3, 3, 3, 64, 1)), ("conv3d_3/bias:0", (1,)),
    ]


def critic_out_dims(ndomain=16):
    """Output extents of the four stride-2 convs: 'valid' then three TF 'same' (ceil(n/2))."""
    dims = (NHOURS, ndomain, ndomain)
    out = []
    for li in range(4):
        dims = tuple((n - 3) // 2 + 1 for n in dims) if li == 0 else tuple(-(-n // 2) for n in dims)
        out.append(dims)
    return out


def critic_param_shapes(ndomain=16, n_cond_channels=1):
    d, h, w = critic_out_dims(ndomain)[-1]
    cin = 1 + n_cond_channels
    return [
        ("conv3d_4/kernel:0", (3, 3, 3, cin, 64)), ("conv3d_4/bias:0", (64,)),
        ("conv3d_5/kernel:0", (3, 3, 3, 64, 128)), ("conv3d_5/bias:0", (128,)),
        ("conv3d_6/kernel:0", (3, 3, 3, 128, 256)), ("conv3d_6/bias:0", (256,)),
        ("conv3d_7/kernel:0", (3, 3, 3, 256, 256)), ("conv3d_7/bias:0", (256,)),
        ("dense_1/kernel:0", (d * h * w * 256, 1)), ("dense_1/bias:0", (1,)),
    ]


def param_count(shapes):
    return int(sum(int(np.prod(s)) for _, s in shapes))


def flatten(arrays):
    return np.concatenate([np.asarray(a, np.float32).ravel() for a in arrays])


def unflatten(flat, shapes):
    out, off = [], 0
    flat = np.asarray(flat)
    for _, s in shapes:
        n = int(np.prod(s))
        out.append(flat[off:off + n].reshape(s).copy())
        off += n
    if off != flat.size:
        raise ValueError(f"slab has {flat.size} elements, layout needs {off}")
    return out


def init_generator(rng, ndomain=16, n_cond_channels=1):
    """RandomNormal(stddev=0.02) kernels, zero biases (reference :315)."""
    return [(rng.standard_normal(s) * 0.02).astype(np.float32) if n.endswith("kernel:0") else np.zeros(s, np.float32)
            for n, s in gen_param_shapes(ndomain, n_cond_channels)]


def init_critic(rng, ndomain=16, n_cond_channels=1):
    """Keras default glorot_uniform kernels, zero biases (reference :286-304 pass none)."""
    out = []
    for n, s in critic_param_shapes(ndomain, n_cond_channels):
        if n.endswith("kernel:0"):
            rec = int(np.prod(s[:-2]))
            lim = math.sqrt(6.0 / (rec * s[-2] + rec * s[-1]))
            out.append(rng.uniform(-lim, lim, s).astype(np.float32))
        else:
            out.append(np.zeros(s, np.float32))
    return out


def infer_config_from_gen(arrays):
    """(ndomain, n_cond_channels) of a generator from its first Dense kernel [100 + nd*nd*nc, 256*(nd/8)^2*3]
    (T:318-326; nc = 2/3 for the revision1/additional_inputs variants)."""
    n_in, n_nodes = arrays[0].shape
    s = int(round(math.sqrt(n_nodes / 768.0)))
    nd = 8 * s
    if s < 1 or 768 * s * s != n_nodes or (n_in - LATENT_DIM) % (nd * nd) or not 1 <= (n_in - LATENT_DIM) // (nd * nd) <= 3:
        raise ValueError(f"cannot infer ndomain from a dense kernel of shape {arrays[0].shape}")
    return nd, (n_in - LATENT_DIM) // (nd * nd)


def infer_ndomain_from_gen(arrays):
    return infer_config_from_gen(arrays)[0]


def save_weights(path, arrays, shapes, kind):
    """.npz container or Keras-layout .h5 (h5io: h5py when present, else the built-in h5lite)."""
    if path.endswith(".npz"):
        np.savez(path, __kind__=np.array(kind), **{f"{i:02d}:{n}": a for i, ((n, _), a) in enumerate(zip(shapes, arrays))})
        return
    from . import h5io
    h5io.save_keras_h5(path, arrays, shapes, kind)


def load_weights(path):
    """Returns the list of arrays in Keras weight order.  .h5 files are read by enumerating
    layer_names / weight_names in order and binding by position and shape, never by name."""
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"weight file {path!r} not found (the reference ships its trained_models/*.h5 blobs separately)")
    if path.endswith(".npz"):
        with np.load(path) as f:
            keys = sorted(k for k in f.files if k != "__kind__")
            return [f[k].astype(np.float32) for k in keys]
    from . import h5io
    return h5io.load_keras_h5(path)
