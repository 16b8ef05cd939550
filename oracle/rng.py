"""Counter-based RNG shared by the oracle and the HIP path (test infrastructure).

The reference draws dropout masks (gan_train_cwgangp_pixelnorm.py:289-301) and the
RandomWeightedAverage alpha (…:223) from TensorFlow's stateful global RNG, which is never
seeded, so no bit pattern is pinned by the reference.  Both sides of the parity tests use
this stateless definition instead (mirrored in csrc/rdgan_rng.h):

    key(seed, stream) = mix(lo32(seed) ^ mix(hi32(seed) ^ 0x9E3779B9)) + stream * 0x85EBCA6B
    bits(idx)         = mix(mix(idx) ^ key)                       (all uint32, wrapping)
    uniform(idx)      = (bits >> 8) * 2**-24            in [0, 1)
    keep(idx)         = byte (idx & 3) of bits(idx >> 2) >= 64      P(keep) = 192/256 = 0.75 for rate 0.25
                        (one hash word decides four consecutive elements: the kernels' epilogues hold four consecutive
                        channels per lane and pay the hash once per quad)

``mix`` is the lowbias32 integer finaliser.  ``idx`` is the flat NDHWC element index of the
tensor the mask is applied to (for the critic step: the 3B batch [real; fake; interpolated]).
"""
import numpy as np

STREAM_D1, STREAM_D2, STREAM_D3, STREAM_D4, STREAM_ALPHA = 1, 2, 3, 4, 5
DROP_THRESHOLD = 64  # 0.25 * 2**8: one byte of the hash word per element
DROP_SCALE = np.float32(1.0) / np.float32(0.75)


def mix32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def make_key(seed, stream):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo = np.uint32(seed & 0xFFFFFFFF)
    hi = np.uint32(seed >> 32)
    with np.errstate(over="ignore"):
        k = mix32(lo ^ mix32(hi ^ np.uint32(0x9E3779B9)))
        k = np.uint32((int(k) + int(stream) * 0x85EBCA6B) & 0xFFFFFFFF)
    return k


def bits(seed, stream, n, start=0):
    idx = np.arange(start, start + n, dtype=np.uint64).astype(np.uint32)
    return mix32(mix32(idx) ^ make_key(seed, stream))


def uniform(seed, stream, n, start=0):
    """float32 uniforms in [0,1) for element indices start..start+n-1."""
    return (bits(seed, stream, n, start) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def dropout_scale_mask(seed, stream, shape):
    """float32 mask holding 0 or 1/0.75 (inverted dropout, rate 0.25) of the given shape.
    seed == 0 means dropout off (all ones) -- the ``predict`` path of the reference."""
    n = int(np.prod(shape))
    if seed == 0:
        return np.ones(shape, np.float32)
    words = np.repeat(bits(seed, stream, (n + 3) // 4), 4)[:n]
    byte = (words >> (np.uint32(8) * (np.arange(n, dtype=np.uint32) & np.uint32(3)))) & np.uint32(0xFF)
    keep = byte >= np.uint32(DROP_THRESHOLD)
    return np.where(keep, DROP_SCALE, np.float32(0)).astype(np.float32).reshape(shape)
