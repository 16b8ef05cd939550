"""Keras whole-model HDF5 (`Model.save`, reference gan_train_cwgangp_pixelnorm.py:520-521;
`load_model`, raindisagg_gan_pretrained.py:43) weights reader / writer.

Layout handled (Keras 2.2.4-tf): group ``model_weights`` with attr ``layer_names``; one sub-group per
layer with attr ``weight_names``; float32 datasets under the (possibly nested) weight name.  Weights are
enumerated in that order and bound by POSITION and shape, because auto-generated layer names depend on
creation order in the writing process (the critic's layers are ``conv3d_4..7`` / ``dense_1`` when it is
built after the generator, T:361-362).  Uses h5py when importable, otherwise the dependency-free
``h5lite`` (same results; pinned against real h5py files in tests/test_h5lite.py).
"""
import numpy as np

from . import h5lite


def _names(attr):
    if attr is None:
        return []
    out = []
    for n in np.asarray(attr).ravel().tolist():
        out.append(n.decode() if isinstance(n, bytes) else str(n))
    return out


def _load_with_h5py(path):
    import h5py
    out = []
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        for lname in _names(g.attrs["layer_names"]):
            lg = g[lname]
            for wname in _names(lg.attrs.get("weight_names")):
                out.append(np.asarray(lg[wname], dtype=np.float32))
    return out


def _load_with_h5lite(path):
    root = h5lite.read_h5(path)
    g = root["model_weights"] if "model_weights" in root else root
    out = []
    for lname in _names(g.attrs["layer_names"]):
        lg = g[lname]
        for wname in _names(lg.attrs.get("weight_names")):
            out.append(np.asarray(lg[wname], dtype=np.float32))
    return out


def load_keras_h5(path, prefer_h5py=True):
    if prefer_h5py:
        try:
            import h5py  # noqa: F401
            return _load_with_h5py(path)
        except ImportError:
            pass
    return _load_with_h5lite(path)


def keras_tree(arrays, shapes, kind):
    """The weight part of the Keras layout as an h5lite tree (consumable by ``Model.load_weights``).
    ``model_config`` is not emitted: the reference's generator config embeds a marshalled Python lambda
    (check_numerics, T:349-350) that cannot be reproduced portably."""
    seq = "sequential" if kind == "generator" else "sequential_1"
    lg = h5lite.Group(attrs={"weight_names": np.array([n.encode() for n, _ in shapes])})
    for (n, s), a in zip(shapes, arrays):
        node = lg
        parts = n.split("/")
        for p in parts[:-1]:
            node = node.children.setdefault(p, h5lite.Group())
        node.children[parts[-1]] = np.asarray(a, np.float32).reshape(s)
    mw = h5lite.Group(attrs={"layer_names": np.array([seq.encode()]), "backend": b"tensorflow",
                             "keras_version": b"2.2.4-tf"}, children={seq: lg})
    return h5lite.Group(attrs={"backend": b"tensorflow", "keras_version": b"2.2.4-tf"}, children={"model_weights": mw})


def save_keras_h5(path, arrays, shapes, kind):
    h5lite.write_h5(path, keras_tree(arrays, shapes, kind))
