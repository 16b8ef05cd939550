"""Keras whole-model HDF5 (`Model.save`, reference gan_train_cwgangp_pixelnorm.py:520-521;
`load_model`, raindisagg_gan_pretrained.py:43) reader / writer.

Layout handled (Keras 2.2.4-tf): group ``model_weights`` with attr ``layer_names``; one
sub-group per layer with attr ``weight_names``; datasets float32.  Weights are enumerated in
that order and bound by position, because auto-generated layer names depend on creation
order in the writing process.  Needs ``h5py`` (not installed in every interpreter): raises
ImportError with a clear message otherwise -- use the .npz container instead.
"""
import numpy as np


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError as e:  # pragma: no cover - depends on the interpreter
        raise ImportError("reading/writing Keras .h5 files needs h5py; use a .npz weight file "
                          "(pr_disagg_radar_gan_amd.weights.save_weights) in this interpreter") from e


def _names(attr):
    return [n.decode() if isinstance(n, bytes) else str(n) for n in attr]


def load_keras_h5(path):
    h5py = _h5py()
    out = []
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        for lname in _names(g.attrs["layer_names"]):
            lg = g[lname]
            for wname in _names(lg.attrs.get("weight_names", [])):
                out.append(np.asarray(lg[wname], dtype=np.float32))
    return out


def save_keras_h5(path, arrays, shapes, kind):
    """Writes the weight part of the Keras layout (consumable by ``Model.load_weights``).
    ``model_config`` is not emitted: the reference's generator config embeds a marshalled
    Python lambda (check_numerics, :349-350) that cannot be reproduced portably."""
    h5py = _h5py()
    seq = "sequential" if kind == "generator" else "sequential_1"
    with h5py.File(path, "w") as f:
        f.attrs["backend"] = np.bytes_("tensorflow")
        f.attrs["keras_version"] = np.bytes_("2.2.4-tf")
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = np.array([seq.encode()])
        g.attrs["backend"] = np.bytes_("tensorflow")
        g.attrs["keras_version"] = np.bytes_("2.2.4-tf")
        lg = g.create_group(seq)
        lg.attrs["weight_names"] = np.array([n.encode() for n, _ in shapes])
        for (n, s), a in zip(shapes, arrays):
            lg.create_dataset(n, data=np.asarray(a, np.float32).reshape(s))
