"""Writes tests/golden/keras_layout_tiny.h5 with REAL h5py (run with an interpreter that has it, e.g.
/opt/conda/bin/python3.9): the Keras 2.2.4-tf whole-model layout of the reference's generator checkpoint
(functional model with a nested Sequential, gan_train_cwgangp_pixelnorm.py:312-357, saved at :520) with tiny
stand-in shapes -- root attrs incl. a variable-length-string ``model_config``, ``model_weights`` with
``layer_names``, per-layer ``weight_names``, datasets under nested name paths.  Fixture data only: shapes are
NOT the real network's, values are seeded noise.  Used to pin pr_disagg_radar_gan_amd/h5lite.py's reader
against files produced by the HDF5 library itself."""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(7)
    names = ["dense/kernel:0", "dense/bias:0", "conv3d/kernel:0", "conv3d/bias:0", "conv3d_1/kernel:0", "conv3d_1/bias:0",
             "conv3d_2/kernel:0", "conv3d_2/bias:0", "conv3d_3/kernel:0", "conv3d_3/bias:0"]
    shapes = [(12, 24), (24,), (3, 3, 3, 4, 4), (4,), (3, 3, 3, 4, 2), (2,), (3, 3, 3, 2, 2), (2,), (3, 3, 3, 2, 1), (1,)]
    path = os.path.join(HERE, "keras_layout_tiny.h5")
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = b"2.2.4-tf"
        f.attrs["backend"] = b"tensorflow"
        f.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {"name": "model", "layers": ["..."] * 40}})
        g = f.create_group("model_weights")
        layers = ["input_2", "flatten", "input_1", "concatenate", "sequential"]
        g.attrs["layer_names"] = np.array([n.encode() for n in layers])
        g.attrs["backend"] = b"tensorflow"
        g.attrs["keras_version"] = b"2.2.4-tf"
        for n in layers[:-1]:
            g.create_group(n).attrs["weight_names"] = np.array([], dtype="S1")
        s = g.create_group("sequential")
        s.attrs["weight_names"] = np.array([n.encode() for n in names])
        for n, shp in zip(names, shapes):
            s.create_dataset(n, data=rng.standard_normal(shp).astype(np.float32))
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
