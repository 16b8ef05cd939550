"""The dependency-free HDF5 subset reader/writer behind the Keras `.h5` weight layout."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from pr_disagg_radar_gan_amd import h5io, h5lite, models, weights as W

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "keras_layout_tiny.h5")      # written by REAL h5py (make_keras_h5_fixture.py)
CONDA_PY = "/opt/conda/bin/python3.9"


def test_reads_file_written_by_real_h5py():
    t = h5lite.read_h5(FIXTURE)
    assert t.attrs["keras_version"] in ("2.2.4-tf", b"2.2.4-tf") or np.asarray(t.attrs["keras_version"]).item() == b"2.2.4-tf"
    assert json.loads(t.attrs["model_config"])["class_name"] == "Model"          # variable-length string via the global heap
    mw = t["model_weights"]
    assert [x.decode() for x in mw.attrs["layer_names"]] == ["input_2", "flatten", "input_1", "concatenate", "sequential"]
    arrays = h5io.load_keras_h5(FIXTURE, prefer_h5py=False)
    rng = np.random.default_rng(7)
    shapes = [(12, 24), (24,), (3, 3, 3, 4, 4), (4,), (3, 3, 3, 4, 2), (2,), (3, 3, 3, 2, 2), (2,), (3, 3, 3, 2, 1), (1,)]
    assert len(arrays) == 10
    for a, s in zip(arrays, shapes):
        assert np.array_equal(a, rng.standard_normal(s).astype(np.float32))


def test_generator_h5_roundtrip_full_size(tmp_path):
    rng = np.random.default_rng(0)
    g = W.init_generator(rng, 16)
    path = str(tmp_path / "gen_0001.h5")
    models.Generator(g, 16).save(path)
    back = W.load_weights(path)
    assert len(back) == 10 and all(np.array_equal(a, b) for a, b in zip(g, back))
    m = models.load_generator(path)
    assert m.ndomain == 16 and m.count_params() == 3974273
    d = W.init_critic(rng, 16)
    path = str(tmp_path / "disc_0001.h5")
    models.Critic(d, 16).save(path)
    assert all(np.array_equal(a, b) for a, b in zip(d, W.load_weights(path)))


def test_rejects_non_hdf5(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all")
    with pytest.raises(h5lite.H5Error):
        h5lite.read_h5(str(p))


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with real h5py here")
def test_real_h5py_reads_what_h5lite_writes(tmp_path):
    rng = np.random.default_rng(1)
    shapes = W.critic_param_shapes(16)
    arrays = W.init_critic(rng, 16)
    path = str(tmp_path / "disc.h5")
    h5io.save_keras_h5(path, arrays, shapes, "critic")
    code = (
        "import h5py, numpy as np, json, sys\n"
        "f = h5py.File(sys.argv[1], 'r'); g = f['model_weights']\n"
        "out = []\n"
        "for ln in g.attrs['layer_names']:\n"
        "    lg = g[ln.decode()]\n"
        "    for wn in lg.attrs['weight_names']:\n"
        "        d = np.asarray(lg[wn.decode()]); out.append([wn.decode(), list(d.shape), float(d.astype(np.float64).sum())])\n"
        "print(json.dumps(out))\n")
    r = subprocess.run([CONDA_PY, "-c", code, path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert [g[0] for g in got] == [n for n, _ in shapes]
    for (name, shp, s), a in zip(got, arrays):
        assert tuple(shp) == a.shape
        assert abs(s - float(a.astype(np.float64).sum())) < 1e-6 * max(1.0, abs(s))
