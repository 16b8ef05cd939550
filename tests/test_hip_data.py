"""-m gpu: device input pipeline (tile gather / normalisation, valid-tile scan) is BIT-IDENTICAL to the numpy
restatement of the reference's host code."""
import numpy as np
import pytest

from oracle import data_np as od

pytestmark = pytest.mark.gpu


def _data(n_days=5, ny=70, nx=53, seed=0):
    rng = np.random.default_rng(seed)
    d = rng.gamma(0.15, 3.0, (n_days, 24, ny, nx)).astype(np.float32)
    d[rng.random(d.shape) < 0.5] = 0.0                    # lots of dry hours, like radar data
    d[1, :, 10:14, 20:30] = np.nan                        # a patch of missing data on day 1
    return d


@pytest.mark.parametrize("nd,stride", [(16, 16), (16, 5), (32, 16)])
def test_valid_tile_scan_matches_reference_loop(nd, stride):
    from pr_disagg_radar_gan_amd.data_pipeline import DeviceDataset
    d = _data()
    ds = DeviceDataset(d, ndomain=nd)
    got = ds.valid_indices(stride=stride, tp_thresh_daily=5, n_thresh=20)
    ref = od.valid_indices(d, nd, stride, 5, 20)
    assert got == ref and len(ref) > 0
    assert not any(t == 1 and i < 14 and i + nd > 10 and j < 30 and j + nd > 20 for t, i, j in got)   # no box over the NaN patch


@pytest.mark.parametrize("nd", [16, 32])
def test_gather_is_bit_identical(nd):
    from pr_disagg_radar_gan_amd.data_pipeline import DeviceDataset
    d = _data(seed=1)
    idx = np.array(od.valid_indices(d, nd, 7, 5, 20))
    ds = DeviceDataset(d, idx, ndomain=nd)
    ixs = np.random.default_rng(2).integers(0, len(idx), 37)
    batch, cond = ds.gather(ixs)
    rb, rc = od.gather_real(d, idx, ixs, nd)
    assert batch.shape == (37, 24, nd, nd, 1) and cond.shape == (37, nd, nd, 1)
    assert np.array_equal(batch.cpu().numpy(), rb)
    assert np.array_equal(cond.cpu().numpy(), rc)
    np.testing.assert_allclose(batch.cpu().numpy().sum(axis=1), 1.0, rtol=1e-5)
    ds.check_flags()
    _, c2 = ds.gather(ixs, with_batch=False)
    assert np.array_equal(c2.cpu().numpy(), rc)
    np.random.seed(3)
    b3, c3 = ds.sample_real(8)
    np.random.seed(3)
    ix3 = np.random.randint(len(idx), size=8)
    assert np.array_equal(b3.cpu().numpy(), od.gather_real(d, idx, ix3, nd)[0])


def test_gather_flags_missing_data():
    from pr_disagg_radar_gan_amd.data_pipeline import DeviceDataset
    d = _data(seed=4)
    ds = DeviceDataset(d, np.array([[1, 8, 18]]), ndomain=16)      # overlaps the NaN patch
    ds.gather([0])
    with pytest.raises(AssertionError):
        ds.check_flags()
