"""numpy restatement of the reference's input pipeline (test infrastructure; see oracle/__init__.py).
T = gan_train_cwgangp_pixelnorm.py, V = compute_valid_indices.py."""
import numpy as np


def gather_real(data, indices_all, ixs, ndomain, norm_scale=127.4):
    """T:150-166 with the window view written as explicit slices: returns (batch, batch_cond)."""
    idcs = indices_all[ixs]
    n = len(ixs)
    batch = np.empty((n, data.shape[1], ndomain, ndomain), np.float32)
    for i, (t, y, x) in enumerate(idcs):                 # == data_wview[t, :, y, x] of T:154-155
        batch[i] = data[t, :, y:y + ndomain, x:x + ndomain]
    batch = np.expand_dims(batch, -1)                    # T:157
    batch_cond = np.sum(batch, axis=1)                   # T:159
    for i in range(n):                                   # T:162-163
        batch[i] = batch[i] / batch_cond[i]
    batch_cond = batch_cond / norm_scale                 # T:166
    return batch, batch_cond


def extra_condition(batch_cond, idcs_batch, ndomain, kind, timelist_all=None, min_lonidx=0, max_lonidx=1):
    """The extra condition channels of revision1/additional_inputs, per sample loops instead of the reference's
    tile + transpose: 'lon' (…_lon.py:175-184) appends (xidx - min_lonidx)/max_lonidx, 'doy' (…_doy.py:173-186)
    appends sin and cos of 2 pi doy/365 with doy = timelist_all[tidx]; each constant over the tile."""
    n = len(idcs_batch)
    nex = 1 if kind == "lon" else 2
    extra = np.empty((n, ndomain, ndomain, nex), np.float64)
    for i, (t, y, x) in enumerate(idcs_batch):
        if kind == "lon":
            extra[i, :, :, 0] = (x - min_lonidx) / max_lonidx
        else:
            extra[i, :, :, 0] = np.sin(2 * np.pi * timelist_all[t] / 365)
            extra[i, :, :, 1] = np.cos(2 * np.pi * timelist_all[t] / 365)
    return np.concatenate([batch_cond, extra], axis=-1)


def valid_indices(data, ndomain=16, stride=16, tp_thresh_daily=5, n_thresh=20):
    """V:74-92 (the numba loop, plain numpy)."""
    n_days, _, ny, nx = data.shape
    out = []
    for tidx in range(n_days):
        sub = np.sum(data[tidx], axis=0)
        for ii in range(0, ny - ndomain, stride):
            for jj in range(0, nx - ndomain, stride):
                subsub = sub[ii:ii + ndomain, jj:jj + ndomain]
                if not np.any(np.isnan(subsub)):
                    if np.sum(subsub > tp_thresh_daily) >= n_thresh:
                        out.append((tidx, ii, jj))
    return out


def crps_ensemble(obs, ens):
    """properscoring.crps_ensemble(obs, ens, axis=0) by its definition (O(n^2), fp64):
    mean_i |x_i - y| - 0.5 mean_{i,j} |x_i - x_j|  (generate_and_evaluate_crps.py:188)."""
    ens = np.asarray(ens, np.float64); obs = np.asarray(obs, np.float64)
    a = np.mean(np.abs(ens - obs[None]), axis=0)
    b = np.zeros_like(a)
    for i in range(ens.shape[0]):
        b += np.mean(np.abs(ens - ens[i][None]), axis=0)
    return a - 0.5 * b / ens.shape[0]
