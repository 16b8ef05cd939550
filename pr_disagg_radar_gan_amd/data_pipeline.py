"""Device-side input pipeline: the radar array lives in HBM (288 GB per MI355X hold the
reference's 8-year, hourly data set), tiles are gathered and normalised by a HIP kernel straight into the
tensors the training step consumes.  Replaces, for the hot loop, the reference's view_as_windows fancy-index
gather from a disk memmap, its per-sample Python divide loop and the multiprocessing queues
(gan_train_cwgangp_pixelnorm.py:143-193, :440-449), and compute_valid_indices.py's numba scan."""
import ctypes

import numpy as np
import torch

from . import _lib
from . import weights as W
from .engine import require_gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class DeviceDataset:
    def __init__(self, data, indices=None, ndomain=16, norm_scale=W.NORM_SCALE, device=None):
        """data: float32 array (n_days, 24, ny, nx) (numpy / memmap); indices: (n_samples, 3) (tidx, yidx, xidx)."""
        require_gpu()
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if data.ndim != 4 or data.shape[1] != W.NHOURS or data.dtype != np.float32:
            raise ValueError("data must be float32 with shape (n_days, 24, ny, nx)")       # reference :131-138
        self.n_days, _, self.ny, self.nx = data.shape
        self.ndomain, self.norm_scale = int(ndomain), float(norm_scale)
        self.data = torch.from_numpy(np.ascontiguousarray(data)).to(self.device)
        self.flags = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.indices = None
        self.extra = None
        if indices is not None:
            self.set_indices(indices)

    def set_extra_condition(self, kind, timelist=None, min_lonidx=0, max_lonidx=1):
        """Extra condition channels of the revision-1 variants, appended behind the daily sum by gather():
        'lon' -> (xidx - min_lonidx) / max_lonidx (…_lon.py:175-184); 'doy' -> sin, cos of 2 pi doy / 365 with
        doy = timelist[tidx] (…_doy.py:173-186); None -> the one-channel condition."""
        if kind is None:
            self.extra = None
        elif kind == 'lon':
            self.extra = ('lon', float(min_lonidx), float(max_lonidx))
        elif kind == 'doy':
            tl = np.asarray(timelist, dtype=np.float64)
            if tl.shape != (self.n_days,):
                raise ValueError("timelist must hold one day-of-year value per day of the data array")
            self.extra = ('doy', torch.from_numpy(tl).to(self.device))
        else:
            raise ValueError("extra condition kind must be None, 'lon' or 'doy'")

    @property
    def n_cond_channels(self):
        return 1 if self.extra is None else (2 if self.extra[0] == 'lon' else 3)

    def _extra_channels(self, sel, cond):
        nd = self.ndomain
        if self.extra[0] == 'lon':
            vals = [(sel[:, 2].double() - self.extra[1]) / self.extra[2]]
        else:
            ang = 2 * np.pi * self.extra[1][sel[:, 0].long()] / 365
            vals = [torch.sin(ang), torch.cos(ang)]
        planes = [v.float().view(-1, 1, 1, 1).expand(-1, nd, nd, 1) for v in vals]
        return torch.cat([cond] + planes, dim=-1).contiguous()

    def set_indices(self, indices):
        idx = np.ascontiguousarray(np.asarray(indices), dtype=np.int32)
        if idx.ndim != 2 or idx.shape[1] != 3:
            raise ValueError("indices must have shape (n_samples, 3)")
        if idx[:, 0].max() >= self.n_days or idx[:, 1].max() + self.ndomain > self.ny or idx[:, 2].max() + self.ndomain > self.nx:
            raise ValueError("index outside the data array")
        self.indices = torch.from_numpy(idx).to(self.device)
        self.n_samples = idx.shape[0]

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def gather(self, ixs, with_batch=True):
        """tiles at self.indices[ixs] -> (fractions (n,24,nd,nd,1) or None, cond (n,nd,nd,n_cond_channels)) device tensors"""
        ixs = torch.as_tensor(ixs, dtype=torch.long, device=self.device)
        sel = self.indices[ixs].contiguous()
        n, nd = sel.shape[0], self.ndomain
        batch = torch.empty((n, W.NHOURS, nd, nd, 1), dtype=torch.float32, device=self.device) if with_batch else None
        cond = torch.empty((n, nd, nd, 1), dtype=torch.float32, device=self.device)
        rc = self.lib.rdgan_data_gather(_p(self.data), self.n_days, self.ny, self.nx, _p(sel), n, nd, self.norm_scale,
                                        _p(batch) if with_batch else ctypes.c_void_p(0), _p(cond), _p(self.flags), self._stream())
        _lib.check(rc, None, "rdgan_data_gather")
        if self.extra is not None:
            cond = self._extra_channels(sel, cond)
        return batch, cond

    def check_flags(self):
        """the reference's asserts (T:169-172), checked once per call site instead of per batch element"""
        f = int(self.flags.item())
        if f & 1:
            raise AssertionError("NaN/Inf in gathered batch or condition (daily sum of zero or missing data)")
        if f & 2:
            raise AssertionError("hourly fraction outside [0, 1]")

    def sample_real(self, n_batch):
        """generate_real_samples (T:143-174): random valid tiles -> [fractions, normalised daily sums]"""
        ixs = np.random.randint(self.n_samples, size=n_batch)
        return self.gather(ixs, True)

    def sample_latent(self, n_batch, latent_dim=W.LATENT_DIM):
        """generate_latent_points (T:177-193): latent noise and the conditions of random real tiles"""
        latent = torch.from_numpy(np.random.normal(size=(n_batch, latent_dim)).astype(np.float32)).to(self.device)
        ixs = np.random.randint(0, self.n_samples, size=n_batch)
        _, cond = self.gather(ixs, False)
        return latent, cond

    def valid_indices(self, stride=16, tp_thresh_daily=5, n_thresh=20):
        """compute_valid_indices.py:74-92 -> list of (tidx, ii, jj) in the reference's order"""
        nd = self.ndomain
        nbi, nbj = len(range(0, self.ny - nd, stride)), len(range(0, self.nx - nd, stride))
        if nbi < 1 or nbj < 1:
            return []
        valid = torch.empty((self.n_days, nbi, nbj), dtype=torch.int32, device=self.device)
        rc = self.lib.rdgan_data_valid_tiles(_p(self.data), self.n_days, self.ny, self.nx, nd, int(stride),
                                             float(tp_thresh_daily), int(n_thresh), _p(valid), self._stream())
        _lib.check(rc, None, "rdgan_data_valid_tiles")
        t, i, j = np.nonzero(valid.cpu().numpy())
        return [(int(a), int(b) * stride, int(c) * stride) for a, b, c in zip(t, i, j)]
