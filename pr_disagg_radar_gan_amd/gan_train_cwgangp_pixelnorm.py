"""Host-side mirror of the reference's training script ``gan_train_cwgangp_pixelnorm.py`` (and of
its ``alternative_domains/…_largedomain.py`` variant through ``configure(ndomain=64, n_thresh=40)``)
on the MI355X-native engine.

The reference executes everything at import (loads data :117, builds the networks :361, plots
:413-425, trains :527).  Here the same names are functions/objects of an importable module and
``python -m pr_disagg_radar_gan_amd.gan_train_cwgangp_pixelnorm`` is the script:

    constants of :51-78, ``params`` (:113), create_generator(), create_discriminator(),
    PixelNormalization, RandomWeightedAverage, GradientPenalty, wasserstein_loss,
    generate_real_samples / generate_latent_points(_as_generator) / generate_fake_samples / generate,
    train(n_epochs, _batch_size, start_epoch=0), hist.

One training iteration = n_disc critic updates + 1 generator update (:468-482); the whole step
(three critic passes, gradient penalty double backward, Adam) runs in librdgan_hip.so, and with
torch.distributed initialised the minibatch is sharded over ranks (trainer.WGANGPTrainer).
"""
import os
import pickle

import numpy as np

from . import models
from . import weights as W

# ---- constants (reference :51-78)
startdate = '20090101'
enddate = '20161231'
ndomain = 16
stride = 16
tres = 1
tp_thresh_daily = 5
n_thresh = 20
norm_scale = W.NORM_SCALE
n_disc = 5
GRADIENT_PENALTY_WEIGHT = 10
latent_dim = W.LATENT_DIM
batch_size = 32
n_epoch_and_batch_size_list = ((50, 32),)
plot_format = 'png'
name = 'wgancp_pixelnorm'
nhours = 24 // tres
exchange = None           # gradient exchange of the data-parallel trainer: None / "auto", "allreduce", "sharded" (trainer.WGANGPTrainer)
n_channel = 1             # 2: + longitude index (revision1/additional_inputs/…_lon.py:126,136), 3: + sin/cos day of year (…_doy.py:135)

plotdir = f'plots_{name}/'
outdir = f'trained_models/{name}/'
converted_data_path = 'data/'
indices_data_path = 'data/'


def _params():
    return f'{startdate}-{enddate}-tp_thresh_daily{tp_thresh_daily}_n_thresh{n_thresh}_ndomain{ndomain}_stride{stride}'


params = _params()        # reference :113 -- names every artefact
data = None               # (n_days, 24, ny, nx) float32, np.load(mmap_mode='r') (reference :117)
indices_all = None        # (n_samples, 3) rows (tidx, yidx, xidx) (reference :120-124)
timelist_all = None       # (n_days,) day of year, n_channel == 3 only (…_doy.py:128, reformat_data_make_timelist.py)
min_lonidx = max_lonidx = 0   # n_channel == 2 only (…_lon.py:126-127)
n_samples = 0
generator = None
critic = None
hist = {'d_loss': [], 'g_loss': []}
_trainer = None
resume_from = None        # path of a trainer checkpoint to continue from (set before train(); see resume())
device_dataset = None     # data_pipeline.DeviceDataset when the radar array is resident in HBM (use_device_dataset)


def configure(**kw):
    """Change module constants (e.g. ``configure(ndomain=64, n_thresh=40)`` for the large-domain variant,
    reference L:59,65) and refresh ``params``."""
    global params
    g = globals()
    for k, v in kw.items():
        if k not in g:
            raise KeyError(k)
        g[k] = v
    if ndomain % 8:
        raise ValueError("ndomain must be a multiple of 8 (reference L:324)")
    if n_channel not in (1, 2, 3):
        raise ValueError("n_channel must be 1 (daily sum), 2 (+ longitude) or 3 (+ sin/cos day of year)")
    params = _params()


# ---- data (reference :111-140)
def load_data(data_ifile=None, indices_file=None, timelist_ifile=None):
    global data, indices_all, n_samples, timelist_all
    data_ifile = data_ifile or f'{converted_data_path}/{startdate}-{enddate}_tres{tres}.npy'
    indices_file = indices_file or f'{indices_data_path}/valid_indices_smhi_radar_{params}.pkl'
    data = np.load(data_ifile, mmap_mode='r')
    with open(indices_file, 'rb') as f:
        indices_all = np.array(pickle.load(f))
    if n_channel == 3:                                                        # …_doy.py:114,128
        timelist_all = np.load(timelist_ifile or f'{converted_data_path}/{startdate}-{enddate}_tres{tres}_doy.npy')
    _check_data()


def use_arrays(data_array, indices, timelist=None):
    """Train on in-memory arrays (tests, synthetic data) instead of the reference's files."""
    global data, indices_all, n_samples, timelist_all
    data, indices_all = data_array, np.asarray(indices)
    timelist_all = None if timelist is None else np.asarray(timelist)
    _check_data()


def _check_data():
    global n_samples, min_lonidx, max_lonidx
    if n_channel == 3 and (timelist_all is None or len(timelist_all) != data.shape[0]):
        raise ValueError("n_channel == 3 needs the day-of-year list, one entry per day of the data array")
    min_lonidx, max_lonidx = np.min(indices_all[:, 2]), np.max(indices_all[:, 2])      # …_lon.py:126-127
    n_days, nh, ny, nx = data.shape
    assert len(indices_all.shape) == 2 and indices_all.shape[1] == 3        # reference :131-138
    assert nh == 24 // tres
    assert np.max(indices_all[:, 0]) < n_days and np.max(indices_all[:, 1]) + ndomain <= ny \
        and np.max(indices_all[:, 2]) + ndomain <= nx
    assert data.dtype == 'float32'
    n_samples = len(indices_all)


def use_device_dataset(enable=True):
    """Keep the radar array in HBM and gather / normalise training tiles with the HIP kernel of
    data_pipeline.DeviceDataset instead of on the host (the reference's :143-193 run in worker processes)."""
    global device_dataset
    if not enable:
        device_dataset = None
        return None
    from .data_pipeline import DeviceDataset
    device_dataset = DeviceDataset(np.asarray(data), indices_all, ndomain=ndomain, norm_scale=norm_scale)
    if n_channel == 2:
        device_dataset.set_extra_condition('lon', min_lonidx=min_lonidx, max_lonidx=max_lonidx)
    elif n_channel == 3:
        device_dataset.set_extra_condition('doy', timelist=timelist_all)
    return device_dataset


def _gather_tiles(ixs):
    """(n, 24, ndomain, ndomain) windows at indices_all[ixs] -- what the reference does with
    view_as_windows + fancy indexing (:154-155)."""
    idcs = indices_all[ixs]
    out = np.empty((len(ixs), nhours, ndomain, ndomain), np.float32)
    for i, (t, y, x) in enumerate(idcs):
        out[i] = data[t, :, y:y + ndomain, x:x + ndomain]
    return out


def _add_extra_condition(batch_cond, idcs_batch):
    """The extra condition channels of the revision-1 variants, constant over each tile: the normalised
    longitude index (…_lon.py:175-184) or sin/cos of the day of year (…_doy.py:173-186)."""
    if n_channel == 1:
        return batch_cond
    plane = np.ones((1, ndomain, ndomain, 1))
    if n_channel == 2:
        lon = (idcs_batch[:, 2] - min_lonidx) / max_lonidx
        extra = [lon[:, None, None, None] * plane]
    else:
        doy = timelist_all[idcs_batch[:, 0]]
        extra = [np.sin(2 * np.pi * doy / 365)[:, None, None, None] * plane,
                 np.cos(2 * np.pi * doy / 365)[:, None, None, None] * plane]
    return np.concatenate([batch_cond] + extra, axis=-1)


def _real_batch(n_batch):
    ixs = np.random.randint(n_samples, size=n_batch)
    batch = _gather_tiles(ixs)[..., None]
    batch_cond = np.sum(batch, axis=1)                     # daily sum = the condition
    batch = batch / batch_cond[:, None]                     # fractions of the daily sum (reference :162-163)
    batch_cond = _add_extra_condition(batch_cond / norm_scale, indices_all[ixs])
    assert batch.shape == (n_batch, nhours, ndomain, ndomain, 1)
    assert batch_cond.shape == (n_batch, ndomain, ndomain, n_channel)
    assert not np.any(np.isnan(batch)) and not np.any(np.isnan(batch_cond))
    assert np.max(batch) <= 1 and np.min(batch) >= 0
    return batch.astype(np.float32), batch_cond.astype(np.float32)


def generate_real_samples(n_batch):
    """reference :143-174 (a generator yielding [batch, batch_cond])."""
    while True:
        yield list(_real_batch(n_batch))


def generate_latent_points(n_batch):
    """reference :177-193: latent ~ N(0,1) and the normalised daily sums of random real tiles."""
    latent = np.random.normal(size=(n_batch, latent_dim))
    ixs = np.random.randint(0, n_samples, size=n_batch)
    batch_cond = _add_extra_condition(np.sum(_gather_tiles(ixs)[..., None], axis=1) / norm_scale, indices_all[ixs])
    assert batch_cond.shape == (n_batch, ndomain, ndomain, n_channel) and not np.any(np.isnan(batch_cond))
    return [latent, batch_cond.astype(np.float32)]


def generate_latent_points_as_generator(n_batch):
    while True:
        yield generate_latent_points(n_batch)


def generate_fake_samples(n_batch):
    latent, cond = generate_latent_points(n_batch)
    return [generator.predict([latent, cond]), cond]


def generate(cond):
    latent = np.random.normal(size=(1, latent_dim))
    return generator.predict([latent, np.expand_dims(cond, 0)])


# ---- layers / losses of the reference, as callables on numpy arrays (the fused HIP kernels are what
# train() and predict() run; these exist so user code that names them keeps working)
def wasserstein_loss(y_true, y_pred):
    """reference :215-216."""
    return np.mean(np.asarray(y_true) * np.asarray(y_pred))


class PixelNormalization:
    """reference :249-270: x / sqrt(mean(x**2, axis=-1, keepdims=True) + 1e-8)."""

    def __call__(self, inputs):
        return self.call(inputs)

    def call(self, inputs):
        x = np.asarray(inputs)
        return x / np.sqrt(np.mean(x ** 2.0, axis=-1, keepdims=True) + 1.0e-8)

    def compute_output_shape(self, input_shape):
        return input_shape


class RandomWeightedAverage:
    """reference :219-227: alpha*real + (1-alpha)*fake with alpha ~ U[0,1) per sample."""

    def __call__(self, inputs):
        return self.call(inputs)

    def call(self, inputs, **kwargs):
        a, b = np.asarray(inputs[0]), np.asarray(inputs[1])
        alpha = np.random.uniform(size=(a.shape[0], 1, 1, 1, 1))
        return alpha * a + (1 - alpha) * b

    def compute_output_shape(self, input_shape):
        return input_shape[0]


class GradientPenalty:
    """reference :230-244: ||d target / d wrt||_2 - 1 per sample.  In the reference this layer makes
    TensorFlow differentiate through a gradient; here that double backward is the explicit dgrad /
    second-forward sweep of rdgan_critic_grad.  As a callable it takes the gradient itself."""

    def __call__(self, grad):
        g = np.asarray(grad)
        return np.sqrt(np.sum(g.reshape(g.shape[0], -1) ** 2, axis=1, keepdims=True)) - 1

    def compute_output_shape(self, input_shapes):
        return (input_shapes[1][0], 1)


# ---- networks (reference :272-357)
def create_generator(seed=None):
    """RandomNormal(stddev=0.02) kernels, zero biases (reference :315)."""
    rng = np.random.default_rng(seed)
    return models.Generator(W.init_generator(rng, ndomain, n_channel), ndomain, n_channel)


def create_discriminator(seed=None):
    """Keras-default glorot_uniform kernels, zero biases (reference :286-304)."""
    rng = np.random.default_rng(seed)
    return models.Critic(W.init_critic(rng, ndomain, n_channel), ndomain, n_channel)


def build_networks(seed=None):
    """reference :360-409: generator, critic and the two compiled training graphs (here: the trainer)."""
    global generator, critic, _trainer
    generator = create_generator(seed)
    critic = create_discriminator(None if seed is None else seed + 1)
    _trainer = None
    return generator, critic


def _get_trainer(per_rank_batch):
    global _trainer, resume_from
    import torch
    import torch.distributed as dist
    from .trainer import WGANGPTrainer
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    eng = models.get_engine(ndomain, per_rank_batch, n_channel)
    if _trainer is None or _trainer.eng is not eng:
        _trainer = WGANGPTrainer(eng, generator.get_weights(), critic.get_weights(), n_disc=n_disc,
                                 world_size=world, rank=rank, process_group=dist.group.WORLD if world > 1 else None,
                                 exchange=exchange)
        generator.adopt_slab(_trainer.gparams)
        critic.adopt_slab(_trainer.dparams)
        if resume_from:
            _trainer.load_checkpoint(resume_from)
            resume_from = None
    return _trainer, world, rank


def resume(path):
    """Continue a run from the checkpoint _end_of_epoch writes (weights, Adam state, shared step counter, RNG
    positions): the next train(..., start_epoch=<epochs done>) call continues bit-identically.  The reference has no
    counterpart -- it saves weights only (:520-521) and restarts Adam from zero."""
    global resume_from, _trainer
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    resume_from, _trainer = path, None


def train(n_epochs, _batch_size, start_epoch=0, make_plots=False, max_batches_per_epoch=None, save_models=True):
    """reference :431-521: train with a fixed batch size for ``n_epochs``; per iteration n_disc critic
    steps then one generator step; prints the losses, raises ValueError on NaN, appends to ``hist`` and
    after each epoch writes hist.csv and saves gen_/disc_{params}_{epoch:04d}.h5 (Keras weight layout)."""
    global batch_size
    import torch
    from .trainer import shard_slice
    if generator is None or critic is None:
        build_networks()
    if data is None:
        raise RuntimeError("no training data: call load_data(...) or use_arrays(...) first")
    batch_size = _batch_size
    sample_gen = generate_real_samples(batch_size)
    gan_sample_gen = generate_latent_points_as_generator(batch_size)
    trainer, world, rank = _get_trainer(batch_size // max(1, _world_size()))
    sl = shard_slice(batch_size, world, rank)
    dev = trainer.eng.device
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl], dtype=np.float32)).to(dev)
    bat_per_epo = int(n_samples / batch_size)
    if max_batches_per_epoch:
        bat_per_epo = min(bat_per_epo, max_batches_per_epoch)
    for i in range(n_epochs):
        epoch = 1 + i + start_epoch
        for j in range(bat_per_epo):
            crit = []
            if device_dataset is not None:               # tiles gathered on the GPU, one shard per rank
                per = batch_size // world
                for _ in range(n_disc):
                    X_real, cond_real = device_dataset.sample_real(per)
                    latent = torch.from_numpy(np.random.normal(size=(per, latent_dim)).astype(np.float32)).to(dev)
                    crit.append((X_real, cond_real, latent))
                gen_batch = device_dataset.sample_latent(per, latent_dim)
                device_dataset.check_flags()
            else:
                for _ in range(n_disc):
                    X_real, cond_real = next(sample_gen)
                    latent = np.random.normal(size=(batch_size, latent_dim))
                    crit.append((to_dev(X_real), to_dev(cond_real), to_dev(latent)))
                latent, cond = next(gan_sample_gen)
                gen_batch = (to_dev(latent), to_dev(cond))
            d_loss, g_loss, bad = trainer.iteration(crit, gen_batch)
            d_loss, g_loss = float(d_loss), float(g_loss)
            if rank == 0:
                print(f'{epoch}, {j + 1}/{bat_per_epo}, d_loss {d_loss} g:{g_loss} ')
            if np.isnan(g_loss) or np.isnan(d_loss) or float(bad) != 0:
                raise ValueError('encountered nan in g_loss and/or d_loss')          # reference :487-488
            hist['d_loss'].append(d_loss)
            hist['g_loss'].append(g_loss)
        # every rank: the sharded exchange keeps the Adam second moments current only where they are owned, and gathering
        # them for the checkpoint is a collective -- it has to run in front of the rank gate, not behind it
        trainer.sync_state()
        if rank == 0:
            _end_of_epoch(epoch, make_plots, save_models)
    return hist


def _world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _end_of_epoch(epoch, make_plots, save_models):
    import pandas as pd
    os.makedirs(plotdir, exist_ok=True)
    os.makedirs(outdir, exist_ok=True)
    pd.DataFrame(hist).to_csv('hist.csv')                                            # reference :517
    if make_plots:
        _plot_epoch(epoch)
    if save_models:
        ext = 'h5'
        generator.save(f'{outdir}/gen_{params}_{epoch:04d}.{ext}')                   # reference :520-521
        critic.save(f'{outdir}/disc_{params}_{epoch:04d}.{ext}')
        if _trainer is not None:
            _trainer.save_checkpoint(f'{outdir}/checkpoint_{params}.npz', extra={'epoch': epoch}, synced=True)


def _plot_epoch(epoch, n_plot=30):
    """reference :495-516: fake-sample grid and loss curves."""
    import matplotlib
    matplotlib.use('agg')
    from matplotlib import pyplot as plt
    from matplotlib.colors import LogNorm
    X_fake, cond_fake = generate_fake_samples(n_plot)
    fig, axes = plt.subplots(n_plot, 25, figsize=(25, 25), squeeze=False)
    for r in range(n_plot):
        axes[r, 0].imshow(cond_fake[r].squeeze(), cmap=plt.cm.gist_earth_r, norm=LogNorm(vmin=0.01, vmax=1))
        for h in range(1, 24):
            axes[r, h].imshow(X_fake[r, h].squeeze(), vmin=0, vmax=1, cmap=plt.cm.hot_r)
    for ax in axes.ravel():
        ax.set_axis_off()
    fig.suptitle(f'epoch {epoch:04d}')
    fig.savefig(f'{plotdir}/fake_samples_{params}_{epoch:04d}.{plot_format}')
    plt.close(fig)
    fig = plt.figure()
    plt.plot(hist['d_loss'], label='d_loss')
    plt.plot(hist['g_loss'], label='g_loss')
    plt.ylabel('batch')
    plt.legend()
    fig.savefig(f'{plotdir}/training_loss_{params}.{plot_format}')
    plt.close(fig)


def main():
    load_data()
    build_networks()
    start_epoch = 0
    for n_epochs, bs in n_epoch_and_batch_size_list:                                  # reference :526-529
        train(n_epochs, bs, start_epoch, make_plots=True)
        start_epoch += n_epochs


if __name__ == '__main__':
    main()
