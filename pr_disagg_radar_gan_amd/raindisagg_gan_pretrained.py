"""Drop-in for the reference's ``raindisagg_gan_pretrained.py`` (the public inference API) on the
MI355X-native engine:

    from pr_disagg_radar_gan_amd.raindisagg_gan_pretrained import generate_scenarios, plot_scenarios

Same names, argument meaning, return types and module globals (``norm_scale``,
``generator_file``, ``latent_dim``, ``gen``).  Differences, on purpose: the generator file is
loaded on first use instead of at import (the reference loads at import, :43), a ``.npz`` weight
file is accepted beside Keras ``.h5``, and a missing file raises ``FileNotFoundError`` naming it.
"""
import numpy as np

from . import models
from . import weights as W

norm_scale = W.NORM_SCALE          # reference :13
generator_file = 'trained_models/gen_20090101-20161231-tp_thresh_daily5_n_thresh20_ndomain16_stride16_0020.h5'   # :14
latent_dim = W.LATENT_DIM          # reference :47 derives it from the loaded model's first input


class _LazyGenerator:
    """Stands in for the module-level ``gen`` of the reference (:43): resolves ``generator_file`` on
    first attribute access."""

    def __init__(self):
        object.__setattr__(self, "_model", None)

    def _resolve(self):
        if self._model is None:
            object.__setattr__(self, "_model", models.load_generator(generator_file))
        return self._model

    def __getattr__(self, name):
        return getattr(self._resolve(), name)


gen = _LazyGenerator()


def set_generator(model_or_path):
    """Use another generator (a ``models.Generator`` or a weight file path) for generate_scenarios."""
    global gen, generator_file
    if isinstance(model_or_path, str):
        generator_file = model_or_path
        gen = models.load_generator(model_or_path)
    else:
        gen = model_or_path
    return gen


def generate_scenarios(cond, n_scenarios):
    """reference :52-65.  cond: ndarray (ndomain, ndomain, 1), daily sum in mm/day (un-normalised).
    Returns ndarray (n_scenarios, 24, ndomain, ndomain) in mm/h; every scenario sums to ``cond`` over
    the 24 hours.  Uses the global numpy RNG for the latent noise, like the reference (:56)."""
    # the generator takes normalized daily sums, so we have to divide by norm_scale
    cond = np.asarray(cond) / norm_scale
    latent = np.random.normal(size=(n_scenarios, latent_dim))
    cond_batch = np.repeat(cond[np.newaxis], repeats=n_scenarios, axis=0)
    generated = gen.predict([latent, cond_batch])
    generated = generated.squeeze()          # also drops the batch axis for n_scenarios == 1, as the reference does
    return generated * cond.squeeze() * norm_scale


def plot_scenarios(scenarios):
    """reference :68-90: one row per scenario, 24 hourly panels, LogNorm(0.01, 50), gist_earth_r.
    Keeps the reference's indexing ``scenarios[iplot, jplot - 1]`` (the column labelled 00:00 shows
    hour index -1, i.e. the last hour) so figures are identical to the reference's."""
    from matplotlib import pyplot as plt
    from matplotlib.colors import LogNorm

    scenarios = np.asarray(scenarios)
    nrows = len(scenarios)
    fig, axes = plt.subplots(nrows, 24, figsize=(24, nrows), squeeze=False)
    norm = LogNorm(vmin=0.01, vmax=50)
    image = None
    for (irow, hour), ax in np.ndenumerate(axes):
        image = ax.imshow(scenarios[irow, hour - 1], cmap=plt.cm.gist_earth_r, norm=norm)
        ax.set_axis_off()
        if irow == 0:
            ax.annotate(f'{hour:02d}:00', xy=(0.5, 1), xytext=(0, 5), xycoords='axes fraction',
                        textcoords='offset points', size='large', ha='center', va='baseline')
    fig.subplots_adjust(right=0.93)
    colorbar = fig.colorbar(image, cax=fig.add_axes([0.93, 0.15, 0.007, 0.7]))
    colorbar.set_label('fraction of daily precipitation', fontsize=16)
    colorbar.ax.tick_params(labelsize=16)
    return fig
