"""MI355X-native (gfx950) cWGAN-GP hot path of RainDisaggGAN (sipposip/pr-disagg-radar-gan).

Host-side mirror of the reference's Python surface over hand-written HIP kernels:
  * ``raindisagg_gan_pretrained`` -- generate_scenarios / plot_scenarios (reference file of the same name)
  * ``gan_train_cwgangp_pixelnorm`` -- create_generator / create_discriminator / train (idem)
  * ``engine.Engine`` -- thin object over the C ABI of include/rdgan.h
"""
from .engine import Engine, require_gpu  # noqa: F401
from . import weights  # noqa: F401
