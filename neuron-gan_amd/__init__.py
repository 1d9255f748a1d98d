"""neuron-gan_amd: MI355X-native PGGAN / WGAN-GP training step behind the neuron-gan API.

The directory name carries a hyphen, so it is loaded under the module name `neuron_gan_amd`
(see `__graft_entry__.load_package`).  Layout:
    csrc/              hand-written gfx950 kernels + the C ABI of include/ngan.h  -> libngan_hip.so
    _C.py              ctypes binding (fails loudly if the library is missing; no CPU fallback)
    ops.py             differentiable operators closed under double-backward
    models.py          Generator_PG / Discriminator_PG with the reference's surface and state_dict keys
    loss_functions.py  D_W_loss / G_W_loss / D_grad_pen_loss
    utils.py           sample_latent_vec
    configs/config.py  module-as-singleton configuration
    train.py           the G/D step driver (flat parameters, fused Adam, data-parallel gradient exchange), epoch driver, CLI
    data.py            device-resident dataset with the reference's augmentation chain as one launch per batch
    workmodel.py       algorithmic FLOP / byte model of an iteration (what bench.py's roofline figures divide by)
"""
from . import _C, ops, utils, models, loss_functions, train, data, workmodel  # noqa: F401
from .configs import config  # noqa: F401

__version__ = "0.1.0"
