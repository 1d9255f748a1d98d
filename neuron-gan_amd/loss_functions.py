"""WGAN / WGAN-GP losses with the reference's interface (loss_functions.py:7-74, 148-180), running on the HIP ops.

    D_W_loss(G, D, drift_epsilon)(real)      -> (loss, score_real, score_fake)
    G_W_loss(G, D)(real)                     -> (loss, z)
    D_grad_pen_loss(G, D, Lambda)(real)      -> loss   (create_graph double-backward through the critic)

Latents come from `utils.sample_latent_vec` unless a tensor is passed through the optional `z=` / `epsilon=`
keywords (how the parity tests inject the reference's draws).  NaN handling: the reference dumps locals and
raises `ValueError` (loss_functions.py:35-41, 70-72); here the check is optional (`check_nan`) because
`torch.isnan(...)` in an `if` is a host sync on the GPU -- the training loop checks once per epoch instead.
"""
import torch
import torch.nn as nn

from . import ops
from .utils import sample_latent_vec


def _latents(net, batch, device, z):
    if z is not None:
        return z
    return sample_latent_vec((batch, net.latent_dim), device=device)


class D_W_loss(nn.Module):
    def __init__(self, generator_net, discriminator_net, drift_epsilon=0.0, check_nan=True):
        super().__init__()
        self.generator_net = generator_net
        self.discriminator_net = discriminator_net
        self.drift_epsilon = drift_epsilon
        self.check_nan = check_nan

    def forward(self, real_images, z=None, fake_images=None):
        batch_size, device = real_images.size(0), real_images.device
        if fake_images is None:
            z = _latents(self.generator_net, batch_size, device, z)
            with torch.no_grad():
                fake_images = self.generator_net(z)
        # D(real) and D(fake) share the weights and no op couples samples, so they run as ONE critic pass over the
        # concatenated batch (the reference makes two calls, loss_functions.py:21, 29; per-sample results are identical)
        with ops.first_order_only():      # differentiated once (train.py:365): fused PixelNorm-backward epilogues apply
            scores = self.discriminator_net(torch.cat([real_images, fake_images], dim=0))
        # -mean(real) + mean(fake) + drift * mean(real^2) (loss_functions.py:22, 29, 33, 45) as one launch each way
        D_loss, score_real, score_fake = ops.WLossHead.apply(scores, batch_size, float(self.drift_epsilon) if self.drift_epsilon > 0 else 0.0)
        if self.check_nan:
            if torch.isnan(score_real):
                raise ValueError('Real loss is nan.')
            if torch.isnan(score_fake):
                raise ValueError('Fake loss is nan.')
        return D_loss, score_real, score_fake


class G_W_loss(nn.Module):
    def __init__(self, generator_net, discriminator_net, check_nan=True):
        super().__init__()
        self.generator_net = generator_net
        self.discriminator_net = discriminator_net
        self.check_nan = check_nan

    def forward(self, real_images_batch, z=None):
        batch_size, device = real_images_batch.size(0), real_images_batch.device
        z_latent = _latents(self.generator_net, batch_size, device, z)
        fake_images = self.generator_net(z_latent)
        with ops.first_order_only():      # differentiated once (train.py:384)
            G_loss = ops.WLossHead.apply(self.discriminator_net(fake_images), batch_size, 0.0)[0]       # -mean(D(G(z)))
        if self.check_nan and torch.isnan(G_loss):
            raise ValueError('Generator loss is nan.')
        return G_loss, z_latent


class D_grad_pen_loss(nn.Module):
    def __init__(self, generator_net, discriminator_net, Lambda):
        super().__init__()
        self.generator_net = generator_net
        self.discriminator_net = discriminator_net
        self.Lambda = Lambda
        self.last_grad_norms = None  # per-sample |grad D| of the last call (monitoring / parity tests)
        self._ones_cache = None

    def _ones(self, like):
        if self._ones_cache is None or self._ones_cache.shape != like.shape or self._ones_cache.device != like.device:
            self._ones_cache = torch.ones_like(like)
        return self._ones_cache

    def forward(self, real_images, z=None, epsilon=None, x_tilde=None):
        if not self.Lambda > 0:
            # the reference returns the integer CPU scalar torch.tensor(0) here (loss_functions.py:179), which only survives being
            # stacked / accumulated with device tensors by accident; same value, on the images' device
            return torch.zeros((), device=real_images.device)
        batch_size, device = real_images.size(0), real_images.device
        if x_tilde is None:
            z_latent = _latents(self.generator_net, batch_size, device, z)
            with torch.no_grad():
                x_tilde = self.generator_net(z_latent)
        if epsilon is None:
            epsilon = torch.rand((batch_size, 1, 1, 1), device=device)
        x_hat = ops.xhat(real_images, x_tilde, epsilon)
        x_hat.requires_grad_()
        output = self.discriminator_net(x_hat)
        # d(sum of the scores)/d(x_hat) (loss_functions.py:175): grad_outputs = ones instead of a sum node
        # (inputs_only: x_hat is the only input asked for, so the critic's weight gradients of this pass would be thrown away)
        with ops.inputs_only():
            Disc_grad = torch.autograd.grad(outputs=output, inputs=x_hat, grad_outputs=self._ones(output), create_graph=True)[0]
        penalty, norms = ops.GradPenaltyHead.apply(Disc_grad, float(self.Lambda))
        self.last_grad_norms = norms.detach()
        return penalty
