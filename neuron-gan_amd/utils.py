"""Training utilities on the hot path: the latent sampler (reference utils.py:54-92)."""
import torch

Latent_vecs_memo = {}


def sample_latent_vec(size: tuple, seed=None, mode='randn', device=torch.device('cpu')):
    """Latents are drawn on the CPU generator and moved afterwards, exactly like the reference (utils.py:69-91):
    'randn' = normal draws clamped to [-5, 5] and projected on the unit sphere; 'rand' = uniform in [-1, 1).
    With a seed, the global RNG state is saved/restored around the draw and the result is memoised."""
    device = torch.device(device)
    key = None
    if seed is not None:
        key = (size, mode, seed)
        if key in Latent_vecs_memo:
            return Latent_vecs_memo[key].to(device)
        rng_state = torch.get_rng_state()
        torch.manual_seed(seed)
    if mode == 'rand':
        z = 2 * torch.rand(*size, device='cpu') - 1
    elif mode == 'randn':
        z = torch.randn(*size, device='cpu').clamp(-5, 5)
        z = z / z.norm(p=2, dim=1, keepdim=True)
    else:
        raise ValueError('{} is not supported'.format(mode))
    if seed is not None:
        torch.set_rng_state(rng_state)
        Latent_vecs_memo[key] = z
    if device.type != 'cpu':
        z = z.to(device)
    return z


def sample_latent_vec_device(size: tuple, device, generator=None):
    """Same distribution drawn directly on the GPU (graph-capturable): used by the benchmark / fast training loop,
    where reproducing the CPU RNG stream is not required."""
    from . import ops
    z = torch.randn(*size, device=device, generator=generator)
    return ops.latent_normalize_(z, 5.0)          # clamp(-5, 5) and the projection on the unit sphere in one launch


# ---------------------------------------------------------------------------------------------------------------------
# Checkpoints: same dictionary layout as the reference's Checkpointer (utils.py:142-223), so files written
# by either side load in the other.  One optional extra key, 'optimizer_state', carries the fused Adam's moments and
# per-tensor step counts (the reference does not save optimiser state at all).
# ---------------------------------------------------------------------------------------------------------------------
import os  # noqa: E402

import numpy as np  # noqa: E402


def get_saved_attrs(model):
    """{name: value} of the attributes a net lists in `saved_attrs` (reference utils.py:124-129)."""
    return {a: getattr(model, a) for a in getattr(model, 'saved_attrs', [])}


def set_saved_attrs(model, saved_attrs_dict):
    for name, value in saved_attrs_dict.items():
        if not hasattr(model, name):
            raise ValueError('{} is not an attribute of {}', name, model)
        if name == 'alpha' and hasattr(model, '_set_alpha'):
            model._set_alpha(float(value))      # keeps the host mirror of the fade-in coefficient in step
        else:
            setattr(model, name, value)
    return saved_attrs_dict


def _numpy_allow_list():
    """What a reference checkpoint needs beyond tensors and plain containers: numpy arrays of plain numeric dtypes (the four loss
    series, utils.py:160-169).  Array reconstruction from a raw buffer executes nothing from the file."""
    core = getattr(np, "_core", None) or np.core
    return [np.ndarray, np.dtype, core.multiarray._reconstruct] + \
           [type(np.dtype(t)) for t in (np.float64, np.float32, np.float16, np.int64, np.int32, np.int16, np.int8, np.uint8, np.bool_)]


def load_checkpoint_dict(filename, device=torch.device('cpu')):
    """torch.load with the weights-only unpickler: checkpoint files come from users and from the reference's download link
    (gen_dis_default.pth, setup.py:79), so nothing in them may execute.  The reference's own `torch.load` (utils.py:185,
    models.py:397) is the full unpickler; the only non-tensor payload its checkpoints hold are numpy loss series, which are
    allow-listed explicitly.  A file that needs anything else is refused with torch's error."""
    with torch.serialization.safe_globals(_numpy_allow_list()):
        return torch.load(filename, map_location=device, weights_only=True)


class Checkpointer:
    def __init__(self, Generator_net, Discriminator_net, lr: float, filename: str, N_epochs=100, verbose=True,
                 device=torch.device('cpu'), extra_checkpoint_period=50e3, trainer=None):
        self.Generator_net = Generator_net
        self.Discriminator_net = Discriminator_net
        self.lr = lr
        self.filename = filename
        self.epoch = 0
        self.Loss_real = np.zeros(N_epochs)
        self.Loss_fake = np.zeros(N_epochs)
        self.Loss_G = np.zeros(N_epochs)
        self.Loss_D = np.zeros(N_epochs)
        self.verbose = verbose
        self.device = device
        self.extra_checkpoint_period = extra_checkpoint_period
        self.trainer = trainer      # optional PGGANTrainer: adds / restores 'optimizer_state'

    def save_state(self, epoch):
        self.epoch = epoch
        cpu = lambda sd: {k: v.detach().to('cpu').clone() for k, v in sd.items()}
        checkpoint_dict = {'epoch': self.epoch,
                           'Generator_state': cpu(self.Generator_net.state_dict()),
                           'Generator_attrs': {k: (v.detach().cpu() if torch.is_tensor(v) else v)
                                               for k, v in get_saved_attrs(self.Generator_net).items()},
                           'Discriminator_state': cpu(self.Discriminator_net.state_dict()),
                           'Discriminator_attrs': {k: (v.detach().cpu() if torch.is_tensor(v) else v)
                                                   for k, v in get_saved_attrs(self.Discriminator_net).items()},
                           'lr': self.lr,
                           'Loss_real': self.Loss_real[:epoch], 'Loss_fake': self.Loss_fake[:epoch],
                           'Loss_G': self.Loss_G[:epoch], 'Loss_D': self.Loss_D[:epoch]}
        if self.trainer is not None:
            checkpoint_dict['optimizer_state'] = self.trainer.optimizer_state()
        torch.save(checkpoint_dict, self.filename)
        if epoch % self.extra_checkpoint_period == 0:
            base, ext = os.path.splitext(self.filename)
            torch.save(checkpoint_dict, base + '_{:d}k'.format(int(epoch / 1000)) + ext)
        if self.verbose:
            print('Training state at epoch {} saved in {}.'.format(self.epoch, self.filename))

    def load_state(self, filename=None):
        """filename None: resume everything from self.filename; otherwise load only the networks from `filename`.
        (The reference reads the networks from self.filename in both cases, utils.py:213-215 -- a slip that this
        implementation does not reproduce.)"""
        source = self.filename if filename is None else filename
        checkpoint_dict = load_checkpoint_dict(source, self.device)
        if filename is None:
            self.epoch = checkpoint_dict['epoch']
            self.Loss_real[:self.epoch] = checkpoint_dict['Loss_real']
            self.Loss_fake[:self.epoch] = checkpoint_dict['Loss_fake']
            self.Loss_G[:self.epoch] = checkpoint_dict['Loss_G']
            self.Loss_D[:self.epoch] = checkpoint_dict['Loss_D']
        if 'Generator_attrs' in checkpoint_dict and 'Discriminator_attrs' in checkpoint_dict:
            gen_attrs = {k: v for k, v in checkpoint_dict['Generator_attrs'].items() if k in self.Generator_net.saved_attrs}
            dis_attrs = {k: v for k, v in checkpoint_dict['Discriminator_attrs'].items() if k in self.Discriminator_net.saved_attrs}
            if hasattr(self.Generator_net, 'set_resolution'):
                res, alpha = gen_attrs['image_size'], float(gen_attrs['alpha'])
                if self.Generator_net.image_size != res:
                    self.Generator_net.set_resolution(res, alpha)
                    self.Discriminator_net.set_resolution(res, alpha)
            set_saved_attrs(self.Generator_net, gen_attrs)
            set_saved_attrs(self.Discriminator_net, dis_attrs)
        gen_state = type(self.Generator_net).from_state_dict(source, verbose=False).state_dict()
        dis_state = type(self.Discriminator_net).from_state_dict(source, verbose=False).state_dict()
        self.Generator_net.load_state_dict(gen_state, strict=False)
        self.Discriminator_net.load_state_dict(dis_state, strict=False)
        if self.trainer is not None:
            self.trainer.refresh_stage()
            if filename is None and 'optimizer_state' in checkpoint_dict:
                self.trainer.load_optimizer_state(checkpoint_dict['optimizer_state'])
        if self.verbose:
            print(('Loaded training state from {}' if filename is None else 'Loaded weights from {}').format(source))


# ---------------------------------------------------------------------------------------------------------------------
# Sampling: reference utils.py:346-355 (gen_samples), 568-610 (plot_gen_samples)
# ---------------------------------------------------------------------------------------------------------------------
def gen_samples(Generator, N_images=16, seed=None):
    device = next(Generator.parameters()).device
    z_latent = sample_latent_vec((N_images, Generator.latent_dim), seed=seed, device=device)
    with torch.no_grad():
        images = Generator(z_latent).detach()
    return images, z_latent


def make_image_grid(images: torch.Tensor, nrow: int, normalize=True, padding=2) -> torch.Tensor:
    """(N, C, H, W) -> (C, H', W') grid, the subset of torchvision.utils.make_grid the reference uses (utils.py:21, 608)."""
    images = images.detach().float().cpu()
    if normalize:
        lo, hi = float(images.min()), float(images.max())
        images = (images - lo) / max(hi - lo, 1e-5)
    n, c, h, w = images.shape
    ncol = min(nrow, n)
    nrows = (n + ncol - 1) // ncol
    grid = torch.zeros(c, nrows * (h + padding) + padding, ncol * (w + padding) + padding)
    for i in range(n):
        r, col = divmod(i, ncol)
        grid[:, padding + r * (h + padding):padding + r * (h + padding) + h,
             padding + col * (w + padding):padding + col * (w + padding) + w] = images[i]
    return grid


def plot_gen_samples(Generator, eval_noise=None, N_images=16, seed=None, filename=None):
    """Generate a grid of samples: eval mode, seeded + memoised latents, low-resolution samples enlarged to the final size
    with nearest-neighbour interpolation (utils.py:598-601), written as a PNG when `filename` is given.  Returns the grid."""
    was_training = Generator.training
    Generator.train(False)
    if eval_noise is None:
        images, _ = gen_samples(Generator, N_images, seed=seed)
    else:
        with torch.no_grad():
            images = Generator(eval_noise).detach()
        N_images = images.size(0)
    Generator.train(was_training)
    images = images.cpu()
    if images.size(-1) != Generator.image_size_max:
        images = torch.nn.functional.interpolate(images, size=(Generator.image_size_max, Generator.image_size_max))
    grid = make_image_grid(images, nrow=int(np.round(np.sqrt(N_images))))
    if filename is not None:
        from PIL import Image
        arr = (grid.clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).numpy()
        Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr).save(filename)
    return grid


def Calculate_D_steps(Loss_real, Loss_fake, N_min, N_max, Period):
    """Number of critic steps for the next epoch (reference utils.py:105-120): the critic is trained less when the gap between
    the real and fake scores is large compared to the spread of the real score over the last `Period` epochs."""
    if Loss_real and Loss_fake:
        real_std = np.std(Loss_real[-Period:])
        diff = np.mean(np.abs(np.subtract(Loss_fake[-Period:], Loss_real[-Period:])))
        n_steps = np.round(real_std / diff * N_max)
        n_steps = np.min([n_steps, N_max])
        n_steps = np.max([n_steps, N_min])
        return int(n_steps)
    return N_max


def similarity_loss(images_batch, Z_batch, Lambda=1.0):
    """Reference loss_functions.py:185-205: squared difference between the pairwise cosine similarities of the latents and of the
    images.  The reference evaluates it on the REAL batch and the latents (train.py:380), so it carries no gradient to either net:
    it is a monitored number, computed here with a few tiny torch matmuls (B x B), off the hot path."""
    b = images_batch.size(0)
    im = images_batch.reshape(b, -1)
    z = Z_batch.reshape(b, -1)
    im = im / im.norm(2, dim=1, keepdim=True)
    z = z / z.norm(2, dim=1, keepdim=True)
    return Lambda * torch.pow(z @ z.t() - im @ im.t(), 2).sum() / (b * (b - 1))
