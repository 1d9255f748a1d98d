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
    z = torch.randn(*size, device=device, generator=generator).clamp_(-5, 5)
    return z / z.norm(p=2, dim=1, keepdim=True)
