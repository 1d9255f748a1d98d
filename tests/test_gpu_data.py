"""Device input pipeline (neuron-gan_amd/data.py, csrc/augment.hip) against a plain-torch CPU restatement of the torchvision
tensor code path the reference's transforms run (data/NeuronDataset.py:112-126, 149-164).  torchvision is not installed here, so
this pins the kernel to the RESTATEMENT (affine_grid-style centred grid + grid_sample(nearest, zeros, align_corners=False),
blend-and-clamp colour ops, aten's antialiased bilinear resize), not to torchvision itself: parity with torchvision is unpinned.
Tolerance: nearest-neighbour sampling can flip a source pixel where a rotated coordinate lands within float rounding of .5, so up
to 1e-3 of the pixels may differ; everything else agrees to 2e-5."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def ref_augment(img, angle, tx, ty, flip, brightness, contrast, contrast_first, R, S):
    """img (P, P) in [0, 1] -> (S, S) in [-1, 1]"""
    P = img.shape[-1]
    rot = math.radians(angle)
    c, s = math.cos(rot), math.sin(rot)
    # torchvision _get_inverse_affine_matrix(center 0, angle, translate, scale 1, shear 0): [cos, sin, .; -sin, cos, .] with the
    # translation folded in, then _gen_affine_grid (pixel-centre base grid, theta rescaled by half the size)
    theta = torch.tensor([[c, s, c * -tx + s * -ty], [-s, c, -s * -tx + c * -ty]], dtype=torch.float64)
    xs = torch.linspace(-P * 0.5 + 0.5, P * 0.5 - 0.5, P, dtype=torch.float64)
    base = torch.stack([xs.view(1, P).expand(P, P), xs.view(P, 1).expand(P, P), torch.ones(P, P, dtype=torch.float64)], dim=-1)
    grid = (base.view(-1, 3) @ (theta.t() / torch.tensor([0.5 * P, 0.5 * P], dtype=torch.float64))).view(1, P, P, 2)
    out = F.grid_sample(img.double()[None, None], grid, mode="nearest", padding_mode="zeros", align_corners=False)[0, 0].float()
    if flip:
        out = torch.flip(out, dims=[0])
    ops = [lambda v: (v * brightness).clamp(0, 1), lambda v: (contrast * v + (1 - contrast) * v.mean()).clamp(0, 1)]
    if contrast_first:
        ops = ops[::-1]
    for op in ops:
        out = op(out)
    top = int(round((P - R) / 2.0))
    out = out[top:top + R, top:top + R] * 2 - 1
    if S != R:
        out = F.interpolate(out[None, None], size=(S, S), mode="bilinear", antialias=True, align_corners=False)[0, 0]
    return out


@pytest.mark.parametrize("S", [64, 32, 8])
def test_augment_batch_matches_restatement(ngan, S):
    torch.manual_seed(4)
    R, N = 64, 5
    imgs = torch.rand(N, 1, R, R) ** 2
    ds = ngan.data.NeuronDataset(imgs, augmentations=True, im_translation=0.1, device="cuda:0", seed=7)
    ds.set_image_size(S)
    idx = [0, 3, 4, 1, 1, 2, 0]
    p = ds.draw_params(len(idx))
    p["angle"][0], p["tx"][0], p["ty"][0] = 0.0, 0.0, 0.0            # one identity geometry: must be exact
    p["angle"][1] = 90.0
    got = ds.batch(idx, params=p).cpu()
    assert got.shape == (len(idx), 1, S, S)
    padded = ds.images.cpu()
    for k, i in enumerate(idx):
        want = ref_augment(padded[i], float(p["angle"][k]), float(p["tx"][k]), float(p["ty"][k]), int(p["flip"][k]),
                           float(p["brightness"][k]), float(p["contrast"][k]), int(p["contrast_first"][k]), R, S)
        diff = (got[k, 0] - want).abs()
        tol = 2e-5 if S == R else 2e-3        # a flipped nearest-neighbour source pixel is diluted by the resize filter
        assert float((diff > tol).float().mean()) <= 1e-3, (k, float(diff.max()))
        assert abs(float(got[k, 0].mean()) - float(want.mean())) < 2e-4


def test_no_augmentation_is_crop_renormalise_resize(ngan):
    torch.manual_seed(5)
    imgs = torch.rand(3, 32, 32)
    ds = ngan.data.NeuronDataset(imgs, augmentations=False, device="cuda:0")
    full = ds.batch([2, 0]).cpu()
    assert torch.allclose(full[:, 0], imgs[[2, 0]] * 2 - 1, atol=1e-6)
    ds.set_image_size(16)
    half = ds.batch([1]).cpu()
    want = F.interpolate((imgs[1] * 2 - 1)[None, None], size=(16, 16), mode="bilinear", antialias=True, align_corners=False)
    assert torch.allclose(half, want, atol=1e-5)
    it = ngan.data.DatasetIterator(ds, batch_size=2)
    sizes = [b.shape[0] for b in it]
    assert sizes == [2, 1]                                             # the reference iterator's short last batch
    with pytest.raises(IndexError):
        ds.batch([3])


def test_epoch_driver_on_the_device_dataset(ngan):
    """train.pggan_train fed by data.NeuronDataset: augmented batches at the stage's resolution across a growth event, an
    adaptive critic schedule (reference utils.Calculate_D_steps) and the similarity-loss monitor switched on."""
    import types

    import numpy as np
    models, train = ngan.models, ngan.train
    cfg = types.SimpleNamespace(adapt_critic=True, sim_loss_lambda=0.5, sim_loss_lambda_decay_rate=0.1, n_critic=2, batch_size=4,
                                transit_sch=[2], N_epochs=4, alpha_step=0.5, learning_rate=1e-3, checkpointing_period=100, ID="t002")
    torch.manual_seed(6)
    G = models.Generator_PG([32, 16], image_size_init=8, latent_dim=32).to("cuda:0")
    D = models.Discriminator_PG([16, 32], image_size_init=8).to("cuda:0")
    data = ngan.data.NeuronDataset(torch.rand(6, 1, 16, 16), augmentations=True, im_translation=0.1, device="cuda:0", seed=1,
                                   noise_mean=[0.05] * 6, noise_std=[0.01] * 6)
    seen = []
    orig = data.batch
    data.batch = lambda idx, params=None: (lambda out: (seen.append(tuple(out.shape)), out)[1])(orig(idx, params))
    tr = train.PGGANTrainer(G, D, learning_rate=cfg.learning_rate, alpha_step=cfg.alpha_step, n_critic=cfg.n_critic, device_latents=True)
    series = train.pggan_train(tr, data, cfg, log=lambda *_: None)
    assert all(len(v) == 4 and np.isfinite(v).all() for v in series.values())
    assert G.image_size == 16 and data.image_size == 16
    assert (4, 1, 8, 8) in seen and (4, 1, 16, 16) in seen and (2, 1, 16, 16) in seen     # stage sizes and the short last batch
    assert float(data.images.min()) > 0.0                                                   # noise fill replaced the zero padding
