"""Device-resident dataset with the reference's augmentation chain.

Restates /root/reference/data/NeuronDataset.py for the HIP path: `NeuronDataset` keeps every image on the GPU (the reference's
`load_all` mode, which its on-device `DatasetIterator` requires, NeuronDataset.py:171-175) already padded by image_size // 4
(NeuronDataset.py:70-71) and, when noise statistics are given, with the zero pixels replaced by Gaussian noise (13-21).  A batch
is produced by ONE call of `ngan_augment_batch` (csrc/augment.hip): RandomAffine(degrees=180, translate=(t, t), nearest, fill 0),
RandomVerticalFlip, ColorJitter(brightness 0.25, contrast 0.25, random order), CenterCrop, Renormalize((0,1) -> (-1,1)) and the
antialiased Resize to the current stage's resolution (112-126, 149-164) -- instead of the reference's per-image Python loop
(NeuronDataset.py:199-207), which could not feed a step that consumes ~2 000 images per second.

Not restated: reading PNGs with PIL and the multi-Otsu estimate of the noise statistics (skimage is not installed; pass
`noise_mean` / `noise_std` to use the noise fill).  torchvision itself is absent from this image, so the parity tests compare with
a plain-torch CPU restatement of its tensor code path (tests/test_gpu_data.py): parity with torchvision is unpinned.
"""
import math

import torch

from . import _C


def _as_images(images):
    if images.dim() == 3:
        images = images.unsqueeze(1)
    if images.dim() != 4 or images.shape[1] != 1 or images.shape[2] != images.shape[3]:
        raise ValueError(f"expected square single-colour images (N, 1, R, R) or (N, R, R), got {tuple(images.shape)}")
    return images.float()


class NeuronDataset:
    def __init__(self, images, image_size=None, augmentations=True, im_translation=0.0, device="cuda", noise_mean=None,
                 noise_std=None, seed=None):
        images = _as_images(images)
        n, _, r, _ = images.shape
        self.image_size = self.image_size_max = int(image_size or r)
        if self.image_size_max != r:
            raise ValueError(f"images are {r} pixels wide, image_size is {self.image_size_max}")
        self.augmentations = bool(augmentations)
        self.im_translation = float(im_translation)
        self.device = torch.device(device)
        self.load_all = True
        pad = r // 4                                                            # NeuronDataset.py:70
        self.canvas = r + 2 * pad
        padded = torch.zeros(n, self.canvas, self.canvas)
        padded[:, pad:pad + r, pad:pad + r] = images[:, 0]
        self.gen = torch.Generator(device="cpu")
        if seed is not None:
            self.gen.manual_seed(int(seed))
        if noise_mean is not None:                                              # replace_zero_with_noise, NeuronDataset.py:13-21
            mean = torch.as_tensor(noise_mean, dtype=torch.float32).reshape(-1, 1, 1)
            std = torch.as_tensor(noise_std, dtype=torch.float32).reshape(-1, 1, 1)
            noise = torch.randn(padded.shape, generator=self.gen) * std + mean
            padded = torch.where(padded == 0, noise, padded)
        self.images = padded.to(self.device).contiguous()
        self._ws = None

    def __len__(self):
        return self.images.shape[0]

    def set_image_size(self, size: int):
        assert size <= self.image_size_max, 'The image size ({}) must be < {}.'.format(size, self.image_size_max)
        assert self.image_size_max % size == 0, 'The image size ({}) must divide {}.'.format(size, self.image_size_max)
        self.image_size = int(size)

    # ---- random draws of one batch (torchvision's distributions: RandomAffine.get_params, RandomVerticalFlip, ColorJitter) ----
    def draw_params(self, batch):
        g = self.gen
        if not self.augmentations:
            z, o = torch.zeros(batch), torch.ones(batch)
            return dict(angle=z, tx=z, ty=z, flip=z.int(), brightness=o, contrast=o, contrast_first=z.int())
        u = lambda lo, hi: torch.empty(batch).uniform_(lo, hi, generator=g)
        max_d = self.im_translation * self.canvas
        return dict(angle=u(-180.0, 180.0), tx=torch.round(u(-max_d, max_d)) if max_d > 0 else torch.zeros(batch),
                    ty=torch.round(u(-max_d, max_d)) if max_d > 0 else torch.zeros(batch),
                    flip=(torch.rand(batch, generator=g) < 0.5).int(),
                    brightness=u(0.75, 1.25), contrast=u(0.75, 1.25),
                    contrast_first=(torch.rand(batch, generator=g) < 0.5).int())

    @staticmethod
    def pack_params(p):
        """host dict of per-sample tensors -> the (B, 8) record array of include/ngan.h"""
        b = p["angle"].shape[0]
        rad = p["angle"].double() * (math.pi / 180.0)
        rec = torch.zeros(b, 8, dtype=torch.float32)
        rec[:, 0], rec[:, 1] = torch.cos(rad).float(), torch.sin(rad).float()
        rec[:, 2], rec[:, 3] = p["tx"].float(), p["ty"].float()
        rec[:, 4], rec[:, 5] = p["brightness"].float(), p["contrast"].float()
        ints = rec.view(torch.int32)
        ints[:, 6], ints[:, 7] = p["flip"].int(), p["contrast_first"].int()
        return rec

    def batch(self, indices, params=None):
        """(B, 1, S, S) augmented images in [-1, 1] at the current stage resolution for the given image indices."""
        idx = torch.as_tensor(indices, dtype=torch.int32)
        b = idx.numel()
        if b == 0:
            raise ValueError("empty batch")
        if int(idx.min()) < 0 or int(idx.max()) >= len(self):
            raise IndexError("image index out of range")
        rec = self.pack_params(params if params is not None else self.draw_params(b)).to(self.device, non_blocking=True)
        idx = idx.to(self.device, non_blocking=True)
        need = _C.lib().ngan_augment_workspace_bytes(b, self.canvas) // 4
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.device, dtype=torch.float32)
        s = self.image_size
        out = torch.empty(b, 1, s, s, device=self.device, dtype=torch.float32)
        _C.call("ngan_augment_batch", self.images, idx, rec, self._ws, out, len(self), b, self.canvas, self.image_size_max, s)
        return out

    def __getitem__(self, i):
        return self.batch([int(i)])[0]


class DatasetIterator:
    """Sequential batches over a device dataset (reference NeuronDataset.py:170-207): the last batch may be short."""

    def __init__(self, dataset: NeuronDataset, batch_size: int, device=None):
        if not dataset.load_all:
            raise Exception('On-device iteration is only possible when all images are loaded.')
        self.dataset = dataset
        self.N_images = len(dataset)
        self.batch_size = batch_size
        self.device = dataset.device
        self.image_ind = 0

    def __iter__(self):
        self.image_ind = 0
        return self

    def __next__(self):
        if self.image_ind < self.N_images:
            last = min(self.image_ind + self.batch_size, self.N_images)
            out = self.dataset.batch(list(range(self.image_ind, last)))
            self.image_ind = last
            return out
        raise StopIteration
