"""Algorithmic work of the PGGAN / WGAN-GP training iteration: the numbers bench.py divides by.

A net at a stage is written out as its list of contraction layers (the reference's module order: models.py:295-316, 468-490 and
the merged / fading `Conv2d_scale_block`s, models.py:245-268); the FLOP count of a pass is 2 x MAC summed over that list, the
minimum activation traffic is (input + output elements) summed over it with every resample / LeakyReLU / PixelNorm fused away.
One iteration (train.py:357-385, n_critic = 1) costs  W_alg = 5 F_G + 14 F_D  per image:
  generator: 2 detached forwards in the critic step (loss_functions.py:26, 167) + forward, input- and weight-gradient = 5;
  critic:    forward of real and fake (2) + their input- and weight-gradients (4) + the gradient penalty: forward, input gradient
             with graph, the backward of those two (conv(gg, w) and wgrad(gg, gy); input- and weight-gradient of the forward
             nodes) = 6, + forward and input gradient in the generator step (2; the reference's discarded critic weight
             gradients there are not counted) = 14.
"""
import math
from typing import List, NamedTuple, Tuple


class Layer(NamedTuple):
    name: str
    kind: str            # "linear", "conv3x3", "conv1x1", "valid"
    cin: int
    cout: int
    out_px: int          # output pixels per image (1 for the linear stem and the critic's full-extent conv)
    in_px: int           # pixels of the tensor read from HBM (1/4 of out_px behind a bilinear x2, 4x behind an avg-pool)
    taps: int

    @property
    def flops(self) -> float:
        return 2.0 * self.taps * self.cin * self.cout * self.out_px

    def io_elements(self, resample_fused: bool = False) -> float:
        """input + output elements of one pass.  resample_fused=False counts the input the contraction sees (after the block's
        resample: what hooks on the reference modules measure, SURVEY.md 8d's E); True counts the tensor a resample-on-load kernel
        actually reads from HBM."""
        seen = self.in_px if (resample_fused or self.kind != "conv3x3") else self.out_px
        return float(self.cin * seen + self.cout * self.out_px)


def _stage(res: int, image_size_init: int, alpha: float) -> Tuple[int, int]:
    n_up = int(round(math.log2(res / image_size_init)))
    if image_size_init * 2 ** n_up != res:
        raise ValueError(f"resolution {res} is not image_size_init * 2^n")
    return n_up, (n_up if alpha >= 1 else n_up - 1)


def generator_layers(widths: List[int], image_size_init: int, res: int, latent_dim: int, alpha: float = 1.0,
                     n_colors: int = 1) -> List[Layer]:
    n_up, merged = _stage(res, image_size_init, alpha)
    s = image_size_init
    out = [Layer("stem", "linear", latent_dim, widths[0] * s * s, 1, 1, 1),
           Layer("conv0", "conv3x3", widths[0], widths[0], s * s, s * s, 9)]
    for i in range(n_up):
        s *= 2
        out.append(Layer(f"block{i}.conv1", "conv3x3", widths[i], widths[i + 1], s * s, s * s // 4, 9))
        out.append(Layer(f"block{i}.conv2", "conv3x3", widths[i + 1], widths[i + 1], s * s, s * s, 9))
        if i + 1 == merged:      # the stable image head (fading: the old head, up-sampled afterwards)
            out.append(Layer("ToIm", "conv1x1", widths[i + 1], n_colors, s * s, s * s, 1))
    if merged == 0:
        out.insert(2, Layer("ToIm", "conv1x1", widths[0], n_colors, image_size_init ** 2, image_size_init ** 2, 1))
    if alpha < 1:
        out.append(Layer("ToIm_new", "conv1x1", widths[n_up], n_colors, s * s, s * s, 1))
    return out


def discriminator_layers(widths: List[int], image_size_init: int, res: int, alpha: float = 1.0, n_colors: int = 1) -> List[Layer]:
    """widths are listed from the highest resolution to the lowest, as the reference's N_dis_features (config.py:63)."""
    n_up, merged = _stage(res, image_size_init, alpha)
    nd = len(widths)
    out = []
    s = res
    first = nd - 1 - n_up
    if alpha < 1:
        # fading: new FromImage at full resolution + the new block, old FromImage on the 2x2-averaged image
        out.append(Layer("FromIm_new", "conv1x1", n_colors, widths[first], s * s, s * s, 1))
        out.append(Layer("block_new.conv1", "conv3x3", widths[first], widths[first + 1], s * s // 4, s * s, 9))
        out.append(Layer("block_new.conv2", "conv3x3", widths[first + 1], widths[first + 1], s * s // 4, s * s // 4, 9))
        s //= 2
        first += 1
        out.append(Layer("FromIm", "conv1x1", n_colors, widths[first], s * s, 4 * s * s, 1))
    else:
        out.append(Layer("FromIm", "conv1x1", n_colors, widths[first], s * s, s * s, 1))
    for j in range(first, nd - 1):
        out.append(Layer(f"block{j}.conv1", "conv3x3", widths[j], widths[j + 1], s * s // 4, s * s, 9))
        out.append(Layer(f"block{j}.conv2", "conv3x3", widths[j + 1], widths[j + 1], s * s // 4, s * s // 4, 9))
        s //= 2
    out.append(Layer("conv_last", "conv3x3", widths[-1], widths[-1], s * s, s * s, 9))
    out.append(Layer("score", "valid", widths[-1], 1, 1, s * s, s * s))
    return out


def forward_flops(g_widths: List[int], d_widths: List[int], image_size_init: int, res: int, latent_dim: int,
                  alpha: float = 1.0, n_colors: int = 1) -> Tuple[float, float]:
    """(F_G, F_D): FLOP per image of one forward pass of each net at this stage."""
    fg = sum(l.flops for l in generator_layers(g_widths, image_size_init, res, latent_dim, alpha, n_colors))
    fd = sum(l.flops for l in discriminator_layers(d_widths, image_size_init, res, alpha, n_colors))
    return fg, fd


def iteration_flops(g_widths, d_widths, image_size_init, res, latent_dim, alpha=1.0, n_colors=1) -> float:
    """W_alg = 5 F_G + 14 F_D, FLOP per image per training iteration."""
    fg, fd = forward_flops(g_widths, d_widths, image_size_init, res, latent_dim, alpha, n_colors)
    return 5.0 * fg + 14.0 * fd


def iteration_io_elements(g_widths, d_widths, image_size_init, res, latent_dim, alpha=1.0, n_colors=1, resample_fused=False) -> float:
    """E = 5 E_G + 14 E_D: activation elements moved per image per iteration when LeakyReLU / PixelNorm are fused into the
    contractions (2.50 / 15.48 / 102.50 / 308.61 M elements at 16 / 64 (alpha .5) / 256 / 512)."""
    eg = sum(l.io_elements(resample_fused) for l in generator_layers(g_widths, image_size_init, res, latent_dim, alpha, n_colors))
    ed = sum(l.io_elements(resample_fused) for l in discriminator_layers(d_widths, image_size_init, res, alpha, n_colors))
    return 5.0 * eg + 14.0 * ed
