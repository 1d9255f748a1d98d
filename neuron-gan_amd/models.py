"""Progressively-growing generator / critic with the reference's public surface, executing on gfx950 kernels.

Drop-in for the PG classes of /root/reference/models.py (`Generator_PG`, `Discriminator_PG` and their building
blocks): same class names, constructor signatures, growth methods, attributes and `state_dict` keys, because the
module tree is the same (parameters are created by the same torch constructors in the same order, so a seed
gives the same weights).  What differs is execution: `forward` does not run the children one by one; a small
planner walks the tree, recognises  [resample] -> Conv2d_normalized -> LeakyReLU -> PixelNorm  groups and issues
one fused HIP kernel per group on channels-last tensors (`ops.py`).  There is no PyTorch/CPU fallback: tensors
must live on the GPU and the extension must be built.

Reference behaviours kept on purpose: LeakyReLU comes BEFORE PixelNorm; the equalised-LR
constant multiplies the input, so the bias is unscaled; ToImage has no weight_scale and applies tanh; fade-in
mixes post-tanh images; the critic uses PixelNorm too and has no minibatch-stddev layer.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .configs import config

__all__ = ['Generator_PG', 'Discriminator_PG']

latent_dim_default = config.latent_dim
image_size_default = config.image_size
N_colors_default = config.N_colors
LeakyReLU_neg_slope_default = config.LeakyReLU_leak


def kaiming_init(model: nn.Module, neg_slope=LeakyReLU_neg_slope_default):
    """He-normal weights, zero bias (reference models.py:31-34)."""
    torch.nn.init.kaiming_normal_(model.weight, a=neg_slope, mode='fan_in', nonlinearity='leaky_relu')
    if model.bias is not None:
        model.bias.data.zero_()


# ---- layout helpers: modules speak logical NCHW (like the reference), kernels speak (B, H, W, C) ---------------
def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1)


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)


def _he_gain(act_func):
    if act_func is None:
        return 1.0
    return torch.nn.init.calculate_gain(nonlinearity=act_func[0], param=act_func[1])


class Interpolate(nn.Module):
    """F.interpolate as a module (reference models.py:78-89).  The HIP path implements the two uses the PG nets
    make of it: bilinear x2 (align_corners=None) and bilinear x0.5 (== 2x2 mean)."""

    def __init__(self, size=None, scale_factor=None, mode='nearest', align_corners=None):
        super().__init__()
        self.size = size
        self.scale_factor = scale_factor
        self.mode = mode
        self.align_corners = align_corners

    def resample_code(self):
        if self.size is None and self.mode == 'bilinear' and not self.align_corners:
            if self.scale_factor == 2:
                return ops.RES_UP2
            if self.scale_factor == 0.5:
                return ops.RES_POOL2
        raise NotImplementedError(f'Interpolate({self.extra_repr()}) has no HIP kernel; only bilinear x2 / x0.5 are on the PG path')

    def forward(self, x):
        code = self.resample_code()
        y = ops.Up2.apply(to_nhwc(x)) if code == ops.RES_UP2 else ops.Pool2.apply(to_nhwc(x))
        return to_nchw(y)

    def extra_repr(self):
        out = f'size={self.size}' if self.size is not None else f'scale_factor={self.scale_factor}'
        out += f', mode={self.mode}'
        if self.align_corners is not None:
            out += f', align_corners={self.align_corners}'
        return out


class AvgPool2(nn.AvgPool2d):
    """nn.AvgPool2d(2) whose standalone forward runs the HIP pooling kernel."""

    def forward(self, x):
        if self.kernel_size not in (2, (2, 2)):
            raise NotImplementedError('only 2x2 average pooling is on the PG path')
        return to_nchw(ops.Pool2.apply(to_nhwc(x)))


class PixelNorm(nn.Module):
    """x / sqrt(mean_c(x^2) + eps)  (reference models.py:104-126)."""

    def __init__(self, epsilon=1e-8):
        super().__init__()
        self.epsilon = epsilon

    def forward(self, x):
        if self.epsilon != ops.PIXELNORM_EPS:
            raise NotImplementedError('the HIP kernels are built for the reference epsilon 1e-8')
        y, _ = ops.LReLUPN.apply(to_nhwc(x), None, 1.0)  # slope 1 == no activation
        return to_nchw(y)

    def extra_repr(self):
        return 'epsilon={}'.format(self.epsilon)


class ToImage(nn.Module):
    """1x1 conv to colour space (no bias, no weight_scale) + tanh  (reference models.py:133-149)."""

    def __init__(self, in_channels, N_colors):
        super().__init__()
        self.in_channels = in_channels
        self.N_colors = N_colors
        self.layers = nn.Sequential()
        conv = nn.Conv2d(in_channels, N_colors, kernel_size=1, stride=1, padding=0, bias=False)
        kaiming_init(conv)
        self.layers.append(conv)
        self.layers.append(nn.Tanh())

    def nhwc(self, x):
        return ops.ToImage.apply(x, self.layers[0].weight)

    def forward(self, x):
        return to_nchw(self.nhwc(to_nhwc(x)))

    def extra_repr(self):
        return 'in_channels={}, N_colors={}'.format(self.in_channels, self.N_colors)


class FromImage(nn.Module):
    """1x1 conv from colour space with bias, no activation  (reference models.py:156-165)."""

    def __init__(self, N_colors, out_channels):
        super().__init__()
        self.out_channels = out_channels
        self.N_colors = N_colors
        self.conv = nn.Conv2d(N_colors, out_channels, kernel_size=1, stride=1, padding=0)
        kaiming_init(self.conv)

    def nhwc(self, x, pool=False):
        return ops.FromImage.apply(x, self.conv.weight, self.conv.bias, pool)

    def forward(self, x):
        return to_nchw(self.nhwc(to_nhwc(x)))

    def extra_repr(self):
        return 'N_colors={}, out_channels={}'.format(self.N_colors, self.out_channels)


class Conv2d_normalized(nn.Conv2d):
    """Conv2d whose INPUT is multiplied by gain/sqrt(fan) (reference models.py:172-204)."""

    def __init__(self, *args, scale_mode='fan_in', act_func=('leaky_relu', LeakyReLU_neg_slope_default), **kwargs):
        super().__init__(*args, **kwargs)
        kaiming_init(self)
        self.weight_scale_mode = scale_mode
        if scale_mode == 'fan_in':
            n_connections = self.weight.shape[1] * np.prod(self.kernel_size)
        elif scale_mode == 'fan_out':
            n_connections = self.weight.shape[0] * np.prod(self.kernel_size)
        else:
            raise ValueError('{} is not a supported mode', scale_mode)
        self.scale_value = float(_he_gain(act_func) / np.sqrt(n_connections))
        self.register_buffer('weight_scale', torch.tensor(self.scale_value), persistent=False)

    def is_3x3(self):
        return (self.kernel_size == (3, 3) and self.padding == (1, 1) and self.stride == (1, 1) and self.groups == 1
                and self.dilation == (1, 1) and self.padding_mode == 'zeros')

    def nhwc(self, x, resample=ops.RES_NONE):
        """pre-activation output on a channels-last tensor"""
        if self.is_3x3():
            return ops.Conv.apply(x, self.weight, self.bias, resample, self.scale_value)
        if self.padding == (0, 0) and self.out_channels == 1 and tuple(x.shape[1:3]) == self.kernel_size:
            return ops.FinalDot.apply(x, self.weight, self.bias, self.scale_value).reshape(-1, 1, 1, 1)
        raise NotImplementedError(f'no HIP kernel for {self}')

    def forward(self, x):
        return to_nchw(self.nhwc(to_nhwc(x)))


class Linear_normalized(nn.Linear):
    """Linear whose INPUT is multiplied by gain/sqrt(fan) (reference models.py:208-241).  On the PG path it is
    always followed by Unflatten -> LeakyReLU -> PixelNorm and runs fused with them (see `run_layers`)."""

    def __init__(self, *args, scale_mode='fan_in', act_func=('leaky_relu', LeakyReLU_neg_slope_default), **kwargs):
        super().__init__(*args, **kwargs)
        kaiming_init(self)
        self.weight_scale_mode = scale_mode
        if scale_mode == 'fan_in':
            n_connections = self.weight.shape[1]
        elif scale_mode == 'fan_out':
            n_connections = self.weight.shape[0]
        else:
            raise ValueError('{} is not a supported mode', scale_mode)
        self.scale_value = float(_he_gain(act_func) / math.sqrt(n_connections))
        self.register_buffer('weight_scale', torch.tensor(self.scale_value), persistent=False)

    def forward(self, x):
        raise NotImplementedError('Linear_normalized runs fused with Unflatten/LeakyReLU/PixelNorm on the HIP path; '
                                  'call the enclosing network')


class Conv2d_scale_block(nn.Sequential):
    """resample -> conv -> LReLU -> PixelNorm -> conv -> LReLU -> PixelNorm  (reference models.py:245-268).
    Runs as two fused kernels: the resampling is applied while the first conv stages its input tile."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=1, scale_factor=None,
                 LeakyReLU_neg_slope=LeakyReLU_neg_slope_default):
        super().__init__()
        if scale_factor < 1:
            self.append(AvgPool2(kernel_size=int(1 / scale_factor)))
        else:
            self.append(Interpolate(scale_factor=scale_factor, mode='bilinear'))
        act = ('leaky_relu', LeakyReLU_neg_slope)
        self.append(Conv2d_normalized(in_channels, out_channels, kernel_size, stride=1, padding=padding,
                                      padding_mode='zeros', bias=False, act_func=act))
        self.append(nn.LeakyReLU(negative_slope=LeakyReLU_neg_slope))
        self.append(PixelNorm())
        self.append(Conv2d_normalized(out_channels, out_channels, kernel_size, stride=1, padding=padding,
                                      padding_mode='zeros', bias=False, act_func=act))
        self.append(nn.LeakyReLU(negative_slope=LeakyReLU_neg_slope))
        self.append(PixelNorm())

    def nhwc(self, x):
        return run_layers(self, x)

    def forward(self, x):
        return to_nchw(self.nhwc(to_nhwc(x)))


# ---- the fusion planner ---------------------------------------------------------------------------------------
def _resample_of(m):
    if isinstance(m, Interpolate):
        return m.resample_code()
    if isinstance(m, nn.AvgPool2d):
        if m.kernel_size not in (2, (2, 2)):
            raise NotImplementedError('only 2x2 average pooling is on the PG path')
        return ops.RES_POOL2
    return None


def _plan(mods, steps=None, pending=None):
    """Flatten an nn.Sequential of PG building blocks (nested Conv2d_scale_blocks included) into execution steps, merging
    [resample] conv LeakyReLU PixelNorm  groups:  ('stem', linear, size, slope) | ('conv_lrelu_pn', conv, resample, slope) |
    ('conv', conv, resample) | ('lrelu_pn', slope) | ('resample', code) | ('flatten',)"""
    top = steps is None
    if top:
        steps, pending = [], [ops.RES_NONE]
    mods = list(mods)
    i, n = 0, len(mods)

    def flush():
        if pending[0] != ops.RES_NONE:
            steps.append(('resample', pending[0]))
            pending[0] = ops.RES_NONE

    while i < n:
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < n else None
        nxt2 = mods[i + 2] if i + 2 < n else None
        nxt3 = mods[i + 3] if i + 3 < n else None
        res = _resample_of(m)
        if isinstance(m, Conv2d_scale_block):
            flush()
            _plan(m, steps, pending)
            flush()
            i += 1
        elif res is not None:
            flush()
            pending[0] = res
            i += 1
        elif (isinstance(m, Linear_normalized) and isinstance(nxt, nn.Unflatten) and isinstance(nxt2, nn.LeakyReLU)
              and isinstance(nxt3, PixelNorm)):
            c, s, s2 = nxt.unflattened_size
            if s != s2 or m.bias is not None:
                raise NotImplementedError('generator stem must be square and bias-free')
            steps.append(('stem', m, s, nxt2.negative_slope))
            i += 4
        elif isinstance(m, Conv2d_normalized) and m.is_3x3() and isinstance(nxt, nn.LeakyReLU) and isinstance(nxt2, PixelNorm):
            steps.append(('conv_lrelu_pn', m, pending[0], nxt.negative_slope))
            pending[0] = ops.RES_NONE
            i += 3
        elif isinstance(m, Conv2d_normalized):
            if m.is_3x3():
                steps.append(('conv', m, pending[0]))
                pending[0] = ops.RES_NONE
            else:
                flush()
                steps.append(('conv', m, ops.RES_NONE))
            i += 1
        elif isinstance(m, nn.LeakyReLU) and isinstance(nxt, PixelNorm):
            flush()
            steps.append(('lrelu_pn', m.negative_slope))
            i += 2
        elif isinstance(m, PixelNorm):
            flush()
            steps.append(('lrelu_pn', 1.0))
            i += 1
        elif isinstance(m, nn.Flatten):
            flush()
            steps.append(('flatten',))
            i += 1
        else:
            raise NotImplementedError(f'no HIP kernel for layer {type(m).__name__} at position {i}')
    if top:
        flush()
    return steps


def _odd_width(m):
    return m.weight.shape[0] % 16 != 0 or m.weight.shape[1] % 16 != 0


def _conv_any_width(x, m, res, slope):
    """3x3 conv (+ LeakyReLU -> PixelNorm when `slope` is given) for channel counts that are not multiples of 16 -- the reference's
    constructors take any widths and its presets 0004-0006 end in 8-channel blocks (configs/config.py:86-92).  The contraction
    kernels work on multiples of 16, so the weight, bias and input are zero-padded up to the next multiple, the conv runs on the
    padded shapes, the padding channels are cut off again and LeakyReLU -> PixelNorm runs on the REAL channel count (its mean is
    over the layer's own channels).  Padding and slicing are torch's differentiable ops, every kernel involved is closed under
    double-backward, so all gradient orders are exact; it is an unfused compatibility path (no PixelNorm hand-off, no ToImage /
    first-block fusion), not a fast one."""
    co, ci = m.weight.shape[0], m.weight.shape[1]
    if co % 4:
        raise NotImplementedError(f'the channels-last kernels take channel counts that are multiples of 4, got {co}')
    cip, cop = -(-ci // 16) * 16, -(-co // 16) * 16
    w = torch.nn.functional.pad(m.weight, (0, 0, 0, 0, 0, cip - ci, 0, cop - co))
    b = torch.nn.functional.pad(m.bias, (0, cop - co)) if m.bias is not None else None
    if x.shape[-1] != cip:
        x = torch.nn.functional.pad(x, (0, cip - x.shape[-1]))
    c = ops.Conv.apply(x.contiguous(), w, b, res, m.scale_value)
    if cop != co:
        c = c[..., :co].contiguous()
    if slope is None:
        return c
    y, _ = ops.LReLUPN.apply(c, None, slope)
    return y


def _exec(steps, x, link=None, to_image=None, then=None):
    """Run planned steps on a channels-last tensor (or on latents for the stem).  `link`: PNLink of the LeakyReLU->PixelNorm that
    produced x, if x has no other consumer.  Returns (output, link of the output).  Inside `ops.first_order_only()` consecutive
    LeakyReLU->PixelNorm producers and conv consumers are linked (ops.PNLink).  `to_image`: a ToImage module applied to the result;
    where the last step is a fused conv and the shape allows it, conv + ToImage run as one kernel (ops.ConvLReLUPNToImage).
    `then`: the planned step the caller will run on the result (the critic runs its first block and the rest as two calls)."""
    linking = ops.first_order_enabled() and torch.is_grad_enabled()
    last = len(steps) - 1
    for idx, st in enumerate(steps):
        kind = st[0]
        if kind == 'conv_lrelu_pn' and _odd_width(st[1]):
            x, link = _conv_any_width(x, st[1], st[2], st[3]), None
        elif kind == 'conv' and st[1].is_3x3() and _odd_width(st[1]):
            x, link = _conv_any_width(x, st[1], st[2], None), None
        elif kind == 'conv_lrelu_pn':
            _, m, res, slope = st
            if (idx == last and to_image is not None and ops.first_order_enabled()
                    and ops.to_image_fusable(x, m.weight, to_image.layers[0].weight, res)):
                t = ops.ConvLReLUPNToImage.apply(x, m.weight, m.bias, to_image.layers[0].weight, res, m.scale_value, slope, link,
                                                 torch.is_grad_enabled())
                return t, None
            out_link = ops.PNLink() if linking else None
            # the next step is an avg-pooled conv: ask this conv's kernel for the pooled copy of its output (ops._run_conv)
            nxt = steps[idx + 1] if idx < last else then
            pool_out = (nxt is not None and nxt[0] in ('conv_lrelu_pn', 'conv') and len(nxt) > 2 and nxt[2] == ops.RES_POOL2
                        and not _odd_width(nxt[1]))
            if pool_out:
                x, _ = ops.ConvLReLUPN.apply(x, m.weight, m.bias, res, m.scale_value, slope, link, out_link, True)
            elif link is not None or out_link is not None:
                x, _ = ops.ConvLReLUPN.apply(x, m.weight, m.bias, res, m.scale_value, slope, link, out_link)
            else:
                x, _ = ops.ConvLReLUPN.apply(x, m.weight, m.bias, res, m.scale_value, slope)
            link = out_link
        elif kind == 'stem':
            _, m, size, slope = st
            out_link = ops.PNLink() if linking else None
            if out_link is not None:
                x, _ = ops.LinearLReLUPN.apply(x, m.weight, size, m.scale_value, slope, out_link)
            else:
                x, _ = ops.LinearLReLUPN.apply(x, m.weight, size, m.scale_value, slope)
            link = out_link
        else:
            link = None           # every other operator consumes the gradient w.r.t. its input as it is
            if kind == 'conv':
                x = st[1].nhwc(x, st[2])
            elif kind == 'lrelu_pn':
                x, _ = ops.LReLUPN.apply(x, None, st[1])
            elif kind == 'resample':
                x = _apply_resample(x, st[1])
            elif kind == 'flatten':
                x = x.reshape(x.shape[0], -1)
            else:
                raise AssertionError(kind)
    if to_image is not None:
        return to_image.nhwc(x), None
    return x, link


def run_layers(seq, x):
    """Evaluate an nn.Sequential of PG building blocks on a channels-last tensor (or on latents for the stem),
    fusing  [resample] conv LeakyReLU PixelNorm  groups into single kernels."""
    return _exec(_plan(seq), x)[0]


def _apply_resample(x, code):
    if code == ops.RES_UP2:
        return ops.Up2.apply(x)
    if code == ops.RES_POOL2:
        return ops.Pool2.apply(x)
    return x


class _ProgressiveNet(nn.Module):
    """Growth state machine shared by both nets (reference models.py:355-392, 526-564)."""
    alpha: torch.Tensor

    def _init_alpha(self, persistent):
        self.register_buffer('alpha', torch.tensor(1.0), persistent=persistent)
        self._alpha_host = np.float32(1.0)

    def _set_alpha(self, value):
        # same fp32 arithmetic as the reference's 0-dim tensor, kept on the host so `forward` never syncs
        self._alpha_host = np.float32(value)
        self.alpha.fill_(float(self._alpha_host))

    def alpha_value(self) -> float:
        return float(self._alpha_host)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if prefix + 'alpha' in state_dict:
            self._alpha_host = np.float32(float(state_dict[prefix + 'alpha']))

    def increase_resolution(self):
        assert self.alpha_value() >= 1, 'The previous transition has not ended.'
        self._set_alpha(np.float32(0) * self._alpha_host)
        self.N_layers += 1
        self.image_size *= 2
        assert self.image_size <= self.image_size_max, (
            f'The image size ({self.image_size}) is greater than the maximum ({self.image_size_max})')

    def advance_transition(self, alpha_step=0.1):
        self._set_alpha(self._alpha_host + np.float32(alpha_step))
        if self.alpha_value() >= 1.0:
            self._merge_pending_block()

    def set_resolution(self, res: int, alpha=1.0):
        assert res % self.image_size == 0, 'The resolution must be divisible by {}'.format(self.image_size)
        assert math.log2(res / self.image_size).is_integer(), (
            f'{res} cannot be attained by multiplying the initial resolution ({self.image_size}) by a power of 2.')
        assert res <= self.image_size_max, 'The resolution must be smaller than {}'.format(self.image_size_max)
        while self.image_size != res:
            self.increase_resolution()
            self.advance_transition(alpha if self.image_size == res else 1.0)

    def _tag_parameters(self):
        """state_dict names change when a block moves from conv_block_list into layers; the construction-time name is a
        stable identity for per-parameter optimiser state."""
        for name, p in self.named_parameters():
            p._ngan_name = name

    def _saved_attr_names(self, extra):
        names = ['LeakyReLU_neg_slope', 'N_colors', 'N_features_per_layer', 'N_layers', 'N_layers_max', 'image_size',
                 'image_size_init', 'image_size_max', 'training'] + extra
        return sorted(names) + ['alpha']


def _drop_legacy_entries(state_dict, list_prefix, n_drop, from_start):
    """Old-format checkpoints keep already-merged entries of a ModuleList (reference models.py:38-63): drop `n_drop` of them
    from one end and renumber the rest when they are dropped from the start."""
    idx = sorted({int(k.split('.')[1]) for k in state_dict if k.startswith(list_prefix + '.')})
    if not idx or n_drop <= 0:
        return state_dict
    assert n_drop <= len(idx), 'Cannot remove more than {} layers'.format(len(idx))
    dropped = set(idx[:n_drop] if from_start else idx[len(idx) - n_drop:])
    out = type(state_dict)()
    for k, v in state_dict.items():
        if k.startswith(list_prefix + '.'):
            parts = k.split('.')
            i = int(parts[1])
            if i in dropped:
                continue
            if from_start:
                parts[1] = str(i - n_drop)
            k = '.'.join(parts)
        out[k] = v
    return out


def _drop_prefix(state_dict, prefix):
    return type(state_dict)((k, v) for k, v in state_dict.items() if not k.startswith(prefix + '.'))


def _count_list_entries(state_dict, list_prefix):
    idx = {int(k.split('.')[1]) for k in state_dict if k.startswith(list_prefix + '.')}
    return max(idx) + 1 if idx else 0


class Generator_PG(_ProgressiveNet):
    def __init__(self, N_features_per_layer: list, image_size_init=4, latent_dim=latent_dim_default,
                 LeakyReLU_neg_slope=LeakyReLU_neg_slope_default, N_colors=N_colors_default):
        N_scaling = len(N_features_per_layer) - 1
        super().__init__()
        self.latent_dim = latent_dim
        self.N_features_per_layer = N_features_per_layer
        self.N_layers = 1
        self.N_layers_max = len(N_features_per_layer)
        self.N_colors = N_colors
        self.image_size_init = image_size_init
        self.image_size = image_size_init
        self.image_size_max = 2 ** N_scaling * image_size_init
        self.LeakyReLU_neg_slope = LeakyReLU_neg_slope
        self._init_alpha(persistent=False)

        act = ('leaky_relu', LeakyReLU_neg_slope)
        f0 = N_features_per_layer[0]
        self.layers = nn.Sequential()
        self.layers.append(Linear_normalized(latent_dim, f0 * image_size_init ** 2, bias=False, act_func=act))
        self.layers.append(nn.Unflatten(dim=1, unflattened_size=(f0, image_size_init, image_size_init)))
        self.layers.append(nn.LeakyReLU(negative_slope=LeakyReLU_neg_slope))
        self.layers.append(PixelNorm())
        self.layers.append(Conv2d_normalized(f0, f0, kernel_size=3, stride=1, padding=1, padding_mode='zeros',
                                             bias=False, act_func=act))
        self.layers.append(nn.LeakyReLU(negative_slope=LeakyReLU_neg_slope))
        self.layers.append(PixelNorm())

        self.conv_block_list = nn.ModuleList()
        for i in range(len(N_features_per_layer) - 1):
            self.conv_block_list.append(Conv2d_scale_block(in_channels=N_features_per_layer[i],
                                                           out_channels=N_features_per_layer[i + 1],
                                                           scale_factor=2, kernel_size=3))
        self.ToIm_list = nn.ModuleList()
        for i in range(len(N_features_per_layer)):
            self.ToIm_list.append(ToImage(N_features_per_layer[i], N_colors))
        self.ToIm = self.ToIm_list.pop(0)
        self.upsample = Interpolate(scale_factor=2, mode='bilinear')
        self.saved_attrs = self._saved_attr_names(['latent_dim'])
        self._tag_parameters()

    def forward(self, x):
        # the generator is only ever differentiated once (train.py:384): its conv chain always runs in first-order mode
        with ops.first_order_only():
            if self.alpha_value() < 1:
                h, _ = _exec(_plan(self.layers), x)          # h has two consumers: no PixelNorm hand-off across it
                im_start = ops.Up2.apply(self.ToIm.nhwc(h))
                im_end, _ = _exec(_plan(self.conv_block_list[0]), h, None, to_image=self.ToIm_list[0])
                out = ops.Lerp.apply(im_start, im_end, self.alpha.reshape(1))
            else:
                out, _ = _exec(_plan(self.layers), x, None, to_image=self.ToIm)
        return to_nchw(out)

    def _merge_pending_block(self):
        self.layers.append(self.conv_block_list.pop(0))
        self.ToIm = self.ToIm_list.pop(0)

    @classmethod
    def from_state_dict(cls, filename, device=torch.device('cpu'), verbose=True):
        """Rebuild a generator from a checkpoint written by `Checkpointer` (reference models.py:394-444), including
        checkpoints in the older layout that still carry merged ToIm_list / conv_block_list entries."""
        from .utils import load_checkpoint_dict
        saved = load_checkpoint_dict(filename, device)
        attrs = saved['Generator_attrs']
        ctor = {k: attrs[k] for k in ('N_features_per_layer', 'image_size_init', 'LeakyReLU_neg_slope', 'N_colors') if k in attrs}
        if 'latent_dim' in attrs:
            ctor['latent_dim'] = attrs['latent_dim']
        obj = cls(**ctor)
        obj.set_resolution(attrs['image_size'], float(attrs['alpha']))
        state = saved['Generator_state']
        n_toim = _count_list_entries(state, 'ToIm_list')
        if n_toim > len(obj.ToIm_list):
            if verbose:
                print('Warning! Loaded state dict in old format. Keys will be removed to match the new format.')
            state = _drop_legacy_entries(state, 'ToIm_list', n_toim - len(obj.ToIm_list), from_start=True)
            state = _drop_legacy_entries(state, 'conv_block_list',
                                         _count_list_entries(state, 'conv_block_list') - len(obj.conv_block_list), from_start=True)
            state = _drop_prefix(_drop_prefix(state, 'ToIm_prev'), 'last_conv_block')
        obj.load_state_dict(state)
        if verbose:
            print('Loaded training state from {}'.format(filename))
        return obj


class Discriminator_PG(_ProgressiveNet):
    def __init__(self, N_features_per_layer: list, image_size_init=4, LeakyReLU_neg_slope=LeakyReLU_neg_slope_default,
                 N_colors=N_colors_default):
        N_scaling = len(N_features_per_layer) - 1
        super().__init__()
        self.N_features_per_layer = N_features_per_layer
        self.N_layers = 1
        self.N_layers_max = len(N_features_per_layer)
        self.N_colors = N_colors
        self.image_size_init = image_size_init
        self.image_size = image_size_init
        self.image_size_max = 2 ** N_scaling * image_size_init
        self.LeakyReLU_neg_slope = LeakyReLU_neg_slope
        self._init_alpha(persistent=True)

        act = ('leaky_relu', LeakyReLU_neg_slope)
        fl = N_features_per_layer[-1]
        self.layers = nn.Sequential()
        self.layers.append(Conv2d_normalized(fl, fl, kernel_size=3, stride=1, padding=1, padding_mode='zeros', act_func=act))
        self.layers.append(nn.LeakyReLU(negative_slope=LeakyReLU_neg_slope))
        self.layers.append(PixelNorm())
        self.layers.append(Conv2d_normalized(fl, 1, (image_size_init, image_size_init), stride=1, padding=0, act_func=act))
        self.layers.append(nn.Flatten())

        self.conv_block_list = nn.ModuleList()
        for i in range(len(N_features_per_layer) - 1):
            self.conv_block_list.append(Conv2d_scale_block(in_channels=N_features_per_layer[i],
                                                           out_channels=N_features_per_layer[i + 1],
                                                           scale_factor=0.5, kernel_size=3))
        self.FromIm_list = nn.ModuleList()
        for i in range(len(N_features_per_layer)):
            self.FromIm_list.append(FromImage(N_colors, N_features_per_layer[i]))
        self.FromIm = self.FromIm_list.pop(-1)
        self.downsample = Interpolate(scale_factor=0.5, mode='bilinear')
        self.saved_attrs = self._saved_attr_names([])
        self._tag_parameters()

    def forward(self, x):
        x = to_nhwc(x)
        if self.alpha_value() < 1:
            y_start = self.FromIm.nhwc(x, pool=True)           # FromIm(downsample(x)), pooled on load
            y_end, _ = self._from_image_then_block(self.FromIm_list[-1], self.conv_block_list[-1], x)
            y = ops.Lerp.apply(y_start, y_end, self.alpha.reshape(1))
            return _exec(_plan(self.layers), y)[0]
        first = self.layers[0]
        if isinstance(first, Conv2d_scale_block):
            rest = _plan(list(self.layers)[1:])
            y, link = self._from_image_then_block(self.FromIm, first, x, then=rest[0] if rest else None)
            return _exec(rest, y, link)[0]
        return _exec(_plan(self.layers), self.FromIm.nhwc(x))[0]

    @staticmethod
    def _from_image_then_block(from_im, block, x, then=None):
        """block(FromImage(x)) for a down-sampling block -> (output, its PNLink).  FromImage is affine per pixel and AvgPool2d is
        linear, so pool(FromImage(x)) == FromImage(pool(x)): the image is pooled while FromImage loads it and the block's first
        conv runs without resampling -- the C-channel tensor at the image's full resolution is never written."""
        pool = _resample_of(block[0]) == ops.RES_POOL2
        steps = _plan(list(block)[1:] if pool else block)
        first = steps[0]
        if (ops.first_order_enabled() and first[0] == 'conv_lrelu_pn' and first[2] == ops.RES_NONE
                and ops.first_block_fusable(x, from_im.conv.weight, first[1].weight)):
            # differentiated once, one colour: FromImage folds into the conv (ops.FirstBlock), its output is never materialised
            link = ops.PNLink() if torch.is_grad_enabled() else None
            y, _ = ops.FirstBlock.apply(x, from_im.conv.weight, from_im.conv.bias, first[1].weight, first[1].bias, pool,
                                        first[1].scale_value, first[3], link)
            return _exec(steps[1:], y, link, then=then)
        return _exec(steps, from_im.nhwc(x, pool=pool), then=then)

    def _merge_pending_block(self):
        self.layers.insert(0, self.conv_block_list.pop(-1))
        self.FromIm = self.FromIm_list.pop(-1)

    @classmethod
    def from_state_dict(cls, filename, device=torch.device('cpu'), verbose=True):
        """Rebuild a critic from a checkpoint (reference models.py:566-616), old layout included."""
        from .utils import load_checkpoint_dict
        saved = load_checkpoint_dict(filename, device)
        attrs = saved['Discriminator_attrs']
        ctor = {k: attrs[k] for k in ('N_features_per_layer', 'image_size_init', 'LeakyReLU_neg_slope', 'N_colors') if k in attrs}
        obj = cls(**ctor)
        obj.set_resolution(attrs['image_size'], float(attrs['alpha']))
        state = saved['Discriminator_state']
        n_from = _count_list_entries(state, 'FromIm_list')
        if n_from > len(obj.FromIm_list):
            if verbose:
                print('Warning! Loaded state dict in old format. Keys will be removed to match the new format.')
            state = _drop_legacy_entries(state, 'FromIm_list', n_from - len(obj.FromIm_list), from_start=False)
            state = _drop_legacy_entries(state, 'conv_block_list',
                                         _count_list_entries(state, 'conv_block_list') - len(obj.conv_block_list), from_start=False)
            state = _drop_prefix(_drop_prefix(state, 'FromIm_prev'), 'first_conv_block')
        obj.load_state_dict(state)
        if verbose:
            print('Loaded training state from {}'.format(filename))
        return obj
