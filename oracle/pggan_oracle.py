"""CPU oracle for the PGGAN / WGAN-GP training step.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) *restatement* of the algorithm the
reference implements on its hot path.  It is the checker the HIP path is compared
against; it is never imported by the product package (`neuron-gan_amd/`).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may use it.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here
against golden vectors produced by `oracle/make_golden.py`, which imports the
reference's own `models.py` (Generator_PG / Discriminator_PG) in the build
container and records inputs, weights and outputs under `tests/golden/`.

The oracle is functional: a network is a flat dict of tensors keyed by the
reference's `state_dict` names (so a reference checkpoint or a golden fixture can be
fed in directly) plus a small `NetSpec`.  The arithmetic itself lives in torch's
ATen CPU kernels, exactly as it does for the reference (which pins torch==1.13.1,
/root/reference/requirements.txt:4).

Reference citations are `file:line` into /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

PIXELNORM_EPS = 1e-8  # models.py:105


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def he_gain(slope: float) -> float:
    """torch.nn.init.calculate_gain('leaky_relu', slope)  (models.py:198, 235)."""
    return math.sqrt(2.0 / (1.0 + slope * slope))


def weight_scale(fan_in: int, slope: float) -> float:
    """Equalised-LR constant gain/sqrt(fan_in); applied to the INPUT (models.py:201-204, 238-241)."""
    return he_gain(slope) / math.sqrt(fan_in)


def pixel_norm(x: torch.Tensor, eps: float = PIXELNORM_EPS) -> torch.Tensor:
    """x / sqrt(mean_c(x^2) + eps)  (models.py:118, 126)."""
    return x / torch.sqrt(torch.mean(x * x, dim=1, keepdim=True) + eps)


def lrelu(x: torch.Tensor, slope: float) -> torch.Tensor:
    """nn.LeakyReLU(negative_slope)  (models.py:263)."""
    return F.leaky_relu(x, slope)


def up2(x: torch.Tensor) -> torch.Tensor:
    """Interpolate(scale_factor=2, mode='bilinear'), align_corners=None  (models.py:87-89, 257, 335)."""
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=None)


def pool2(x: torch.Tensor) -> torch.Tensor:
    """nn.AvgPool2d(2)  (models.py:254); also equals Interpolate(0.5,'bilinear')  (models.py:507)."""
    return F.avg_pool2d(x, 2)


def scaled_conv(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], slope: float, padding: int) -> torch.Tensor:
    """Conv2d_normalized.forward: conv2d(weight_scale * x, W, b)  (models.py:203-204)."""
    fan_in = w.shape[1] * w.shape[2] * w.shape[3]  # models.py:190
    s = torch.tensor(weight_scale(fan_in, slope), dtype=x.dtype)
    return F.conv2d(s * x, w, b, stride=1, padding=padding)


def scale_block(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, direction: str, slope: float) -> torch.Tensor:
    """Conv2d_scale_block: resample -> conv -> LReLU -> PixelNorm -> conv -> LReLU -> PixelNorm  (models.py:252-268)."""
    x = up2(x) if direction == "up" else pool2(x)
    x = pixel_norm(lrelu(scaled_conv(x, w1, None, slope, 1), slope))
    x = pixel_norm(lrelu(scaled_conv(x, w2, None, slope, 1), slope))
    return x


# --------------------------------------------------------------------------------------
# network description
# --------------------------------------------------------------------------------------
@dataclass
class NetSpec:
    """What is needed besides the weights to evaluate a net at its current stage."""
    image_size_init: int
    slope: float = 0.2
    alpha: float = 1.0  # fade-in coefficient; < 1 means a transition is in progress


def _block_indices(params: Params, prefix: str) -> List[int]:
    """Sorted indices i for which '<prefix>.<i>.1.weight' exists (a Conv2d_scale_block)."""
    out = set()
    for k in params:
        parts = k.split(".")
        if len(parts) == 4 and parts[0] == prefix and parts[2] == "1" and parts[3] == "weight":
            out.add(int(parts[1]))
    return sorted(out)


def generator_features(params: Params, z: torch.Tensor, spec: NetSpec) -> torch.Tensor:
    """Generator_PG.layers(z): Linear -> Unflatten -> LReLU -> PN -> conv -> LReLU -> PN -> merged blocks
    (models.py:295-316, 374)."""
    w0 = params["layers.0.weight"]  # (C0*S*S, latent)
    s0 = torch.tensor(weight_scale(w0.shape[1], spec.slope), dtype=z.dtype)  # models.py:227, 238
    h = F.linear(s0 * z, w0)  # models.py:241
    c0 = w0.shape[0] // (spec.image_size_init ** 2)
    h = h.view(z.shape[0], c0, spec.image_size_init, spec.image_size_init)  # models.py:301-302 (NCHW unflatten)
    h = pixel_norm(lrelu(h, spec.slope))  # models.py:310-311
    h = pixel_norm(lrelu(scaled_conv(h, params["layers.4.weight"], None, spec.slope, 1), spec.slope))  # 312-316
    for i in _block_indices(params, "layers"):  # merged blocks live at layers.7, layers.8, ...
        h = scale_block(h, params[f"layers.{i}.1.weight"], params[f"layers.{i}.4.weight"], "up", spec.slope)
    return h


def to_image(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """ToImage: tanh(conv1x1(x)), no bias, no weight_scale  (models.py:141-149)."""
    return torch.tanh(F.conv2d(x, w))


def generator_forward(params: Params, z: torch.Tensor, spec: NetSpec) -> torch.Tensor:
    """Generator_PG.forward  (models.py:344-353)."""
    h = generator_features(params, z, spec)
    if spec.alpha < 1:
        im_start = up2(to_image(h, params["ToIm.layers.0.weight"]))  # models.py:348
        hb = scale_block(h, params["conv_block_list.0.1.weight"], params["conv_block_list.0.4.weight"], "up", spec.slope)
        im_end = to_image(hb, params["ToIm_list.0.layers.0.weight"])  # models.py:349
        return im_start + spec.alpha * (im_end - im_start)  # models.py:350
    return to_image(h, params["ToIm.layers.0.weight"])  # models.py:353


def from_image(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """FromImage: conv1x1(x) + bias, no activation  (models.py:161-165)."""
    return F.conv2d(x, w, b)


def discriminator_trunk(params: Params, y: torch.Tensor, spec: NetSpec) -> torch.Tensor:
    """Discriminator_PG.layers(y): merged blocks (highest resolution first, models.py:546) ->
    conv3x3+bias -> LReLU -> PN -> conv SxS valid + bias -> Flatten  (models.py:468-490)."""
    merged = _block_indices(params, "layers")
    for i in merged:
        y = scale_block(y, params[f"layers.{i}.1.weight"], params[f"layers.{i}.4.weight"], "down", spec.slope)
    k = len(merged)
    y = pixel_norm(lrelu(scaled_conv(y, params[f"layers.{k}.weight"], params[f"layers.{k}.bias"], spec.slope, 1), spec.slope))
    y = scaled_conv(y, params[f"layers.{k + 3}.weight"], params[f"layers.{k + 3}.bias"], spec.slope, 0)
    return y.flatten(1)


def discriminator_forward(params: Params, x: torch.Tensor, spec: NetSpec) -> torch.Tensor:
    """Discriminator_PG.forward  (models.py:516-524)."""
    if spec.alpha < 1:
        y_start = from_image(pool2(x), params["FromIm.conv.weight"], params["FromIm.conv.bias"])  # models.py:519
        last = max(_block_indices(params, "conv_block_list"))
        last_from = max(int(k.split(".")[1]) for k in params if k.startswith("FromIm_list."))
        y_end = from_image(x, params[f"FromIm_list.{last_from}.conv.weight"], params[f"FromIm_list.{last_from}.conv.bias"])
        y_end = scale_block(y_end, params[f"conv_block_list.{last}.1.weight"], params[f"conv_block_list.{last}.4.weight"],
                            "down", spec.slope)  # models.py:520
        y = y_start + spec.alpha * (y_end - y_start)  # models.py:521
    else:
        y = from_image(x, params["FromIm.conv.weight"], params["FromIm.conv.bias"])  # models.py:524
    return discriminator_trunk(params, y, spec)


# --------------------------------------------------------------------------------------
# latent sampler and losses
# --------------------------------------------------------------------------------------
def sample_latent_vec(size: Tuple[int, int], generator: Optional[torch.Generator] = None,
                      dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """randn -> clamp(-5, 5) -> L2-normalise rows  (utils.py:77-78).  Global RNG stream unless `generator`."""
    z = torch.randn(*size, generator=generator).clamp(-5, 5)
    z = z / z.norm(p=2, dim=1, keepdim=True)
    return z.to(dtype)


def d_w_loss(pg: Params, sg: NetSpec, pd: Params, sd: NetSpec, real: torch.Tensor, z: torch.Tensor,
             drift_epsilon: float):
    """D_W_loss.forward  (loss_functions.py:14-47) with the latent injected."""
    real_score = discriminator_forward(pd, real, sd)
    score_real = real_score.mean()  # loss_functions.py:22
    with torch.no_grad():
        fake = generator_forward(pg, z, sg)  # .detach()  loss_functions.py:26
    score_fake = discriminator_forward(pd, fake, sd).mean()  # loss_functions.py:29
    loss = -score_real + score_fake  # loss_functions.py:32
    if torch.isnan(score_real):
        raise ValueError("Real loss is nan.")  # loss_functions.py:35-37
    if torch.isnan(score_fake):
        raise ValueError("Fake loss is nan.")  # loss_functions.py:38-41
    if drift_epsilon > 0:
        loss = loss + drift_epsilon * torch.square(real_score).mean()  # loss_functions.py:44-45
    return loss, score_real, score_fake


def grad_penalty(pg: Params, sg: NetSpec, pd: Params, sd: NetSpec, real: torch.Tensor, z: torch.Tensor,
                 eps: torch.Tensor, lam: float, return_norms: bool = False):
    """D_grad_pen_loss.forward  (loss_functions.py:157-180) with latent and epsilon injected."""
    if not lam > 0:
        return torch.tensor(0)  # loss_functions.py:179
    with torch.no_grad():
        x_tilde = generator_forward(pg, z, sg)  # loss_functions.py:167
    x_hat = eps * real + (1 - eps) * x_tilde  # loss_functions.py:171
    x_hat.requires_grad_()
    out = discriminator_forward(pd, x_hat, sd)
    (g,) = torch.autograd.grad(outputs=out.sum(), inputs=x_hat, create_graph=True)  # loss_functions.py:175
    norms = g.norm(2, dim=(1, 2, 3))
    gp = lam * torch.mean((norms - 1) ** 2)  # loss_functions.py:176
    return (gp, norms) if return_norms else gp


def g_w_loss(pg: Params, sg: NetSpec, pd: Params, sd: NetSpec, z: torch.Tensor):
    """G_W_loss.forward  (loss_functions.py:59-74) with the latent injected."""
    loss = -discriminator_forward(pd, generator_forward(pg, z, sg), sd).mean()  # loss_functions.py:64-67
    if torch.isnan(loss):
        raise ValueError("Generator loss is nan.")  # loss_functions.py:70-72
    return loss


# --------------------------------------------------------------------------------------
# one training iteration == train.py:357-385 with n_critic = 1, sim_loss off
# --------------------------------------------------------------------------------------
def make_adam(params: Params, lr: float = 1e-4, beta1: float = 0.5) -> torch.optim.Adam:
    """optim.Adam(net.parameters(), lr, betas=(beta1, 0.999))  (train.py:224-225; config.py:36, 40)."""
    leaves = [p for p in params.values() if p.requires_grad]
    return torch.optim.Adam(leaves, lr=lr, betas=(beta1, 0.999))


def zero_grads(params: Params) -> None:
    for p in params.values():
        p.grad = None


def train_step(pg: Params, sg: NetSpec, pd: Params, sd: NetSpec, opt_g, opt_d, real: torch.Tensor,
               z_d: torch.Tensor, z_gp: torch.Tensor, eps: torch.Tensor, z_g: torch.Tensor,
               lam: float = 10.0, drift_epsilon: float = 0.001) -> Dict[str, float]:
    """D step (W loss + drift + GP, backward, Adam) then G step (loss, backward, Adam)."""
    zero_grads(pd)  # train.py:357
    d_loss, s_real, s_fake = d_w_loss(pg, sg, pd, sd, real, z_d, drift_epsilon)  # train.py:358
    gp = grad_penalty(pg, sg, pd, sd, real, z_gp, eps, lam)  # train.py:361
    d_loss = d_loss + gp  # train.py:362
    d_loss.backward()  # train.py:365
    opt_d.step()  # train.py:366
    zero_grads(pg)  # train.py:375
    zero_grads(pd)  # (the reference lets these accumulate and clears them at the next D.zero_grad())
    g_loss = g_w_loss(pg, sg, pd, sd, z_g)  # train.py:376
    g_loss.backward()  # train.py:384
    opt_g.step()  # train.py:385
    return {"D_loss": float(d_loss.detach()), "score_real": float(s_real.detach()), "score_fake": float(s_fake.detach()),
            "GP": float(gp.detach()), "G_loss": float(g_loss.detach())}


def as_leaf_params(state: Dict[str, torch.Tensor], dtype: torch.dtype = torch.float32) -> Params:
    """Turn a state_dict (or npz dict) into grad-requiring leaves; 'alpha' stays a constant."""
    out: Params = {}
    for k, v in state.items():
        t = torch.as_tensor(v).detach().clone().to(dtype)
        if k != "alpha":
            t.requires_grad_(True)
        out[k] = t
    return out


# --------------------------------------------------------------------------------------
# algorithmic work model: conv + linear MACs x2 for one forward pass
# --------------------------------------------------------------------------------------
def forward_flops(g_widths: List[int], d_widths: List[int], image_size_init: int, res: int, latent_dim: int,
                  alpha: float = 1.0, n_colors: int = 1) -> Tuple[float, float]:
    """(F_G, F_D) in FLOP per image for one forward pass at stage `res` (2 x MAC; elementwise ignored)."""
    n_up = int(round(math.log2(res / image_size_init)))
    merged = n_up if alpha >= 1 else n_up - 1
    s0 = image_size_init
    fg = 2.0 * latent_dim * g_widths[0] * s0 * s0 + 2.0 * 9 * g_widths[0] * g_widths[0] * s0 * s0
    size = s0
    for i in range(merged):
        size *= 2
        fg += 2.0 * 9 * (g_widths[i] * g_widths[i + 1] + g_widths[i + 1] ** 2) * size * size
    fg += 2.0 * g_widths[merged] * n_colors * size * size
    if alpha < 1:
        size2 = size * 2
        fg += 2.0 * 9 * (g_widths[merged] * g_widths[merged + 1] + g_widths[merged + 1] ** 2) * size2 * size2
        fg += 2.0 * g_widths[merged + 1] * n_colors * size2 * size2
    # discriminator: widths listed from the highest resolution to the lowest
    nd = len(d_widths)
    fd = 2.0 * 9 * d_widths[-1] ** 2 * s0 * s0 + 2.0 * d_widths[-1] * s0 * s0
    size = s0
    for j in range(merged):
        cin, cout = d_widths[nd - 2 - j], d_widths[nd - 1 - j]
        fd += 2.0 * 9 * (cin * cout + cout * cout) * size * size
        size *= 2
    fd += 2.0 * n_colors * d_widths[nd - 1 - merged] * size * size
    if alpha < 1:
        cin, cout = d_widths[nd - 2 - merged], d_widths[nd - 1 - merged]
        fd += 2.0 * 9 * (cin * cout + cout * cout) * size * size
        fd += 2.0 * n_colors * cin * (2 * size) * (2 * size)
    return fg, fd
