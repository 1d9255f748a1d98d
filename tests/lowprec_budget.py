"""Error budget of the reduced-precision configurations BASELINE.json names (C2: bf16 storage; C5: fp8 convolutions with bf16
accumulation) -- a CPU emulation on the oracle, NOT a product path (the product computes and stores fp32; DESIGN.md row g).

    python tests/lowprec_budget.py            # prints the table DESIGN.md quotes

What is emulated (straight-through rounding ops inserted into oracle/pggan_oracle.py's functions, forward AND backward, closed
under double-backward so the gradient penalty sees them too):
  bf16 storage   every tensor a fused kernel would write to HBM -- block outputs after PixelNorm, FromImage / ToImage outputs, and
                 the gradients flowing through those same points -- is rounded to bf16; arithmetic inside a layer stays fp32
  fp8 convs      additionally the operands of every 3x3 / full-extent convolution (activations and weights, per-tensor scaled to the
                 e4m3 range; gradients in e5m2) are rounded to fp8 and the convolution's result to bf16 (bf16 accumulation is
                 modelled by ONE final rounding, which flatters it)
  bf16 mode      (round 4) what the product's "bf16" mode (precision code 5, csrc/conv3x3_bf16.hip) rounds: bf16 storage as above, PLUS
                 the 3x3 convolutions' weights rounded to bf16 with the equalised-LR scale folded in first (fp32 masters, straight-
                 through gradient: the weight gradient is an fp32 sum of bf16 x bf16 products), PLUS one rounding of a resampled
                 conv input (the kernel blends the 2x2 / bilinear taps in fp32 and stores the blend in its bf16 LDS tile).
                 Accumulation, PixelNorm statistics, biases and the 1x1 / linear / head layers stay fp32.  This row is where the
                 mode's own test tolerances come from (tests/test_gpu_bf16.py).
Reported: relative error of the quantities the north star bounds at 1e-3 (losses, D(x), |grad D|), and of the parameter gradients.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, split_state  # noqa: E402
from oracle import pggan_oracle as O  # noqa: E402


class Round(torch.autograd.Function):
    """y = round_to(x) with a straight-through gradient that is rounded the same way (the gradient tensor is stored in that format too)"""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.kinds = (fwd, bwd)
        return quantize(x, fwd)

    @staticmethod
    def backward(ctx, g):
        fwd, bwd = ctx.kinds
        return Round.apply(g, bwd, fwd), None, None


def quantize(x, kind):
    if kind == "bf16":
        return x.to(torch.bfloat16).to(x.dtype)
    if kind in ("e4m3", "e5m2"):
        dt, top = (torch.float8_e4m3fn, 448.0) if kind == "e4m3" else (torch.float8_e5m2, 57344.0)
        s = top / (x.detach().abs().max().clamp_min(1e-30) * 2.0)     # per-tensor scale with one binade of headroom
        return (x * s).to(dt).to(x.dtype) / s
    return x


def install(mode):
    """patch the oracle's primitives; returns a function that restores them"""
    saved = {k: getattr(O, k) for k in ("pixel_norm", "scaled_conv", "from_image", "to_image")}
    modes_known = ("f32", "bf16", "bf16mode", "fp8")
    assert mode in modes_known, mode
    if mode == "f32":
        return lambda: None
    store = lambda t: Round.apply(t, "bf16", "bf16")

    def pixel_norm(x, eps=O.PIXELNORM_EPS):
        return store(saved["pixel_norm"](x, eps))

    def scaled_conv(x, w, b, slope, padding):
        if mode == "bf16mode" and w.shape[2] == 3:
            fan_in = w.shape[1] * w.shape[2] * w.shape[3]
            sc = O.weight_scale(fan_in, slope)
            wq = Round.apply(w * sc, "bf16", "none") / sc           # the packing kernel rounds scale * W; its gradient is not rounded
            return saved["scaled_conv"](x, wq, b, slope, padding)
        if mode == "fp8":
            x, w = Round.apply(x, "e4m3", "e5m2"), Round.apply(w, "e4m3", "e5m2")
            return store(saved["scaled_conv"](x, w, b, slope, padding))
        return saved["scaled_conv"](x, w, b, slope, padding)

    O.pixel_norm, O.scaled_conv = pixel_norm, scaled_conv
    if mode == "bf16mode":
        saved["up2"], saved["pool2"] = O.up2, O.pool2
        O.up2 = lambda x: store(saved["up2"](x))
        O.pool2 = lambda x: store(saved["pool2"](x))
    O.from_image = lambda x, w, b: store(saved["from_image"](x, w, b))
    O.to_image = lambda x, w: store(saved["to_image"](x, w))

    def restore():
        for k, v in saved.items():
            setattr(O, k, v)
    return restore


def one_step(pg, pd, spec, x, z1, z2, eps, z3):
    d_loss, s_r, s_f = O.d_w_loss(pg, spec, pd, spec, x, z1, 0.001)
    gp, norms = O.grad_penalty(pg, spec, pd, spec, x, z2, eps, 10.0, return_norms=True)
    (d_loss + gp).backward()
    dg = torch.cat([v.grad.reshape(-1) for v in pd.values() if v.grad is not None]).clone()
    for v in pd.values():
        v.grad = None
    g_loss = O.g_w_loss(pg, spec, pd, spec, z3)
    g_loss.backward()
    gg = torch.cat([v.grad.reshape(-1) for v in pg.values() if v.grad is not None]).clone()
    for v in list(pg.values()) + list(pd.values()):
        v.grad = None
    sc = dict(D_loss=float(d_loss), score_real=float(s_r), score_fake=float(s_f), GP=float(gp), G_loss=float(g_loss))
    return sc, norms.detach().clone(), dg, gg


def budget(name, pg, pd, spec, x, z1, z2, eps, z3):
    out = {}
    for mode in ("f32", "bf16", "bf16mode", "fp8"):
        restore = install(mode)
        try:
            out[mode] = one_step(pg, pd, spec, x, z1, z2, eps, z3)
        finally:
            restore()
    ref = out["f32"]
    rows = {}
    for mode in ("bf16", "bf16mode", "fp8"):
        sc, norms, dg, gg = out[mode]
        r = {k: abs(sc[k] - ref[0][k]) / max(abs(ref[0][k]), 1e-12) for k in sc}
        r["scalars / max scalar"] = max(abs(sc[k] - ref[0][k]) for k in sc) / max(abs(v) for v in ref[0].values())
        r["|grad D|"] = float(((norms - ref[1]).abs() / ref[1].abs()).max())
        r["D grads (rel L2)"] = float((dg - ref[2]).norm() / ref[2].norm())
        r["G grads (rel L2)"] = float((gg - ref[3]).norm() / ref[3].norm())
        rows[mode] = r
    return rows


def small_case(fixture):
    fix = load_golden(fixture)
    res, alpha, init = int(fix["meta"][0]), float(fix["meta"][1]), int(fix["meta"][2])
    pg, pd = O.as_leaf_params(split_state(fix, "G/")), O.as_leaf_params(split_state(fix, "D/"))
    t = lambda k: torch.from_numpy(fix[k])
    return pg, pd, O.NetSpec(image_size_init=init, slope=0.2, alpha=alpha), t("real"), t("z_d"), t("z_gp"), t("eps"), t("z_g")


def full_case(res, alpha, batch):
    from __graft_entry__ import load_package
    ngan = load_package()
    torch.manual_seed(1)
    G = ngan.models.Generator_PG(ngan.config.N_gen_features, image_size_init=16)
    D = ngan.models.Discriminator_PG(ngan.config.N_dis_features, image_size_init=16)
    if res != 16:
        G.set_resolution(res, alpha)
        D.set_resolution(res, alpha)
    pg = O.as_leaf_params({k: v.detach().clone() for k, v in G.state_dict().items()})
    pd = O.as_leaf_params({k: v.detach().clone() for k, v in D.state_dict().items()})
    torch.manual_seed(123)
    x = torch.rand(batch, 1, res, res) * 2 - 1
    z = [O.sample_latent_vec((batch, 512)) for _ in range(3)]
    return pg, pd, O.NetSpec(image_size_init=16, slope=0.2, alpha=alpha), x, z[0], z[1], torch.rand(batch, 1, 1, 1), z[2]


CASES = {"small 16x16 warmed (|grad D| = O(1))": lambda: small_case("small_res16_warm"),
         "small 16x16 fade-in warmed": lambda: small_case("small_res16_fade_warm"),
         "small 4x4 fresh": lambda: small_case("small_fresh4"), "small 8x8 init": lambda: small_case("small_res8_init"),
         "small 8x8 warmed": lambda: small_case("small_res8_warm"), "small 16x16 fade-in init": lambda: small_case("small_res16_fade_init"),
         "C1 shape: full widths, 16x16, batch 16": lambda: full_case(16, 1.0, 16),
         "C2 shape: full widths, 64x64 alpha 0.5, batch 16": lambda: full_case(64, 0.5, 16)}


def main():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    for name, make in CASES.items():
        rows = budget(name, *make())
        print(name)
        for mode, r in rows.items():
            label = {"bf16": "bf16 storage", "bf16mode": "bf16 mode (storage + bf16 conv weights)", "fp8": "fp8 convs + bf16 accumulate/storage"}[mode]
            print(f"    {label:38s} " + "  ".join(f"{k} {v:.1e}" for k, v in r.items()))
    return 0


if __name__ == "__main__":
    sys.exit(main())
