"""The "bf16" mode (precision code 5 of include/ngan.h: bf16 ACTIVATION STORAGE, one bf16 MFMA per product group, fp32 accumulation /
PixelNorm statistics / master weights) -- BASELINE.json's C2 configuration, which the reference itself does not have
(/root/reference/train.py:136-144 computes in fp32).  Two kinds of statement:

* kernel level, through the C ABI: every bf16 kernel against an fp64 evaluation of the same operator ON THE bf16-ROUNDED OPERANDS
  (inputs as stored, weights as the packing kernel rounds them).  What is left is the fp32 accumulation order and the one rounding
  of the store: an output must be the bf16 neighbour of the fp64 value (<= 2^-7 relative, i.e. one bf16 ulp, + accumulation noise);
  fp32 outputs (norms, weight gradients, images) to 1e-4 / 1e-5.
* model level, against the reference's golden vectors, at this mode's OWN tolerance.  Where it comes from: tests/lowprec_budget.py
  emulates on the CPU oracle exactly what this mode rounds ("bf16 mode" rows: bf16 storage of every activation and activation
  gradient, bf16 conv weights with the scale folded in, one rounding of a resampled conv input) and reports the deviation from the
  fp32 reference per fixture.  Rounding noise through a LeakyReLU / PixelNorm net is chaotic from fixture to fixture, so the bound
  is the emulation's SPREAD over all fixtures (x 1.5), not a per-fixture prediction (DESIGN.md section 8 has the table):
      reduced-width nets (4 samples):  scalars 5e-2 of the step's largest scalar (emulated: up to 2.8e-2),  |grad D| 6e-2 (4.0e-2),
                                       parameter gradients 3e-1 relative L2 per net (D 1.4e-1, G 1.8e-1)
      full-width nets (C1, C2, C5):    scalars 1e-2 (emulated 6e-5: the penalty of an init-state net is insensitive),  |grad D| 3e-2
                                       (1.1e-2), per-tensor sum|g| checksums 2e-1 (relative L2 of the nets' gradients: 4.6e-2 / 1.2e-1)
  The fp32 mode's 1e-3 bar does not apply to this mode, and this mode is never the headline.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, split_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SLOPE = 0.2
BF = torch.bfloat16


@pytest.fixture
def bf16_mode(ngan):
    ngan.ops.set_conv_precision("bf16")
    yield ngan
    ngan.ops.set_conv_precision("f32")


def rbf(t):
    """round to bf16 and back: the values a bf16 tensor can hold"""
    return t.to(BF).to(t.dtype)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def bf16_close(got_bf16, want64, extra=0.0):
    """got (bf16 tensor) is the bf16 neighbour of want (fp64): |got - want| <= 2^-7 |want| + extra * max|want|"""
    got = got_bf16.detach().double().cpu()
    want = want64.detach().double().cpu()
    err = (got - want).abs()
    bound = 2.0 ** -7 * want.abs() + (extra + 1e-6) * want.abs().max()
    bad = (err > bound)
    return not bool(bad.any()), float((err / (want.abs().max() + 1e-30)).max())


def resample_ref(x, code):
    if code == 1:
        return rbf(F.avg_pool2d(x, 2).float()).double()             # the staging rounds the blended value once
    if code == 2:
        return rbf(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=None).float()).double()
    return x


def pn_ref(a):
    r = torch.sqrt(torch.mean(a * a, dim=1, keepdim=True) + 1e-8)
    return a / r, r


CONV = [
    # B, H, W, K, N, resample, bias -- every (K, N) pair, the three tile heights (PG = 4 needs >= 512 tiles of 8 x 32), ragged edges
    (1, 8, 32, 16, 16, 0, False), (2, 12, 20, 16, 32, 0, True), (1, 16, 16, 16, 64, 0, False), (1, 8, 8, 16, 128, 0, True),
    (2, 12, 20, 32, 16, 0, False), (1, 16, 48, 32, 32, 0, True), (1, 8, 8, 32, 64, 0, False), (1, 4, 4, 32, 128, 0, False),
    (1, 16, 16, 64, 16, 0, False), (1, 8, 8, 64, 32, 0, True), (2, 16, 16, 64, 64, 0, False), (1, 8, 8, 64, 128, 0, False),
    (1, 8, 8, 128, 16, 0, False), (1, 8, 8, 128, 32, 0, False), (1, 16, 16, 128, 64, 0, True), (3, 16, 16, 128, 128, 0, True),
    (4, 128, 256, 16, 16, 0, True), (4, 128, 256, 32, 32, 0, False), (4, 128, 256, 16, 32, 0, False), (4, 128, 256, 32, 16, 0, False),   # 8-row tiles
    (6, 64, 128, 32, 64, 0, False), (2, 200, 328, 16, 16, 0, False),                                                                        # 4-row tiles; ragged
    (2, 8, 8, 16, 16, 1, False), (1, 16, 16, 32, 32, 1, True), (3, 10, 20, 64, 32, 1, False), (4, 128, 128, 16, 32, 1, False),              # avg-pool on load
    (2, 8, 8, 32, 16, 2, False), (1, 32, 32, 16, 16, 2, True), (2, 12, 20, 128, 64, 2, True), (4, 128, 256, 32, 16, 2, False),              # bilinear x2 on load
    (128, 16, 16, 128, 128, 0, True), (16, 64, 64, 32, 64, 0, False), (130, 16, 12, 64, 128, 1, False), (16, 64, 64, 128, 64, 2, True),      # many-channel layers at BASELINE.json C2's sizes (64-pixel tiles of the output-tile split, >= 256 of them)
]


def conv_ref(x, w, bias, res, scale):
    """fp64 conv of the bf16-rounded operands: x as stored, weights as the packing kernel rounds them (scale folded in first)"""
    wq = rbf((w * scale).float()).double()
    return F.conv2d(resample_ref(x.double(), res), wq, bias.double() if bias is not None else None, padding=1)


@pytest.mark.parametrize("case", CONV)
def test_bf16_conv_forward_epilogues_against_fp64(bf16_mode, case):
    """ngan_bf16_conv3x3_fwd, epilogue 0 (plain) and 1 (LeakyReLU -> PixelNorm): outputs are the bf16 neighbours of the fp64 result
    on the rounded operands; the norm (fp32) to 2e-5."""
    ngan = bf16_mode
    ops = ngan.ops
    B, H, W, K, N, res, use_bias = case
    torch.manual_seed(hash(case) % 1000)
    hin, win = (2 * H, 2 * W) if res == 1 else ((H // 2, W // 2) if res == 2 else (H, W))
    x = rbf(torch.randn(B, K, hin, win))
    w = torch.randn(N, K, 3, 3)
    bias = torch.randn(N) * 0.5 if use_bias else None
    scale = 1.3868 / np.sqrt(9 * K)
    xd = nhwc(x).to(DEV).to(BF)
    wd, bd = w.to(DEV), (bias.to(DEV) if use_bias else None)
    c_ref = conv_ref(x, w, bias, res, scale)
    y0, _ = ops._run_conv(xd, wd, bd, res, scale, 0, 0.0)
    assert y0.dtype == BF and tuple(y0.shape) == (B, H, W, N)
    ok, worst = bf16_close(y0.permute(0, 3, 1, 2), c_ref, extra=2e-5)
    assert ok, ("plain", worst)
    y1, rn = ops._run_conv(xd, wd, bd, res, scale, 1, SLOPE)
    # LeakyReLU with the kernel's own activation pattern (ties at rounding level may fall either way; tests/test_gpu_ops.py header)
    mask = (y1.permute(0, 3, 1, 2).double().cpu() > 0).double()
    a = c_ref * (mask + SLOPE * (1 - mask))
    y_ref, r_ref = pn_ref(a)
    assert rn.dtype == torch.float32
    assert float((rn.double().cpu() - r_ref[:, 0]).abs().max() / r_ref.max()) < 2e-5
    ok, worst = bf16_close(y1.permute(0, 3, 1, 2), y_ref, extra=2e-5)
    assert ok, ("lrelu_pn", worst)


DGRAD = [
    # B, H, W (conv resolution), Cout (= contraction), Cin (= outputs), resample of the forward conv
    (2, 12, 20, 16, 16, 0), (1, 16, 16, 32, 64, 0), (4, 128, 256, 16, 16, 0), (4, 128, 256, 32, 16, 0), (1, 8, 8, 128, 128, 0),
    (2, 8, 8, 16, 16, 1), (4, 64, 128, 32, 32, 1), (2, 12, 12, 64, 32, 1),            # pool-adjoint store
    (2, 8, 8, 32, 16, 2), (2, 64, 64, 16, 32, 2),                                     # bilinear adjoint as a second launch
    (128, 16, 16, 128, 128, 0), (16, 64, 64, 64, 64, 0), (64, 16, 16, 128, 64, 1),    # the same for the input gradient
]


@pytest.mark.parametrize("case", DGRAD)
def test_bf16_input_gradient_with_pixelnorm_backward_epilogue(bf16_mode, case):
    """the input-gradient call (flipped packed weights), plain and with the producer's LeakyReLU -> PixelNorm backward fused
    (epilogue 2; behind a pooled conv: at the four pixels of each window; behind a bilinear conv: in the adjoint kernel)"""
    ngan = bf16_mode
    ops = ngan.ops
    B, H, W, Co, Ci, res = case
    torch.manual_seed(sum(case))
    hin, win = (2 * H, 2 * W) if res == 1 else ((H // 2, W // 2) if res == 2 else (H, W))
    g = rbf(torch.randn(B, Co, H, W))
    w = torch.randn(Co, Ci, 3, 3)
    scale = 1.3868 / np.sqrt(9 * Ci)
    yprev = rbf(torch.randn(B, Ci, hin, win))                      # stands for the producer's output
    rprev = torch.rand(B, hin, win) + 0.5
    wq = rbf((w * scale).float()).double()
    gd = nhwc(g).to(DEV).to(BF)
    # reference: d/dx of <conv(resample(x)), g>, with the intermediate roundings of the kernel sequence
    full = F.conv_transpose2d(g.double(), wq, padding=1)           # gradient w.r.t. the conv input at conv resolution
    if res == 1:
        ref = F.interpolate(full, scale_factor=2, mode="nearest") * 0.25
    elif res == 2:
        xs = torch.zeros(B, Ci, hin, win, dtype=torch.float64, requires_grad=True)
        up = F.interpolate(xs, scale_factor=2, mode="bilinear", align_corners=None)
        ref, = torch.autograd.grad(up, xs, rbf(full.float()).double())      # the conv's output is stored (bf16) before the adjoint
    else:
        ref = full
    gx = ops._run_dgrad(gd, w.to(DEV), res, scale)
    assert gx.dtype == BF and tuple(gx.shape) == (B, hin, win, Ci)
    ok, worst = bf16_close(gx.permute(0, 3, 1, 2), ref, extra=3e-5)
    assert ok, ("plain", worst)
    link = ops.PNLink()
    link.y, link.rn, link.slope = nhwc(yprev).to(DEV).to(BF), rprev.to(DEV), SLOPE
    gl = ops._run_dgrad(gd, w.to(DEV), res, scale, link=link)
    yy, rr = yprev.double(), rprev.double()[:, None]
    s = (ref * yy).mean(dim=1, keepdim=True)
    m = torch.where(yy > 0, torch.ones_like(yy), torch.full_like(yy, SLOPE))
    want = m * (ref - yy * s) / rr
    ok, worst = bf16_close(gl.permute(0, 3, 1, 2), want, extra=1e-4)
    assert ok, ("pn_bwd", worst)


def test_bf16_conv_to_image_epilogue(bf16_mode):
    """epilogue 3: conv + LeakyReLU + PixelNorm + ToImage(tanh) in one launch, with and without the stored activation"""
    ngan = bf16_mode
    ops, C = ngan.ops, ngan._C
    for (B, H, W, K, N) in [(2, 24, 40, 16, 16), (4, 128, 256, 16, 16), (2, 16, 16, 32, 32)]:
        torch.manual_seed(B + H)
        x = rbf(torch.randn(B, K, H, W))
        w, wimg = torch.randn(N, K, 3, 3), torch.randn(1, N, 1, 1) * 0.3
        scale = 1.3868 / np.sqrt(9 * K)
        xd = nhwc(x).to(DEV).to(BF)
        assert ops.to_image_fusable(xd, w, wimg, 0)
        t = ops.ConvLReLUPNToImage.apply(xd, w.to(DEV), None, wimg.to(DEV), 0, scale, SLOPE, None, False)
        y1, _ = ops._run_conv(xd, w.to(DEV), None, 0, scale, 1, SLOPE)
        c = conv_ref(x, w, None, 0, scale)
        mask = (y1.permute(0, 3, 1, 2).double().cpu() > 0).double()
        yr, _ = pn_ref(c * (mask + SLOPE * (1 - mask)))
        want = torch.tanh((yr * wimg.double()[0][None]).sum(dim=1))
        assert t.dtype == torch.float32
        assert float((t[..., 0].double().cpu() - want).abs().max()) < 2e-5      # the image comes from the UNROUNDED activation


WGRAD = [(2, 12, 20, 16, 16, 0), (1, 16, 16, 32, 64, 0), (3, 8, 8, 128, 128, 0), (4, 128, 256, 16, 16, 0), (4, 64, 128, 32, 32, 0),
         (2, 8, 8, 16, 32, 1), (4, 64, 64, 32, 32, 1), (2, 16, 16, 64, 32, 2), (2, 128, 128, 16, 16, 2), (1, 4, 4, 64, 64, 0),
         (2, 20, 64, 16, 32, 0), (1, 8, 32, 32, 16, 0), (3, 36, 96, 16, 16, 0)]      # whole 32-pixel tiles (copied staging): ragged height, a single tile, three columns


@pytest.mark.parametrize("case", WGRAD)
def test_bf16_weight_gradient_against_fp64(bf16_mode, case):
    """ngan_bf16_conv3x3_wgrad: bf16 x and g, exact products, fp32 accumulation: 1e-4 relative L2 against fp64 on the same operands"""
    ngan = bf16_mode
    ops = ngan.ops
    B, H, W, Ci, Co, res = case
    torch.manual_seed(sum(case) + 3)
    hin, win = (2 * H, 2 * W) if res == 1 else ((H // 2, W // 2) if res == 2 else (H, W))
    x = rbf(torch.randn(B, Ci, hin, win))
    g = rbf(torch.randn(B, Co, H, W))
    scale = 0.37
    xin = resample_ref(x.double(), res)
    wz = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(F.conv2d(xin, wz, padding=1), wz, g.double())
    got = ops._run_wgrad(nhwc(x).to(DEV).to(BF), nhwc(g).to(DEV).to(BF), res, scale)
    assert got.dtype == torch.float32
    err = float((got.double().cpu() - scale * ref).norm() / (scale * ref).norm())
    assert err < 1e-4, err
    acc = torch.ones(Co, Ci, 3, 3, device=DEV)
    ops._run_wgrad(nhwc(x).to(DEV).to(BF), nhwc(g).to(DEV).to(BF), res, scale, accumulate_into=acc)
    assert float((acc.double().cpu() - 1 - scale * ref).norm() / (scale * ref).norm()) < 1e-4


def test_bf16_pointwise_operators_round_the_fp32_results(bf16_mode):
    """Every per-pixel operator in bf16 storage against ITS OWN fp32 twin on bf16-representable inputs: same arithmetic between load and
    store, so a bf16 output is the rounding of the fp32 output (<= 1 ulp), an fp32 output (norms, images, parameter gradients) equal
    to ~1e-6."""
    ngan = bf16_mode
    C_ = ngan._C
    torch.manual_seed(11)
    npix, Cc = 3 * 20 * 12, 32

    def pair(*shape):
        v = rbf(torch.randn(*shape)).to(DEV)
        return v, v.to(BF)

    def check(name, outs32, outs16):
        for i, (a, b) in enumerate(zip(outs32, outs16)):
            if b.dtype == BF:
                ok, worst = bf16_close(b, a.double(), extra=1e-6)
                assert ok, (name, i, worst)
            else:
                assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-7, (name, i)

    x32, x16 = pair(npix, Cc)
    g32, g16 = pair(npix, Cc)
    h32, h16 = pair(npix, Cc)
    bias = torch.randn(Cc, device=DEV)
    outs = {}
    for tag, (x, g, h, dt) in {"f": (x32, g32, h32, torch.float32), "b": (x16, g16, h16, BF)}.items():
        pre = "ngan_" if tag == "f" else "ngan_bf16_"
        y, rn = torch.empty(npix, Cc, device=DEV, dtype=dt), torch.empty(npix, device=DEV)
        C_.call(pre + "lrelu_pixelnorm_fwd", x, bias, y, rn, npix, Cc, SLOPE, 1e-8)
        outs[tag] = [y, rn]
    check("pn_fwd", outs["f"], outs["b"])
    # backward / backward-of-backward on a shared (bf16-representable) y and norm
    y32 = rbf(outs["f"][0])
    y16 = y32.to(BF)
    rn = outs["f"][1]
    gr = torch.randn(npix, device=DEV)
    for name, n_out in (("bwd2", 1), ("bwdbwd", 3)):
        res = {}
        for tag, (y, g, h, dt) in {"f": (y32, g32, h32, torch.float32), "b": (y16, g16, h16, BF)}.items():
            pre = "ngan_" if tag == "f" else "ngan_bf16_"
            if name == "bwd2":
                gc = torch.empty(npix, Cc, device=DEV, dtype=dt)
                C_.call(pre + "lrelu_pixelnorm_bwd2", g, h, gr, y, rn, gc, npix, Cc, SLOPE)
                res[tag] = [gc]
            else:
                a, b2, c = torch.empty(npix, Cc, device=DEV, dtype=dt), torch.empty(npix, Cc, device=DEV, dtype=dt), torch.empty(npix, device=DEV)
                C_.call(pre + "lrelu_pixelnorm_bwdbwd", h, g, y, rn, a, b2, c, npix, Cc, SLOPE)
                res[tag] = [a, b2, c]
        check(name, res["f"], res["b"])
    # resampling, fade-in, channel sums on a (B, h, w, C) tensor
    B, hh, ww = 3, 10, 6
    f32, f16 = pair(B, 2 * hh, 2 * ww, Cc)
    l32, l16 = pair(B, hh, ww, Cc)
    alpha = torch.tensor([0.37], device=DEV)
    for tag, (hi, lo, dt) in {"f": (f32, l32, torch.float32), "b": (f16, l16, BF)}.items():
        pre = "ngan_" if tag == "f" else "ngan_bf16_"
        o = {}
        o["pool2_fwd"] = torch.empty(B, hh, ww, Cc, device=DEV, dtype=dt); C_.call(pre + "pool2_fwd", hi, o["pool2_fwd"], B, hh, ww, Cc)
        o["up2_adjoint"] = torch.empty(B, hh, ww, Cc, device=DEV, dtype=dt); C_.call(pre + "up2_adjoint", hi, o["up2_adjoint"], B, hh, ww, Cc)
        o["up2_fwd"] = torch.empty(B, 2 * hh, 2 * ww, Cc, device=DEV, dtype=dt); C_.call(pre + "up2_fwd", lo, o["up2_fwd"], B, hh, ww, Cc)
        o["pool2_adjoint"] = torch.empty(B, 2 * hh, 2 * ww, Cc, device=DEV, dtype=dt); C_.call(pre + "pool2_adjoint", lo, o["pool2_adjoint"], B, hh, ww, Cc)
        o["lerp"] = torch.empty_like(lo); C_.call(pre + "lerp", lo, o["pool2_fwd"], alpha, o["lerp"], lo.numel())
        o["fade_a"], o["fade_b"] = torch.empty_like(lo), torch.empty_like(lo)
        C_.call(pre + "fade_bwd", lo, alpha, o["fade_a"], o["fade_b"], lo.numel())
        o["csum"] = torch.empty(Cc, device=DEV); ws = torch.empty(1024 * Cc, device=DEV)
        C_.call(pre + "channel_sum", lo, o["csum"], ws, B * hh * ww, Cc, 1.0)
        rnl = torch.rand(B, hh, ww, generator=torch.Generator().manual_seed(5)).to(DEV) + 0.5
        o["up2_adjoint_pnbwd"] = torch.empty(B, hh, ww, Cc, device=DEV, dtype=dt)
        C_.call(pre + "up2_adjoint_pnbwd", hi, lo, rnl, o["up2_adjoint_pnbwd"], B, hh, ww, Cc, SLOPE)
        outs[tag] = o
    for k in outs["f"]:
        extra = [outs["f"][k]], [outs["b"][k]]
        if k == "lerp":        # its second operand is the pooled tensor, which the bf16 run rounded once already: 2 ulp
            ok, worst = bf16_close(outs["b"][k], outs["f"][k].double(), extra=2.0 ** -8)
            assert ok, (k, worst)
        else:
            check(k, *extra)


def test_bf16_image_edge_operators(bf16_mode):
    """FromImage / ToImage / stem / critic head between fp32 images (latents, scores) and bf16 features: against their fp32 twins"""
    ngan = bf16_mode
    ops = ngan.ops
    torch.manual_seed(3)
    B, H, W, C = 3, 12, 20, 32
    img = torch.randn(B, H, W, 1, device=DEV)
    wf, bfm = torch.randn(C, 1, 1, 1, device=DEV), torch.randn(C, device=DEV)
    g = rbf(torch.randn(B, H, W, C)).to(DEV)

    def both(fn):
        out = {}
        for mode in ("f32", "bf16"):
            ops.set_conv_precision(mode)
            out[mode] = fn(mode)
        ops.set_conv_precision("bf16")
        return out["f32"], out["bf16"]

    def close(a, b, tag, fp32_tol=2e-5):
        for i, (u, v) in enumerate(zip(a, b)):
            if v.dtype == BF:
                ok, worst = bf16_close(v, u.double(), extra=1e-6)
                assert ok, (tag, i, worst)
            else:
                assert float((u - v).abs().max()) <= fp32_tol * float(u.abs().max()) + 1e-7, (tag, i)

    def from_image(mode):
        gg = g if mode == "f32" else g.to(BF)
        x = img.clone().requires_grad_()
        w, b = wf.clone().requires_grad_(), bfm.clone().requires_grad_()
        y = ops.FromImage.apply(x, w, b, False)
        gx, gw, gb = torch.autograd.grad(y, [x, w, b], gg)
        return [y, gx, gw, gb]
    close(*both(from_image), "from_image")

    def to_image(mode):
        x = (g if mode == "f32" else g.to(BF)).clone().requires_grad_()
        w = torch.randn(1, C, 1, 1, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).requires_grad_()
        t = ops.ToImage.apply(x, w)
        gx, gw = torch.autograd.grad(t, [x, w], torch.ones_like(t) * 0.3)
        return [t, gx, gw]
    close(*both(to_image), "to_image")

    K, S, Cs = 64, 4, 32
    z = torch.randn(B, K, device=DEV)
    Wst = torch.randn(Cs * S * S, K, device=DEV)
    gst = rbf(torch.randn(B, S, S, Cs)).to(DEV)

    def stem(mode):
        zz, w = z.clone().requires_grad_(), Wst.clone().requires_grad_()
        with ops.first_order_only():
            y, rn = ops.LinearLReLUPN.apply(zz, w, S, 0.05, SLOPE)
        return [y, rn]
    close(*both(stem), "stem_fwd")
    # stem backward on a shared bf16-representable (y, gc): the weight gradient and the latent gradient are fp32 in both modes
    for fn in ("ngan_linear_wgrad", "ngan_linear_dgrad"):
        res = []
        for gc in (gst, gst.to(BF)):
            if fn.endswith("wgrad"):
                out = torch.empty_like(Wst)
                ngan._C.call(ops._k(fn, gc), z, gc, out, B, K, S * S, Cs, 0.05)
            else:
                out = torch.empty_like(z)
                ngan._C.call(ops._k(fn, gc), gc, Wst, out, B, K, S * S, Cs, 0.05)
            res.append(out)
        assert float((res[0] - res[1]).abs().max()) <= 2e-5 * float(res[0].abs().max()), fn

    yh = rbf(torch.randn(B, 4, 4, 64)).to(DEV)
    wh, bh = torch.randn(1, 64, 4, 4, device=DEV), torch.randn(1, device=DEV)

    def head(mode):
        y = (yh if mode == "f32" else yh.to(BF)).clone().requires_grad_()
        w, b = wh.clone().requires_grad_(), bh.clone().requires_grad_()
        o = ops.FinalDot.apply(y, w, b, 0.01)
        gy, gw, gb = torch.autograd.grad(o, [y, w, b], torch.tensor([[0.5], [-1.0], [2.0]], device=DEV))
        return [o, gy, gw, gb]
    close(*both(head), "head")


SMALL = ["small_fresh4", "small_res8_init", "small_res8_warm", "small_res16_fade_init", "small_res16_fade_warm", "small_res16_warm"]
# this mode's own tolerances (module docstring): reduced-width nets / full-width nets
TOL_SCALAR, TOL_NORM, TOL_GRAD = 5e-2, 6e-2, 3e-1
FULL_TOL_SCALAR, FULL_TOL_NORM, FULL_TOL_GRAD = 1e-2, 3e-2, 2e-1


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("name", SMALL)
def test_bf16_small_nets_against_the_reference_at_the_modes_own_tolerance(bf16_mode, name):
    import test_gpu_models as M
    ngan = bf16_mode
    fix = load_golden(name)
    G, D = M.build_small(ngan, fix)
    with torch.no_grad():
        img = G(torch.from_numpy(fix["z_d"]).to(DEV)).cpu().numpy()
        score = D(torch.from_numpy(fix["real"]).to(DEV)).cpu().numpy()
    assert img.dtype == np.float32 and M.rel(img, fix["G_of_z_d"]) < 5e-2
    assert M.rel(score, fix["D_of_real"]) < 5e-2
    scal, norms, dgrads, ggrads = M.run_step_losses(ngan, G, D, fix)
    want = fix["scalars"]
    # losses and scores: relative to the largest scalar of the step (an init-state score is ~1e-3 next to a penalty of ~10)
    assert float(np.abs(scal - want).max()) < TOL_SCALAR * float(np.abs(want).max()), (scal, want)
    assert M.rel(norms, fix["grad_norms"]) < TOL_NORM, M.rel(norms, fix["grad_norms"])
    # parameter gradients: relative L2 over the whole net (per-tensor numbers in the failure message)
    for tag, got, wantd in (("D", dgrads, split_state(fix, "Dgrad/")), ("G", ggrads, split_state(fix, "Ggrad/"))):
        assert set(got) == set(wantd)
        flat_g = np.concatenate([got[k].ravel() for k in sorted(got)])
        flat_w = np.concatenate([wantd[k].ravel() for k in sorted(got)])
        per = {k: round(rel_l2(got[k], wantd[k]), 4) for k in got}
        assert rel_l2(flat_g, flat_w) < TOL_GRAD, (tag, rel_l2(flat_g, flat_w), per)


@pytest.mark.parametrize("name", ["full_C1", "full_C2", "full_C5"])
def test_bf16_full_width_configs(bf16_mode, name):
    """BASELINE.json's C2 (64x64, batch 64, alpha 0.5: the configuration it names "bf16") and C5's shape (512x512, batch 8), plus C1,
    against the reference's first-step values -- scalars, |grad D| per sample, and the per-tensor sum|g| checksums of both nets' gradients
    (the generator's taken BEFORE the critic's update)."""
    ngan = bf16_mode
    import test_gpu_models as M
    fix = load_golden(name)
    res, alpha, init, latent, batch, lr = fix["meta"]
    res, batch = int(res), int(batch)
    cfg = ngan.config
    torch.manual_seed(1)
    G = ngan.models.Generator_PG(cfg.N_gen_features, image_size_init=16)
    D = ngan.models.Discriminator_PG(cfg.N_dis_features, image_size_init=16)
    if res != 16:
        G.set_resolution(res, float(alpha))
        D.set_resolution(res, float(alpha))
    torch.manual_seed(123)
    x = torch.rand(batch, 1, res, res) * 2 - 1
    G.to(DEV)
    D.to(DEV)
    fx = dict(fix)
    fx["real"] = x.numpy()
    with torch.no_grad():
        img = G(torch.from_numpy(fix["z_d"]).to(DEV))
        assert M.rel(img[:2, 0, :8, :8].cpu().numpy(), fix["G_of_z_d_slice"]) < 3e-2
        score = D(x.to(DEV)).cpu().numpy()
    assert M.rel(score, fix["D_of_real"]) < 5e-2
    G.zero_grad()
    g_pre, _ = ngan.loss_functions.G_W_loss(G, D)(x.to(DEV), z=torch.from_numpy(fix["z_g"]).to(DEV))
    g_pre.backward()
    worst_g = {}
    for k, p in G.named_parameters():
        if p.grad is not None:
            cs = fix["cs/Ggrad_pre/" + k]
            worst_g[k] = abs(float(p.grad.double().abs().sum()) - cs[1]) / cs[1]
    assert max(worst_g.values()) < FULL_TOL_GRAD, worst_g
    G.zero_grad()
    D.zero_grad()
    scal, norms, dgrads, ggrads = M.run_step_losses(ngan, G, D, fx)
    want = fix["scalars"]
    assert float(np.abs(scal - want).max()) < FULL_TOL_SCALAR * float(np.abs(want).max()), (scal, want)
    assert M.rel(norms, fix["grad_norms"]) < FULL_TOL_NORM, M.rel(norms, fix["grad_norms"])
    worst_d = {k: abs(float(np.abs(g.astype(np.float64)).sum()) - fix["cs/Dgrad/" + k][1]) / fix["cs/Dgrad/" + k][1] for k, g in dgrads.items()}
    big = {k: v for k, v in worst_d.items() if fix["cs/Dgrad/" + k][1] > 1e-6}       # (the head bias: a cancelling sum, see test_gpu_models)
    assert max(big.values()) < FULL_TOL_GRAD, big


def test_bf16_step_driver_replay_equals_eager(ngan):
    """The whole step driver in the bf16 mode: a captured HIP graph replayed three times leaves bit-identical parameters to three eager
    iterations on the same draws (the mode is as reproducible as fp32: fixed-order reductions, no atomics); parameters and Adam state stay
    fp32; the first iteration's loss statistics (same initial weights, same draws) are the fp32 mode's within the mode's tolerance.
    (Parameters after several Adam steps are NOT compared across the modes: Adam's first steps move every weight by ~lr times the SIGN of
    its gradient, and the sign of a near-zero gradient is arbitrary in any arithmetic -- tools/step2_sensitivity.py.)"""
    def make():
        torch.manual_seed(5)
        G = ngan.models.Generator_PG([32, 16, 16], image_size_init=8, latent_dim=32)
        D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=8)
        G.set_resolution(32, 1.0)
        D.set_resolution(32, 1.0)
        return ngan.train.PGGANTrainer(G.to(DEV), D.to(DEV), learning_rate=1e-3)
    gen = torch.Generator().manual_seed(9)
    def draw(b=8):
        z = [torch.randn(b, 32, generator=gen) for _ in range(3)]
        z = [(v / v.norm(dim=1, keepdim=True)).to(DEV) for v in z]
        return dict(real=(torch.rand(b, 1, 32, 32, generator=gen) * 2 - 1).to(DEV), z_d=z[0], z_gp=z[1],
                    eps=torch.rand(b, 1, 1, 1, generator=gen).to(DEV), z_g=z[2])
    seq = [draw() for _ in range(3)]
    ngan.ops.set_conv_precision("f32")
    s0 = seq[0]
    ref_stats = {k: float(v) for k, v in make().train_iteration(s0["real"], s0["z_d"], s0["z_gp"], s0["eps"], s0["z_g"]).items()}
    try:
        ngan.ops.set_conv_precision("bf16")
        assert ngan.ops.act_dtype() == BF
        eager, tr = make(), make()
        static = {k: seq[0][k].clone() for k in ("z_d", "z_gp", "eps", "z_g")}
        tr.capture(seq[0]["real"], draws=static)
        first = None
        for s in seq:
            st = eager.train_iteration(s["real"], s["z_d"], s["z_gp"], s["eps"], s["z_g"])
            first = first or {k: float(v) for k, v in st.items()}
            for k, v in static.items():
                v.copy_(s[k])
            tr.replay(s["real"])
        torch.cuda.synchronize()
        for name, p, pe in zip(tr.flat_g.names + tr.flat_d.names, tr.flat_g.params + tr.flat_d.params, eager.flat_g.params + eager.flat_d.params):
            assert p.dtype == torch.float32 and bool(torch.isfinite(p).all())
            assert torch.equal(p, pe), f"{name}: replay differs from eager by {float((p - pe).abs().max())}"
        for flat in (tr.flat_g, tr.flat_d):
            assert flat.exp_avg.dtype == torch.float32 and flat.exp_avg_sq.dtype == torch.float32
        scale = max(1.0, max(abs(v) for v in ref_stats.values()))
        for k, v in ref_stats.items():
            assert abs(first[k] - v) <= 5e-2 * scale, f"{k}: {first[k]} in the bf16 mode, {v} in fp32"
    finally:
        ngan.ops.set_conv_precision("f32")
