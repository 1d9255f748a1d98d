"""Op-level parity of the HIP kernels (through the C ABI) against plain PyTorch fp64 CPU references of the same
ops: forward, first-order gradients and the gradient-penalty-style second order (d/dW of |dL/dx|^2).
Tolerances (fp32 / split-bf16 kernels vs fp64 reference): forward max-norm relative error <= 2e-4; gradients relative L2
error <= 2e-4 (first order) / 1e-3 (second order) with at most 1e-4 of the elements off by more than 1e-2 of the max-norm.
LeakyReLU's derivative is discontinuous at 0: an activation whose pre-activation is smaller than the arithmetic's rounding
error (a few out of ~1e6 elements) gets the other slope in one of two implementations (also between torch fp32 and fp64),
and because a weight gradient is a random-walk sum over all pixels, a handful of such flips already moves it by ~1e-3.
The conv tests therefore give the fp64 reference the activation pattern of the kernel under test (`lrelu_like`), which
compares the arithmetic and not the tie-breaking; `tools/lrelu_mask_sensitivity.py` demonstrates the effect."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
SLOPE = 0.2


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def grad_close(a, b, tol):
    """relative L2 error + outlier fraction (see module docstring)"""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    l2 = float((a - b).norm() / (b.norm() + 1e-30))
    outliers = float(((a - b).abs() > 1e-2 * b.abs().max()).double().mean())
    return l2 < tol and outliers <= 1e-4, (l2, outliers)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2)


def resample_ref(x, code):
    if code == 1:
        return F.avg_pool2d(x, 2)
    if code == 2:
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=None)
    return x


def lrelu_like(c, y_hip_nchw):
    """LeakyReLU(0.2) whose slope pattern is taken from the kernel's own output sign (identical wherever |c| is not
    at rounding level; there the two branches differ by < 1e-5 in value)."""
    mask = (y_hip_nchw.detach().cpu() > 0).to(c.dtype)
    return c * (mask + SLOPE * (1 - mask))


def pn_ref(x):
    return x / torch.sqrt(torch.mean(x * x, dim=1, keepdim=True) + 1e-8)


def run_both(f_hip, f_ref, tensors, grad_names, x_name=None, tol1=2e-4, tol2=1e-3):
    """tensors: dict name -> fp32 CPU tensor in NCHW / parameter layout.  f_* take the dict, return one tensor (NCHW-like).
    Checks forward, grads wrt grad_names, and (if x_name) grads of sum((dL/dx)^2) wrt grad_names."""
    ref_in = {k: v.double().clone().requires_grad_(k in grad_names or k == x_name) for k, v in tensors.items()}
    hip_in = {k: v.to(DEV).clone().requires_grad_(k in grad_names or k == x_name) for k, v in tensors.items()}
    out_r = f_ref(ref_in)
    out_h = f_hip(hip_in)
    assert out_h.shape == out_r.shape, (out_h.shape, out_r.shape)
    assert rel(out_h, out_r) < tol1, f"forward rel err {rel(out_h, out_r)}"
    torch.manual_seed(5)
    v = torch.randn(out_r.shape, dtype=torch.float64)
    names = list(grad_names) + ([x_name] if x_name and x_name not in grad_names else [])
    gr = torch.autograd.grad((out_r * v).sum(), [ref_in[k] for k in names], create_graph=x_name is not None)
    gh = torch.autograd.grad((out_h * v.float().to(DEV)).sum(), [hip_in[k] for k in names], create_graph=x_name is not None)
    for k, a, b in zip(names, gh, gr):
        ok, info = grad_close(a, b, tol1)
        assert ok, f"grad {k}: (rel L2, outlier fraction) = {info}"
    if x_name:
        ix = names.index(x_name)
        l2r = (gr[ix] ** 2).sum()
        l2h = (gh[ix] ** 2).sum()
        g2r = torch.autograd.grad(l2r, [ref_in[k] for k in names], allow_unused=True)
        g2h = torch.autograd.grad(l2h, [hip_in[k] for k in names], allow_unused=True)
        for k, a, b in zip(names, g2h, g2r):
            if b is None or float(b.abs().max()) == 0.0:
                continue
            assert a is not None, f"second-order grad {k} missing on the HIP path"
            ok, info = grad_close(a, b, tol2)
            assert ok, f"second-order grad {k}: (rel L2, outlier fraction) = {info}"


CONV_CASES = [
    # B, H, W, Cin, Cout, resample, bias
    (2, 8, 32, 16, 16, 0, False),
    (1, 16, 16, 16, 32, 0, True),
    (2, 12, 20, 32, 16, 0, False),     # ragged tile edges
    (1, 8, 8, 64, 64, 0, False),
    (1, 4, 4, 128, 128, 0, True),
    (2, 8, 8, 16, 16, 1, False),       # avg-pool on load (input 16x16)
    (1, 16, 16, 32, 32, 1, False),
    (2, 8, 8, 32, 16, 2, False),       # bilinear x2 on load (input 4x4)
    (1, 32, 32, 16, 16, 2, False),
    (1, 40, 72, 16, 16, 0, False),     # several tiles in both directions
    # tile-shape selection (csrc/conv3x3.hip kCfg): 4x16-pixel tiles with channels split over waves ...
    (4, 64, 64, 16, 16, 0, False), (4, 64, 64, 16, 32, 0, False), (4, 64, 64, 16, 64, 0, True), (4, 64, 64, 16, 128, 0, False),
    (16, 16, 16, 32, 128, 1, False),   # 1x16 tiles, 128 channels over 4 waves, pooled input
    (16, 16, 16, 64, 64, 2, False),    # bilinear input, 64 channels
    # ... and the 8x32 tile with many channels per wave
    (1, 256, 256, 16, 64, 0, False), (1, 256, 256, 16, 128, 0, True),
    # persistent kernels (few channels, >= 256 tiles of 8x32): every (K, N) in {16, 32}^2, plain / bilinear input, bias
    (2, 128, 256, 16, 16, 0, True), (2, 128, 256, 32, 16, 0, False), (2, 128, 256, 16, 32, 0, False), (2, 128, 256, 32, 32, 0, True),
    (2, 128, 256, 16, 16, 2, False), (2, 128, 256, 32, 16, 2, False), (2, 128, 256, 32, 32, 2, False),
    (1, 200, 328, 16, 16, 0, False),   # ragged right / bottom edges on the persistent path
    (2, 136, 296, 16, 16, 2, True), (2, 136, 296, 32, 16, 2, True),   # bilinear input folded into the weights: ragged edges, bias
    # split-bf16 kernel for many channels on small images (csrc/conv3x3_mid.hip; exact-fp32 mode runs the generic kernel):
    # all output channels in one workgroup (fused PixelNorm) for N = 32 / 64 / 128 ...
    (4, 64, 64, 64, 32, 0, False), (4, 64, 64, 32, 64, 0, True), (4, 64, 64, 32, 128, 0, True),
    # ... output channels split over workgroups (PixelNorm as a second launch), every K, ragged edges, resampled inputs
    (8, 16, 16, 128, 128, 0, True), (32, 16, 16, 128, 128, 0, False), (3, 10, 20, 64, 32, 1, False), (2, 12, 20, 128, 64, 2, True),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_lrelu_pn_all_orders(ngan, case, conv_precision):
    B, H, W, Cin, Cout, res, use_bias = case
    ops = ngan.ops
    torch.manual_seed(hash(case) % 1000)
    hin, win = (2 * H, 2 * W) if res == 1 else ((H // 2, W // 2) if res == 2 else (H, W))
    scale = 1.3868 / np.sqrt(9 * Cin)
    t = {"x": torch.randn(B, Cin, hin, win), "w": torch.randn(Cout, Cin, 3, 3)}
    if use_bias:
        t["b"] = torch.randn(Cout) * 0.5

    def f_hip(d):
        y, _ = ops.ConvLReLUPN.apply(nhwc(d["x"]), d["w"], d.get("b"), res, scale, SLOPE)
        return nchw(y)

    with torch.no_grad():
        pattern = f_hip({k: v.to(DEV) for k, v in t.items()})

    def f_ref(d):
        c = F.conv2d(scale * resample_ref(d["x"], res), d["w"], d.get("b"), padding=1)
        return pn_ref(lrelu_like(c, pattern))

    run_both(f_hip, f_ref, t, [k for k in t if k != "x"], x_name="x")


LINK_CASES = [
    # B, H, W (of the producer's output), C0 -> C1 (producer) -> C2 (consumer), consumer's resample
    (2, 16, 32, 16, 16, 32, 0),        # generic / small kernels
    (2, 128, 256, 16, 16, 16, 0),      # persistent kernel, fused PixelNorm-backward epilogue
    (2, 128, 256, 32, 32, 16, 0),
    (2, 256, 256, 16, 32, 16, 1),      # consumer pools its input: pool-adjoint store + PixelNorm backward at 4 pixels (persistent)
    (2, 64, 128, 16, 16, 32, 2),       # consumer up-samples its input: up2-adjoint + PixelNorm backward kernel
    (4, 64, 64, 32, 64, 64, 0),        # split-bf16 mid kernel, all channels in one workgroup
    (8, 16, 16, 64, 128, 128, 0),      # mid kernel with the channels split over workgroups (PixelNorm backward as a second launch)
    (4, 32, 32, 32, 64, 128, 1),       # mid kernel, pool-adjoint store
    (3, 10, 12, 32, 128, 64, 2),       # ragged, 128 channels through the up2-adjoint kernel
]


@pytest.mark.parametrize("case", LINK_CASES)
def test_pixelnorm_handoff_matches_unfused(ngan, case, conv_precision):
    """first-order mode (ops.first_order_only + PNLink): the consumer's input-gradient kernel applies the producer's
    LeakyReLU->PixelNorm backward.  Same operator sequence, so the result must match the unfused backward to rounding."""
    B, H, W, C0, C1, C2, res = case
    ops = ngan.ops
    torch.manual_seed(sum(case))
    x = torch.randn(B, H, W, C0, device=DEV)
    w1 = (torch.randn(C1, C0, 3, 3, device=DEV)).requires_grad_()
    w2 = (torch.randn(C2, C1, 3, 3, device=DEV)).requires_grad_()
    s1, s2 = 1.3868 / np.sqrt(9 * C0), 1.3868 / np.sqrt(9 * C1)

    def run(linked):
        xx = x.clone().requires_grad_()
        if linked:
            l0, l1 = ops.PNLink(), ops.PNLink()
            y1, _ = ops.ConvLReLUPN.apply(xx, w1, None, 0, s1, SLOPE, None, l0)
            y2, _ = ops.ConvLReLUPN.apply(y1, w2, None, res, s2, SLOPE, l0, l1)
        else:
            y1, _ = ops.ConvLReLUPN.apply(xx, w1, None, 0, s1, SLOPE)
            y2, _ = ops.ConvLReLUPN.apply(y1, w2, None, res, s2, SLOPE)
        torch.manual_seed(1)
        v = torch.randn_like(y2)
        g = torch.autograd.grad((y2 * v).sum(), [xx, w1, w2])
        return [t.detach().clone() for t in g], (l0.fused if linked else None)

    ref, _ = run(False)
    got, fused = run(True)
    assert fused is True
    for name, a, b in zip(["x", "w1", "w2"], got, ref):
        ok, info = grad_close(a, b, 2e-5)
        assert ok, (name, info)


def test_folded_bilinear_border_modes(ngan):
    """precision 3 through the C ABI: by default ngan_conv3x3_fwd writes the border ring itself; with the per-call flag
    NGAN_CONV_SKIP_BORDER (what the Python layer passes) the ring is left to ngan_conv3x3_up2_border.  Both must give the same
    tensor as the operator, and the flag of one call must not leak into the next (no global state in the library)."""
    ops, C = ngan.ops, ngan._C
    ops.set_conv_precision("bf16x3")
    try:
        B, H, W, K, N = 2, 128, 256, 16, 16
        prec = C.conv3x3_algorithm(B, H, W, K, N, 2, 1)
        assert prec == 3
        torch.manual_seed(2)
        x = torch.randn(B, H // 2, W // 2, K, device=DEV)
        w = torch.randn(N, K, 3, 3, device=DEV)
        bias = torch.randn(N, device=DEV)
        y_ref, rn_ref = ops._run_conv(x, w, bias, 2, 0.1, 1, SLOPE)
        packed = ops._packed(w, 0, 0.1, prec)
        y = torch.full_like(y_ref, float("nan"))
        rn = torch.full_like(rn_ref, float("nan"))
        C.call("ngan_conv3x3_fwd", x, packed, bias, y, rn, B, H, W, K, N, 2, 1, 0, SLOPE, 1e-8, prec, C.CONV_SKIP_BORDER)   # ring untouched
        assert torch.isnan(y[:, 0]).all() and torch.isnan(y[:, :, 0]).all() and not torch.isnan(y[:, 1:-1, 1:-1]).any()
        C.call("ngan_conv3x3_up2_border", x, packed, bias, y, rn, B, H, W, K, N, 1, SLOPE, 1e-8)
        assert torch.equal(y, y_ref) and torch.equal(rn, rn_ref)
        y.fill_(float("nan"))
        rn.fill_(float("nan"))
        C.call("ngan_conv3x3_fwd", x, packed, bias, y, rn, B, H, W, K, N, 2, 1, 0, SLOPE, 1e-8, prec, 0)   # default: whole tensor, right after a flagged call
        assert torch.equal(y, y_ref) and torch.equal(rn, rn_ref)
        with pytest.raises(RuntimeError, match="unknown flags"):
            C.call("ngan_conv3x3_fwd", x, packed, bias, y, rn, B, H, W, K, N, 2, 1, 0, SLOPE, 1e-8, prec, 6)
    finally:
        ops.set_conv_precision("f32")


FIRST_BLOCK_CASES = [
    # B, H, W of the image, FromImage channels C, conv outputs N, pooled on load, biases
    (2, 64, 64, 16, 16, False, True),
    (3, 40, 24, 16, 32, True, True),         # ragged rows, pooled image
    (2, 256, 256, 16, 16, True, True),       # the flagship critic's first layer pair
    (1, 12, 20, 32, 32, False, False),
    (5, 6, 6, 64, 16, False, True),          # image narrower than a workgroup's pixel span
]


@pytest.mark.parametrize("case", FIRST_BLOCK_CASES)
def test_first_block_matches_unfused(ngan, case, conv_precision):
    """ops.FirstBlock == ConvLReLUPN(FromImage(x)) for a one-colour image: forward, and the gradients of both layers' parameters
    and of the image.  The fused form associates the sums differently (FromImage folded into the conv's weights first), so
    pre-activations differ by rounding and single LeakyReLU ties may resolve differently: relative-L2 comparison."""
    B, H, W, C, N, pool, use_bias = case
    ops = ngan.ops
    torch.manual_seed(sum(case[:5]))
    x = torch.rand(B, H, W, 1, device=DEV) * 2 - 1
    wf = torch.randn(C, 1, 1, 1, device=DEV).requires_grad_()
    bf = (torch.randn(C, device=DEV) * 0.5).requires_grad_()
    w = torch.randn(N, C, 3, 3, device=DEV).requires_grad_()
    bc = (torch.randn(N, device=DEV) * 0.3).requires_grad_() if use_bias else None
    scale = 1.3868 / np.sqrt(9 * C)
    assert ops.first_block_fusable(x, wf, w)
    params = [wf, bf, w] + ([bc] if use_bias else [])

    def run(fused, linked=False):
        xx = x.clone().requires_grad_()
        link = ops.PNLink() if linked else None
        if fused:
            y, rn = ops.FirstBlock.apply(xx, wf, bf, w, bc, pool, scale, SLOPE, link)
        else:
            f = ops.FromImage.apply(xx, wf, bf, pool)
            y, rn = ops.ConvLReLUPN.apply(f, w, bc, 0, scale, SLOPE, None, link) if linked else ops.ConvLReLUPN.apply(f, w, bc, 0, scale, SLOPE)
        out = y
        if linked:       # a consumer that applies this layer's LeakyReLU->PixelNorm backward in its input-gradient kernel
            w2 = torch.ones(16, N, 3, 3, device=DEV) * 0.01 + torch.eye(16, N, device=DEV)[:, :, None, None]
            out, _ = ops.ConvLReLUPN.apply(y, w2, None, 0, 0.2, SLOPE, link, ops.PNLink())
        torch.manual_seed(1)
        v = torch.randn_like(out)
        g = torch.autograd.grad((out * v).sum(), [xx] + params)
        return y.detach(), rn.detach(), [t.detach().clone() for t in g], (link.fused if linked else None)

    y_ref, rn_ref, g_ref, _ = run(False)
    y, rn, g, _ = run(True)
    assert rel(y, y_ref) < 3e-5 and rel(rn, rn_ref) < 3e-5
    names = ["x", "wf", "bf", "w"] + (["bc"] if use_bias else [])
    for name, a, b in zip(names, g, g_ref):
        ok, info = grad_close(a, b, 2e-4)
        assert ok, (name, info)
    _, _, g_ref, _ = run(False, linked=True)
    _, _, g, was_fused = run(True, linked=True)
    assert was_fused is True
    for name, a, b in zip(names, g, g_ref):
        ok, info = grad_close(a, b, 2e-4)
        assert ok, ("linked", name, info)


@pytest.mark.parametrize("case", [(2, 256, 256, 16, 16, True), (3, 40, 24, 16, 32, True), (2, 64, 64, 16, 16, False)])
def test_first_block_matches_fp64_reference(ngan, case, conv_precision):
    """ops.FirstBlock against plain torch in fp64 on the CPU (avg_pool2d -> conv 1x1 -> scaled conv 3x3 -> leaky_relu -> pixel
    norm): forward and every gradient to 2e-6 relative L2, in both precision modes (the fused operator is exact fp32)."""
    B, H, W, C, N, pool = case
    ops = ngan.ops
    torch.manual_seed(0)
    x = torch.rand(B, H, W, 1, device=DEV) * 2 - 1
    wf = torch.randn(C, 1, 1, 1, device=DEV).requires_grad_()
    bf = (torch.randn(C, device=DEV) * 0.5).requires_grad_()
    w = torch.randn(N, C, 3, 3, device=DEV).requires_grad_()
    bc = (torch.randn(N, device=DEV) * 0.3).requires_grad_()
    scale = 1.3868 / np.sqrt(9 * C)
    xx = x.clone().requires_grad_()
    y, _ = ops.FirstBlock.apply(xx, wf, bf, w, bc, pool, scale, SLOPE, None)
    torch.manual_seed(1)
    v = torch.randn(y.shape)
    got = [y.detach()] + list(torch.autograd.grad((y * v.to(DEV)).sum(), [xx, wf, bf, w, bc]))

    xr = x.double().cpu().permute(0, 3, 1, 2).clone().requires_grad_()
    P = [t.detach().double().cpu().requires_grad_() for t in (wf, bf, w, bc)]
    f = F.conv2d(F.avg_pool2d(xr, 2) if pool else xr, P[0], P[1])
    a = F.leaky_relu(F.conv2d(f * scale, P[2], P[3], padding=1), SLOPE)
    yr = (a / torch.sqrt((a * a).mean(1, keepdim=True) + 1e-8)).permute(0, 2, 3, 1)
    g = torch.autograd.grad((yr * v.double()).sum(), [xr] + P)
    want = [yr.detach(), g[0].permute(0, 2, 3, 1)] + list(g[1:])
    for name, p_, q_ in zip(["y", "x", "wf", "bf", "w", "bc"], got, want):
        err = float((p_.double().cpu() - q_).norm() / q_.norm())
        assert err < 2e-6, (name, err)


def test_first_block_accumulates_into_existing_grad(ngan):
    """with a pre-existing contiguous .grad (the step driver's flat views) the conv weight's gradient is added in place"""
    ops = ngan.ops
    torch.manual_seed(5)
    x = torch.rand(2, 32, 32, 1, device=DEV)
    wf = torch.randn(16, 1, 1, 1, device=DEV).requires_grad_()
    bf = torch.randn(16, device=DEV).requires_grad_()
    w = torch.randn(16, 16, 3, 3, device=DEV).requires_grad_()
    y, _ = ops.FirstBlock.apply(x, wf, bf, w, None, False, 0.1, SLOPE, None)
    (gw,) = torch.autograd.grad(y.sum(), [w], retain_graph=True)
    w.grad = torch.full_like(w, 2.0)
    buf = w.grad.data_ptr()
    y.sum().backward()
    assert w.grad.data_ptr() == buf
    assert rel(w.grad, gw + 2.0) < 1e-6


@pytest.mark.parametrize("case", [(2, 128, 256, 16, 16, False), (2, 128, 256, 32, 16, True), (1, 200, 328, 16, 32, False)])
def test_conv_to_image_fused(ngan, case, conv_precision):
    """ops.ConvLReLUPNToImage == ToImage(ConvLReLUPN(x)): forward with and without the stored activation, first-order gradients"""
    B, H, W, Cin, Cout, use_bias = case
    ops = ngan.ops
    torch.manual_seed(3)
    x = torch.randn(B, H, W, Cin, device=DEV)
    w = torch.randn(Cout, Cin, 3, 3, device=DEV).requires_grad_()
    bias = (torch.randn(Cout, device=DEV) * 0.3).requires_grad_() if use_bias else None
    wi = (torch.randn(1, Cout, 1, 1, device=DEV) * 0.3).requires_grad_()
    scale = 1.3868 / np.sqrt(9 * Cin)
    assert ops.to_image_fusable(x, w, wi, 0)
    with torch.no_grad():
        t_inf = ops.ConvLReLUPNToImage.apply(x, w, bias, wi, 0, scale, SLOPE, None, False)      # inference form: y is never written
        y, _ = ops.ConvLReLUPN.apply(x, w, bias, 0, scale, SLOPE)
        t_ref = ops.ToImage.apply(y, wi)
    assert rel(t_inf, t_ref) < 1e-5
    params = [p for p in (w, bias, wi) if p is not None]

    def grads(fused):
        xx = x.clone().requires_grad_()
        if fused:
            t = ops.ConvLReLUPNToImage.apply(xx, w, bias, wi, 0, scale, SLOPE, None, True)
        else:
            yy, _ = ops.ConvLReLUPN.apply(xx, w, bias, 0, scale, SLOPE)
            t = ops.ToImage.apply(yy, wi)
        torch.manual_seed(2)
        return t.detach(), torch.autograd.grad((t * torch.randn_like(t)).sum(), [xx] + params)

    t1, g1 = grads(True)
    t0, g0 = grads(False)
    assert rel(t1, t0) < 1e-5
    for a, b in zip(g1, g0):
        ok, info = grad_close(a, b, 2e-5)
        assert ok, info


@pytest.mark.parametrize("case", [(2, 8, 8, 16, 32, 0, True), (1, 8, 16, 32, 16, 1, False), (1, 8, 8, 16, 16, 2, False)])
def test_conv_raw_all_orders(ngan, case, conv_precision):
    B, H, W, Cin, Cout, res, use_bias = case
    ops = ngan.ops
    torch.manual_seed(11)
    hin, win = (2 * H, 2 * W) if res == 1 else ((H // 2, W // 2) if res == 2 else (H, W))
    scale = 0.37
    t = {"x": torch.randn(B, Cin, hin, win), "w": torch.randn(Cout, Cin, 3, 3)}
    if use_bias:
        t["b"] = torch.randn(Cout)

    def f_ref(d):
        return torch.tanh(F.conv2d(scale * resample_ref(d["x"], res), d["w"], d.get("b"), padding=1))

    def f_hip(d):
        return torch.tanh(nchw(ops.Conv.apply(nhwc(d["x"]), d["w"], d.get("b"), res, scale)))

    run_both(f_hip, f_ref, t, [k for k in t if k != "x"], x_name="x")


@pytest.mark.parametrize("C", [16, 32, 64, 128])
def test_lrelu_pixelnorm_all_orders(ngan, C):
    ops = ngan.ops
    torch.manual_seed(C)
    t = {"x": torch.randn(3, C, 5, 7), "b": torch.randn(C) * 0.3}

    def f_ref(d):
        return pn_ref(F.leaky_relu(d["x"] + d["b"].view(1, -1, 1, 1), SLOPE))

    def f_hip(d):
        y, _ = ops.LReLUPN.apply(nhwc(d["x"]), d["b"], SLOPE)
        return nchw(y)

    run_both(f_hip, f_ref, t, ["b"], x_name="x")


@pytest.mark.parametrize("pool", [False, True])
def test_from_image_all_orders(ngan, pool):
    ops = ngan.ops
    torch.manual_seed(2)
    t = {"x": torch.randn(3, 1, 12, 20), "w": torch.randn(16, 1, 1, 1), "b": torch.randn(16)}

    def f_ref(d):
        x = F.avg_pool2d(d["x"], 2) if pool else d["x"]
        return torch.tanh(F.conv2d(x, d["w"], d["b"]))

    def f_hip(d):
        return torch.tanh(nchw(ops.FromImage.apply(nhwc(d["x"]), d["w"], d["b"], pool)))

    run_both(f_hip, f_ref, t, ["w", "b"], x_name="x")


def test_to_image_first_order(ngan):
    ops = ngan.ops
    torch.manual_seed(3)
    t = {"x": torch.randn(2, 16, 9, 11), "w": torch.randn(1, 16, 1, 1) * 0.3}
    run_both(lambda d: nchw(ops.ToImage.apply(nhwc(d["x"]), d["w"])), lambda d: torch.tanh(F.conv2d(d["x"], d["w"])),
             t, ["x", "w"])


@pytest.mark.parametrize("C", [1, 16, 128])
def test_resample_pairs(ngan, C):
    ops = ngan.ops
    torch.manual_seed(4)
    t = {"x": torch.randn(2, C, 6, 10)}
    run_both(lambda d: nchw(ops.Up2.apply(nhwc(d["x"]))) ** 2, lambda d: resample_ref(d["x"], 2) ** 2, t, [], x_name="x")
    run_both(lambda d: nchw(ops.Pool2.apply(nhwc(d["x"]))) ** 2, lambda d: resample_ref(d["x"], 1) ** 2, t, [], x_name="x")


def test_lerp_and_xhat(ngan):
    ops = ngan.ops
    torch.manual_seed(6)
    a, b = torch.randn(2, 8, 8, 16), torch.randn(2, 8, 8, 16)
    alpha = torch.tensor([0.3])
    t = {"a": a, "b": b}
    run_both(lambda d: ops.Lerp.apply(d["a"], d["b"], alpha.to(DEV)) ** 2, lambda d: (d["a"] + 0.3 * (d["b"] - d["a"])) ** 2,
             t, ["b"], x_name="a")
    real, fake, eps = torch.randn(3, 1, 8, 8), torch.randn(3, 1, 8, 8), torch.rand(3, 1, 1, 1)
    got = ops.xhat(real.to(DEV), fake.to(DEV), eps.to(DEV)).cpu()
    assert rel(got, eps * real + (1 - eps) * fake) < 1e-6


@pytest.mark.parametrize("B,C,S", [(3, 32, 4), (2, 20, 3), (5, 128, 16)])   # power-of-two sizes take shifts, 20 x 9 the division path; 128 x 256 = the default head
def test_final_dot_all_orders(ngan, B, C, S):
    ops = ngan.ops
    torch.manual_seed(7)
    t = {"x": torch.randn(B, C, S, S), "w": torch.randn(1, C, S, S), "b": torch.randn(1)}
    scale = 2.5 / (C * S * S) ** 0.5          # scores of order one: the tanh in front of the loss must not saturate
    run_both(lambda d: torch.tanh(ops.FinalDot.apply(nhwc(d["x"]), d["w"], d["b"], scale)),
             lambda d: torch.tanh(F.conv2d(scale * d["x"], d["w"], d["b"]).flatten(1)), t, ["w", "b"], x_name="x")


@pytest.mark.parametrize("B,K,S,C", [(3, 32, 4, 32), (20, 512, 16, 128), (37, 768, 4, 32)])   # MFMA wgrad (K <= 512) / row-streaming wgrad
def test_linear_stem(ngan, B, K, S, C):
    ops = ngan.ops
    torch.manual_seed(8)
    t = {"z": torch.randn(B, K), "w": torch.randn(C * S * S, K) * 0.1}
    scale = 0.0613

    def f_ref(d):
        h = F.linear(scale * d["z"], d["w"]).view(B, C, S, S)
        return pn_ref(F.leaky_relu(h, SLOPE))

    def f_hip(d):
        y, _ = ops.LinearLReLUPN.apply(d["z"], d["w"], S, scale, SLOPE)
        return nchw(y)

    run_both(f_hip, f_ref, t, ["z", "w"])


@pytest.mark.parametrize("B,K,S2,C,gscale", [(16, 512, 256, 128, 1.0), (24, 64, 16, 32, 0.5), (7, 48, 9, 20, 1.0)])
def test_stem_adam_epilogue_against_torch_adam(ngan, B, K, S2, C, gscale):
    """ngan_linear_wgrad_adam: Adam applied to the stem weight in the epilogue of its gradient's factor product (no stored gradient),
    three consecutive steps, against torch.optim.Adam fed the fp64 gradient  scale * gscale * sum_b gc[b] (x) z[b]  (row c*S2 + p of
    the weight pairs with gc[b][p][c]: models.py:299-311's Unflatten)."""
    C_ = ngan._C
    torch.manual_seed(12)
    rows = C * S2
    w = (torch.randn(rows, K) * 0.1)
    p = w.clone().to(DEV)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    lr, b1, b2, eps, scale = 1e-3, 0.5, 0.999, 1e-8, 0.0613
    hyper = torch.tensor([lr, b1, b2, eps, gscale, 1.0 - b1, 1.0 - b2, np.log(b1), np.log(b2)], dtype=torch.float32, device=DEV)
    step = torch.zeros(1, dtype=torch.float32, device=DEV)
    ref = torch.nn.Parameter(w.double())
    opt = torch.optim.Adam([ref], lr=lr, betas=(b1, b2), eps=eps)
    for it in range(3):
        z = torch.randn(B, K)
        gc = torch.randn(B, S2, C)
        ref.grad = scale * gscale * torch.einsum("bpc,bk->cpk", gc.double(), z.double()).reshape(rows, K)
        opt.step()
        step += 1                                        # ngan_adam_step's advance launch does this for every active tensor
        C_.call("ngan_linear_wgrad_adam", z.to(DEV), gc.to(DEV), p, m, v, step, hyper, hyper.numel(), B, K, S2, C, scale)
        # a step moves every element by ~lr * g / |g|.  Where the fp32 and the fp64 gradient agree to 1e-7 of the tensor's scale the
        # two updates agree to ~1e-9 (measured); an element whose gradient is itself ~1e-7 of that scale may move by up to lr the
        # other way (measured: a few dozen of 16.8 M, worst 4e-5 = 0.04 lr): bound their share and the mean
        diff = (p.cpu().double() - ref.detach()).abs()
        assert float((diff > 0.02 * lr).double().mean()) < 2e-4 and float(diff.mean()) < 1e-4 * lr * (it + 1), (it, float(diff.max()))
        assert float(diff.max()) < 1.01 * lr * (it + 1), it
    assert rel(m.cpu(), opt.state[ref]["exp_avg"]) < 1e-5 and rel(v.cpu(), opt.state[ref]["exp_avg_sq"]) < 1e-5


@pytest.mark.parametrize("shape", [(4, 1, 32, 32), (3, 1, 6, 6), (16, 1, 512, 512)])
def test_sample_l2norm(ngan, shape):
    ops = ngan.ops
    torch.manual_seed(9)
    t = {"g": torch.randn(*shape)}
    run_both(lambda d: ops.SampleL2Norm.apply(d["g"]), lambda d: d["g"].norm(2, dim=(1, 2, 3)), t, ["g"])


def test_unsupported_shapes_fail_loudly(ngan):
    ops = ngan.ops
    x = torch.randn(1, 4, 4, 12, device=DEV)
    w = torch.randn(16, 12, 3, 3, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 16|must be|unsupported|N=|K="):
        ops.ConvLReLUPN.apply(x, w, None, 0, 1.0, SLOPE)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.LReLUPN.apply(torch.randn(1, 2, 2, 16), None, SLOPE)


MANY_TILE_CASES = [
    # B, H, W, K, N, epilogue (0 plain, 1 LeakyReLU -> PixelNorm), pool-adjoint store: several tiles per persistent workgroup
    (32, 64, 64, 32, 32, 1, 0), (8, 256, 256, 16, 32, 1, 0), (32, 128, 128, 16, 16, 1, 0), (16, 64, 64, 32, 16, 0, 1),
    (4, 512, 512, 16, 16, 0, 0),
    # 16 -> 16 in exact fp32 runs the Winograd F(2x2, 3x3) form of the tile kernel (precision code 4): pool-adjoint store, and an image
    # height that is not a multiple of the 8-row tile (rows past the image belong to no one: dropped by the per-image descriptor)
    (8, 128, 128, 16, 16, 0, 1), (6, 100, 128, 16, 16, 1, 0),
]


@pytest.mark.parametrize("case", MANY_TILE_CASES)
def test_persistent_conv_with_many_tiles_per_workgroup(ngan, case, conv_precision):
    """Large batches through the C ABI: a persistent workgroup walks several tiles, so its stores of tile t overlap the
    loads and MFMAs of tile t+1.  (A version of the tile kernel whose stores carried the tile offset in the SGPR soffset field
    corrupted single components of < 1 % of the tiles -- only batches of this size showed it, none of the small op cases did.)
    Reference: torch fp32 conv2d on the CPU of the same operands; every element within 2e-4 of the tensor's max-norm."""
    B, H, W, K, N, epi, out_mode = case
    C, ops = ngan._C, ngan.ops
    torch.manual_seed(sum(case))
    x = torch.randn(B, H, W, K)
    w = torch.randn(N, K, 3, 3)
    bias = torch.randn(N) if out_mode == 0 else None
    scale = 1.3868 / np.sqrt(9 * K)
    prec = C.conv3x3_algorithm(B, H, W, K, N, 0, ops.PRECISIONS[conv_precision])
    xd, wd = x.to(DEV), w.to(DEV)
    oh, ow = (2 * H, 2 * W) if out_mode else (H, W)
    y = torch.full((B, oh, ow, N), float("nan"), device=DEV)
    rn = torch.full((B, H, W), float("nan"), device=DEV)
    C.call("ngan_conv3x3_fwd_ex", xd, ops._packed(wd, 0, scale, prec), bias.to(DEV) if bias is not None else None, y, rn if epi else None,
           None, None, None, B, H, W, K, N, 0, epi, out_mode, SLOPE, 1e-8, prec, 0)
    c = F.conv2d(nchw(x) * scale, w, bias, padding=1)
    if epi:
        c = F.leaky_relu(c, SLOPE)
        r = torch.sqrt(torch.mean(c * c, dim=1, keepdim=True) + 1e-8)
        c = c / r
        assert rel(rn.cpu(), r[:, 0]) < 2e-4
    if out_mode:
        c = F.interpolate(c, scale_factor=2, mode="nearest") * 0.25        # adjoint of the 2x2 average
    got = nchw(y.cpu())
    assert not torch.isnan(got).any()
    worst = float((got - c).abs().max() / c.abs().max())
    assert worst < 2e-4, worst


WINO_SHAPES = [  # B, H, W, epilogue, out_mode, mode (0 forward weights, 1 input-gradient weights), resample (0 plain, 2 bilinear x2 on load)
    (8, 128, 128, 0, 0, 0, 0), (8, 128, 128, 1, 0, 0, 0), (6, 100, 128, 1, 0, 1, 0), (1, 200, 328, 0, 0, 0, 0), (1, 200, 328, 1, 0, 1, 0),
    (8, 128, 128, 2, 0, 1, 0), (8, 128, 128, 0, 1, 1, 0), (8, 128, 128, 2, 1, 1, 0), (2, 256, 256, 3, 0, 0, 0), (4, 256, 256, 1, 0, 0, 0),
    (2, 128, 256, 0, 0, 0, 2), (2, 136, 296, 1, 0, 0, 2), (2, 128, 256, 1, 0, 0, 2), (3, 136, 256, 1, 0, 0, 2), (16, 64, 64, 0, 0, 0, 2),
]


@pytest.mark.parametrize("kn", [(16, 16), (16, 32), (32, 16), (32, 32)])
@pytest.mark.parametrize("shape", WINO_SHAPES)
def test_winograd_kernels_against_fp64(ngan, shape, kn):
    """The Winograd F(2x2, 3x3) kernels (precision code 4: conv3x3_tile / _persist for 16 -> 16, conv3x3_wino for the shapes with
    a 32-channel side) through the C ABI against an fp64 evaluation of the WHOLE fused operator -- every epilogue and store mode,
    whole and ragged tiles, forward and input-gradient weight orientation.  The reference applies LeakyReLU by itself: a bias of
    +-12 per output channel keeps every pre-activation away from the kink (epilogues 1 and 3).  Promoted from tools/wino_check.py."""
    B, H, W, epi, om, mode, res = shape
    K, N = kn
    C, ops = ngan._C, ngan.ops
    prec = C.conv3x3_algorithm(B, H, W, K, N, res, 0)
    if prec != 4:
        assert (K, N) != (16, 16) and W % 32, "a Winograd-eligible shape fell back to the direct form"
        pytest.skip("the 32-channel Winograd kernels need whole 32-pixel tiles along x")
    torch.manual_seed(B + H + W + epi + om + K + 3 * N)
    hin, win = (H // 2, W // 2) if res == 2 else (H, W)
    x = torch.randn(B, hin, win, K)
    w = torch.randn(N, K, 3, 3) if mode == 0 else torch.randn(K, N, 3, 3)          # mode 1: the weight of the conv being differentiated
    scale = 1.3868 / np.sqrt(9 * K)
    oh, ow = (2 * H, 2 * W) if om else (H, W)
    sign = torch.tensor([1.0 if (c // 2) % 2 == 0 else -1.0 for c in range(N)])
    bias = (12.0 * sign + 0.5 * torch.randn(N)) if epi in (1, 3) else (torch.randn(N) if epi == 0 and not om else None)
    ay = torch.randn(B, oh, ow, N) if epi == 2 else (torch.randn(N) if epi == 3 else None)
    arn = (torch.rand(B, oh, ow) + 0.5) if epi == 2 else None
    y = torch.full((B, oh, ow, N), float("nan"), device=DEV)
    rn = torch.full((B, H, W), float("nan"), device=DEV)
    aout = torch.full((B, H, W), float("nan"), device=DEV) if epi == 3 else None
    pooled = epi == 1 and om == 0 and C.conv3x3_pooled_output(B, H, W, K, N, res, prec)      # epilogue 1's pooled side output (include/ngan.h)
    if pooled:
        aout = torch.full((B, H // 2, W // 2, N), float("nan"), device=DEV)
    dv = lambda t: None if t is None else t.to(DEV)
    C.call("ngan_conv3x3_fwd_ex", dv(x), ops._packed(dv(w), mode, scale, prec), dv(bias), y, rn if epi in (1, 3) else None, dv(ay), dv(arn), aout,
           B, H, W, K, N, res, epi, om, SLOPE, 1e-8, prec, 0)
    # fp64 reference of the same operator
    wd = w.double() if mode == 0 else w.double().flip(2, 3).transpose(0, 1)
    xin = x.double().permute(0, 3, 1, 2)
    if res == 2:
        xin = F.interpolate(xin, scale_factor=2, mode="bilinear", align_corners=False)
    c = F.conv2d(xin * scale, wd, None if bias is None else bias.double(), padding=1)
    if epi in (1, 3):
        assert float(c.abs().min()) > 1.0
        c = F.leaky_relu(c, SLOPE)
        r = torch.sqrt(torch.mean(c * c, dim=1, keepdim=True) + 1e-8)
        c = c / r
        assert rel(rn.cpu(), r[:, 0]) < 2e-6
    if om:
        c = F.interpolate(c, scale_factor=2, mode="nearest") * 0.25                 # adjoint of the 2x2 average
    if epi == 2:
        a64, r64 = ay.double().permute(0, 3, 1, 2), arn.double().unsqueeze(1)
        m = torch.where(a64 > 0, torch.ones_like(a64), torch.full_like(a64, SLOPE))
        c = m * (c - a64 * torch.mean(c * a64, dim=1, keepdim=True)) / r64
    if pooled:
        assert torch.equal(aout, ops._pooled(y)), "the pooled side output is not bit-identical to ngan_pool2_fwd of the output"
    got = nchw(y.cpu()).double()
    assert not torch.isnan(got).any()
    err_l2, err_max = float((got - c).norm() / c.norm()), float((got - c).abs().max() / c.abs().max())
    assert err_l2 < 1e-6 and err_max < 2e-5, (err_l2, err_max)
    if epi == 3:
        t = torch.tanh(torch.sum(c * ay.double().view(1, N, 1, 1), dim=1))
        assert float((aout.cpu().double() - t).abs().max()) < 2e-5


FULL_SIZE_LAYERS = [  # B, H, W, Cin, Cout, resample: layers of the 512x512 final stage at BASELINE.json's batch 16 (and 32: the critic step)
    (16, 512, 512, 16, 16, 0), (32, 512, 512, 16, 16, 2), (16, 256, 256, 32, 16, 2), (32, 256, 256, 16, 16, 1), (32, 128, 128, 16, 32, 1),
    (32, 128, 128, 32, 32, 0), (16, 128, 128, 32, 32, 2), (32, 64, 64, 32, 32, 1), (32, 32, 32, 64, 64, 0), (32, 16, 16, 128, 128, 0),
]


@pytest.mark.parametrize("layer", FULL_SIZE_LAYERS)
def test_full_size_layers_satisfy_the_adjoint_identities(ngan, layer, conv_precision):
    """At BASELINE.json's full sizes an fp64 reference of a layer takes minutes on the CPU; the convolution triple is checked there
    through identities that hold for ANY correct implementation and need no reference:  <conv(x; W), g> = <x, dgrad(g; W)> =
    <W, wgrad(x, g)>  (the three kernels compute the three partial derivatives of one trilinear form).  The inner products are
    accumulated in fp64 on the GPU; the three numbers must agree to 2e-5 of their magnitude (random fp32 operands, ~1e7 terms).  A wrong
    tile offset, a 32-bit overflow or a dropped border anywhere in an image shows up here; the small cases elsewhere pin the values."""
    B, H, W, Cin, Cout, res = layer
    ops = ngan.ops
    torch.manual_seed(B + H + Cin + 7 * Cout + res)
    hin, win = (2 * H, 2 * W) if res == 1 else ((H // 2, W // 2) if res == 2 else (H, W))
    x = torch.randn(B, hin, win, Cin, device=DEV)
    w = torch.randn(Cout, Cin, 3, 3, device=DEV)
    g = torch.randn(B, H, W, Cout, device=DEV)
    scale = 1.0 / np.sqrt(9 * Cin)
    y = ops.Conv.apply(x, w, None, res, scale)
    gx = ops.ConvDgrad.apply(g, w, res, scale)
    gw = ops.ConvWgrad.apply(x, g, res, scale)
    assert y.shape == g.shape and gx.shape == x.shape and gw.shape == w.shape
    dot = lambda a, b: float((a.double() * b.double()).sum())
    a0, a1, a2 = dot(y, g), dot(x, gx), dot(w, gw)
    mag = float(y.double().norm() * g.double().norm())
    assert abs(a0 - a1) < 2e-5 * mag and abs(a0 - a2) < 2e-5 * mag, (a0, a1, a2, mag)


@pytest.mark.parametrize("shape", [(16, 512, 512, 16), (32, 128, 128, 32), (32, 16, 16, 128)])
def test_full_size_pixelnorm_properties(ngan, shape):
    """LeakyReLU -> PixelNorm at BASELINE.json's sizes through properties that need no reference (models.py:118-126, 263):
    forward  y * r = LeakyReLU(c) and mean_c(y^2) = 1 - eps / r^2;  backward  sum_c (gc / m) * y = 0 per pixel (the gradient w.r.t. the
    pre-activation, with the LeakyReLU slope divided out, is orthogonal to y: PixelNorm's output does not change along y);
    backward-of-backward: the same orthogonality for the term that is linear in the incoming second-order gradient."""
    B, H, W, C = shape
    ops = ngan.ops
    torch.manual_seed(C + H)
    c = torch.randn(B, H, W, C, device=DEV)
    y, r = ops.LReLUPN.apply(c, None, SLOPE)
    lre = torch.where(c > 0, c, SLOPE * c)
    assert float((y * r.unsqueeze(-1) - lre).abs().max()) < 1e-5
    assert float(((y.double() ** 2).mean(-1) - 1.0).abs().max()) < 1e-5
    gy = torch.randn_like(y)
    gc = ops.LReLUPNBwd.apply(gy, None, y, r, SLOPE)
    m = torch.where(y > 0, torch.ones_like(y), torch.full_like(y, SLOPE))
    ortho = ((gc.double() / m.double()) * y.double()).sum(-1)
    scale = float(gy.double().norm(dim=-1).mean() / r.double().mean())
    assert float(ortho.abs().max()) < 1e-4 * scale * np.sqrt(C), (float(ortho.abs().max()), scale)
    h = torch.randn_like(y)
    ggy = torch.empty_like(y); gy_out = torch.empty_like(y); gr_out = torch.empty_like(r)
    ngan._C.call("ngan_lrelu_pixelnorm_bwdbwd", h, gy, y, r, ggy, gy_out, gr_out, y.numel() // C, C, SLOPE)
    ortho2 = (ggy.double() * y.double()).sum(-1)          # ggy = (m h - y mean_c(m h y)) / r is orthogonal to y as well
    assert float(ortho2.abs().max()) < 1e-4 * scale * np.sqrt(C)


@pytest.mark.parametrize("case", [(2, 16, 32, 16, 16, 0), (1, 64, 64, 32, 32, 0), (2, 8, 8, 64, 64, 0), (1, 32, 64, 16, 32, 1), (2, 32, 32, 32, 16, 2),
                                  (4, 16, 16, 128, 128, 0),                                        # 128 channels (mid kernel, channels split over workgroups)
                                  (2, 128, 256, 16, 16, 0), (2, 128, 256, 32, 32, 0), (2, 128, 256, 16, 32, 0), (2, 128, 256, 32, 16, 0),   # Winograd-eligible
                                  (2, 128, 256, 16, 16, 2)])
def test_conv_lrelu_pn_with_the_reference_own_leaky_mask(ngan, case, conv_precision):
    """The other conv tests hand the fp64 reference the kernel's own LeakyReLU sign pattern (`lrelu_like`), so a wrong sign rule in
    the kernel could hide.  Here the pre-activations are kept away from zero -- a bias of +-8 per output channel on top of a conv
    result of |c| < ~4 -- so the sign pattern is unambiguous, half the channels take the 0.2 slope, and the reference applies
    F.leaky_relu by itself.  All orders: forward, first-order gradients, d/dW of |dL/dx|^2."""
    B, H, W, Cin, Cout, res = case
    ops = ngan.ops
    torch.manual_seed(sum(case) + 1)
    hin, win = (2 * H, 2 * W) if res == 1 else ((H // 2, W // 2) if res == 2 else (H, W))
    sign = torch.tensor([1.0 if (c // 2) % 2 == 0 else -1.0 for c in range(Cout)])
    amp = 12.0 if B * H * W > 10000 else 8.0                # (more pre-activations: a wider margin keeps all of them off the kink)
    t = {"x": torch.randn(B, Cin, hin, win), "w": torch.randn(Cout, Cin, 3, 3), "b": amp * sign + 0.5 * torch.randn(Cout)}
    scale = 1.3868 / np.sqrt(9 * Cin)

    def f_hip(d):
        y, _ = ops.ConvLReLUPN.apply(nhwc(d["x"]), d["w"], d["b"], res, scale, SLOPE)
        return nchw(y)

    def f_ref(d):
        c = F.conv2d(resample_ref(d["x"], res) * scale, d["w"], d["b"], padding=1)
        assert float(c.abs().min()) > 1.0                   # every pre-activation is far from the kink
        frac_neg = float((c < 0).double().mean())
        assert 0.3 < frac_neg < 0.7                         # and both slopes are exercised
        return pn_ref(F.leaky_relu(c, SLOPE))

    run_both(f_hip, f_ref, t, ["w", "b"], x_name="x")


def test_loss_heads_and_latent_projection_match_torch(ngan):
    """The fused scalar heads (ops.WLossHead, ops.GradPenaltyHead) and the latent projection against the torch expressions of
    the reference (loss_functions.py:21-45, 67, 176; utils.py:77-78): values and gradients in fp64 on the CPU."""
    ops = ngan.ops
    torch.manual_seed(4)
    b = 24
    scores = torch.randn(2 * b, 1)
    # critic head with drift, and the generator head (n_fake = 0)
    for n_real, drift in ((b, 0.001), (b, 0.0), (2 * b, 0.0)):
        sr = scores.double().clone().requires_grad_()
        real, fake = sr[:n_real], sr[n_real:]
        want = -real.mean() + (fake.mean() if n_real < 2 * b else 0.0) + drift * torch.square(real).mean()
        sh = scores.to(DEV).clone().requires_grad_()
        loss, m_real, m_fake = ops.WLossHead.apply(sh, n_real, drift)
        assert abs(float(loss) - float(want)) < 1e-6 and abs(float(m_real) - float(real.mean())) < 1e-6
        if n_real < 2 * b:
            assert abs(float(m_fake) - float(fake.mean())) < 1e-6
        loss += 0.0                                       # the reference's loop modifies the loss in place: must be allowed
        (3.0 * loss + 0.5 * m_real).backward()
        (3.0 * want + 0.5 * real.mean()).backward()
        assert rel(sh.grad, sr.grad) < 1e-6
    # gradient-penalty head
    g = torch.randn(6, 12, 12, 1)
    gr = g.double().clone().requires_grad_()
    want = 10.0 * torch.mean((gr.norm(2, dim=(1, 2, 3)) - 1) ** 2)
    gh = g.to(DEV).clone().requires_grad_()
    pen, norms = ops.GradPenaltyHead.apply(gh, 10.0)
    assert rel(norms, gr.norm(2, dim=(1, 2, 3))) < 1e-6 and abs(float(pen) - float(want)) < 1e-5 * float(want)
    (2.0 * pen).backward()
    (2.0 * want).backward()
    assert rel(gh.grad, gr.grad) < 1e-5
    # latent projection
    z = torch.randn(7, 512) * 3.0
    got = ops.latent_normalize_(z.to(DEV).clone(), 5.0).cpu()
    zc = z.clamp(-5, 5)
    assert rel(got, zc / zc.norm(p=2, dim=1, keepdim=True)) < 1e-6


def test_inputs_only_skips_parameter_gradients(ngan, monkeypatch):
    """`ops.inputs_only()` (used around the gradient penalty's autograd.grad w.r.t. x_hat, reference loss_functions.py:170-176): the
    input gradient and everything differentiated from it later are bit-identical to the plain pass, and no weight-gradient or
    bias-sum kernel is launched for results the engine would drop."""
    ops = ngan.ops
    torch.manual_seed(8)
    x = nhwc(torch.randn(2, 16, 32, 32)).to(DEV)
    w1 = (torch.randn(32, 16, 3, 3)).to(DEV).requires_grad_()
    b1 = torch.randn(32).to(DEV).requires_grad_()
    w2 = (torch.randn(16, 32, 3, 3)).to(DEV).requires_grad_()
    calls = {"wgrad": 0, "sum": 0}
    real_wgrad, real_sum = ops._run_wgrad, ops._channel_sum

    def count_wgrad(*a, **k):
        calls["wgrad"] += 1
        return real_wgrad(*a, **k)

    def count_sum(*a, **k):
        calls["sum"] += 1
        return real_sum(*a, **k)

    monkeypatch.setattr(ops, "_run_wgrad", count_wgrad)
    monkeypatch.setattr(ops, "_channel_sum", count_sum)

    def penalty(inputs_only):
        xx = x.clone().requires_grad_()
        y, _ = ops.ConvLReLUPN.apply(xx, w1, b1, 0, 0.1, SLOPE)
        out = ops.Conv.apply(y, w2, None, 0, 0.1)
        ones = torch.ones_like(out)
        calls["wgrad"] = calls["sum"] = 0
        if inputs_only:
            with ops.inputs_only():
                gx = torch.autograd.grad(out, xx, ones, create_graph=True)[0]
        else:
            gx = torch.autograd.grad(out, xx, ones, create_graph=True)[0]
        first = dict(calls)
        gw = torch.autograd.grad(gx.square().sum(), [w1, b1, w2])
        return gx.detach(), gw, first

    gx0, gw0, first0 = penalty(False)
    gx1, gw1, first1 = penalty(True)
    assert first0["wgrad"] == 2 and first0["sum"] >= 1           # what a custom Function does when it cannot see the output mask
    assert first1 == {"wgrad": 0, "sum": 0}
    assert torch.equal(gx0, gx1)
    for a, b in zip(gw0, gw1):
        assert torch.equal(a, b)
