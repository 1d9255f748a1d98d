"""Winograd F(2x2, 3x3) form of conv3x3_tile_kernel (exact fp32 arithmetic, precision code 4) against the direct form (NGAN_WINOGRAD=0)
and, for the plain conv, against an fp64 torch convolution on the CPU: every epilogue / store mode of the 16 -> 16 instances,
whole and ragged tiles in y.  Prints max |difference| and relative L2 errors.        python tools/wino_check.py   (on the GPU box)"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # B, H, W, epilogue, out_mode, mode (0 forward weights, 1 input-gradient weights), resample (0 plain, 2 bilinear x2 on load)
    (8, 128, 128, 0, 0, 0, 0), (8, 128, 128, 1, 0, 0, 0), (6, 100, 128, 1, 0, 1, 0), (1, 200, 328, 0, 0, 0, 0), (1, 200, 328, 1, 0, 1, 0),
    (8, 128, 128, 2, 0, 1, 0), (8, 128, 128, 0, 1, 1, 0), (8, 128, 128, 2, 1, 1, 0), (2, 256, 256, 3, 0, 0, 0), (16, 256, 256, 1, 0, 0, 0),
    (16, 256, 256, 0, 0, 1, 0), (2, 128, 256, 0, 0, 0, 2), (2, 136, 296, 1, 0, 0, 2),
]


def run(path):
    from __graft_entry__ import load_package
    pkg = load_package()
    C, ops = pkg._C, pkg.ops
    outs = []
    for (B, H, W, epi, om, mode, res) in SHAPES:
        torch.manual_seed(B + H + W + epi + om)
        K = N = 16
        prec = C.conv3x3_algorithm(B, H, W, K, N, res, 0)
        x = torch.randn(B, H // 2, W // 2, K, device="cuda") if res == 2 else torch.randn(B, H, W, K, device="cuda")
        w = torch.randn(N, K, 3, 3, device="cuda")
        oh, ow = (2 * H, 2 * W) if om else (H, W)
        ay = torch.randn(B, oh, ow, N, device="cuda") if epi == 2 else (torch.randn(N, device="cuda") if epi == 3 else None)
        arn = (torch.rand(B, oh, ow, device="cuda") + 0.5) if epi == 2 else None
        aout = torch.zeros(B, H, W, device="cuda") if epi == 3 else None
        bias = torch.randn(N, device="cuda") if epi in (0, 1, 3) and not om else None
        packed = ops._packed(w, mode, 0.1, prec)
        y = torch.zeros(B, oh, ow, N, device="cuda")
        rn = torch.zeros(B, H, W, device="cuda")
        C.call("ngan_conv3x3_fwd_ex", x, packed, bias, y, rn if epi in (1, 3) else None, ay, arn, aout, B, H, W, K, N, res, epi, om, 0.2, 1e-8, prec, 0)
        torch.cuda.synchronize()
        ref = None
        if epi == 0 and om == 0:
            wd = w.double().cpu()
            if mode == 1:
                wd = wd.flip(2, 3).transpose(0, 1)
            xin = x.double().cpu().permute(0, 3, 1, 2)
            if res == 2:
                xin = torch.nn.functional.interpolate(xin, scale_factor=2, mode="bilinear", align_corners=False)
            ref = torch.nn.functional.conv2d(xin * 0.1, wd, bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
        outs.append((prec, y.cpu(), rn.cpu(), aout.cpu() if aout is not None else None, ref))
    torch.save(outs, path)


if len(sys.argv) > 1:
    run(sys.argv[1])
    sys.exit(0)
subprocess.run([sys.executable, __file__, "/tmp/wa.pt"], env=dict(os.environ, NGAN_WINOGRAD="0"), check=True)
subprocess.run([sys.executable, __file__, "/tmp/wb.pt"], env=dict(os.environ, NGAN_WINOGRAD="1"), check=True)
A, Bq = torch.load("/tmp/wa.pt"), torch.load("/tmp/wb.pt")
for s, a, b in zip(SHAPES, A, Bq):
    d = [float((u - v).abs().max()) if u is not None else 0.0 for u, v in zip(a[1:4], b[1:4])]
    scale = float(a[1].abs().max())
    msg = f"codes {a[0]}/{b[0]}  max|y| {scale:.3f}  max diff y / norm / image: {d[0]:.2e} {d[1]:.2e} {d[2]:.2e}"
    if a[4] is not None:
        e0 = float((a[1].double() - a[4]).norm() / a[4].norm()), float((a[1].double() - a[4]).abs().max())
        e1 = float((b[1].double() - b[4]).norm() / b[4].norm()), float((b[1].double() - b[4]).abs().max())
        msg += f"  | vs fp64: direct rel-L2 {e0[0]:.2e} max {e0[1]:.2e}; winograd rel-L2 {e1[0]:.2e} max {e1[1]:.2e}"
    print(s, msg, flush=True)
