"""Compare ngan_conv3x3_wgrad (fp32 and bf16x3) with an fp64 CPU reference for one shape."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); C = pkg._C
B, H, W, K, N = [int(v) for v in sys.argv[1:6]]
torch.manual_seed(0)
x = torch.randn(B, H, W, K); g = torch.randn(B, H, W, N)
xr = x.permute(0, 3, 1, 2).double(); gr = g.permute(0, 3, 1, 2).double()
w = torch.zeros(N, K, 3, 3, dtype=torch.float64, requires_grad=True)
(F.conv2d(xr, w, padding=1) * gr).sum().backward()
ref = w.grad
xd, gd = x.cuda(), g.cuda()
for prec in (0, 1):
    gw = torch.empty(N, K, 3, 3, device="cuda")
    ws = torch.empty(C.wgrad_workspace_bytes(B, H, W, K, N) // 4, device="cuda")
    C.call("ngan_conv3x3_wgrad", xd, gd, gw, ws, B, H, W, K, N, 0, 1.0, 0, prec)
    err = (gw.cpu().double() - ref).abs()
    print(f"prec{prec} B{B} {H}x{W} K{K} N{N}: max rel err {float(err.max() / ref.abs().max()):.3e}  rms rel {float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}  ref rms {float(ref.pow(2).mean().sqrt()):.3f}")
    if prec == 1:
        e = err / ref.abs().max()
        print("   per-tap max err:", [f"{float(e[:, :, i // 3, i % 3].max()):.1e}" for i in range(9)])
