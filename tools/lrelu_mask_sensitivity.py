import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); ops = pkg.ops; C = pkg._C
B, H, W, Cin, Cout = 2, 128, 256, 16, 16
torch.manual_seed(1)
x = torch.randn(B, H, W, Cin).cuda(); w = torch.randn(Cout, Cin, 3, 3).cuda(); scale = 0.1155
c = F.conv2d(x.permute(0, 3, 1, 2).double() * scale, w.double(), padding=1)
a = F.leaky_relu(c, 0.2); r_ref = torch.sqrt((a * a).mean(1) + 1e-8); y_ref = (a / r_ref.unsqueeze(1)).permute(0, 2, 3, 1)
gy = torch.randn(B, H, W, Cout).cuda()
# reference PN-bwd
yy = y_ref; s = (gy.double() * yy).mean(-1, keepdim=True); m = torch.where(yy > 0, 1.0, 0.2)
gc_ref = m * (gy.double() - yy * s) / r_ref.unsqueeze(-1)
for prec in (0, 1):
    p = C.conv3x3_algorithm(B, H, W, Cin, Cout, 0, prec)
    packed = torch.empty(C.conv3x3_packed_floats(Cout, Cin, p), device="cuda")
    C.call("ngan_conv3x3_pack_weights", w, packed, Cout, Cin, 0, scale, p)
    y = torch.empty(B, H, W, Cout, device="cuda"); rn = torch.empty(B, H, W, device="cuda")
    C.call("ngan_conv3x3_fwd", x, packed, None, y, rn, B, H, W, Cin, Cout, 0, 1, 0, 0.2, 1e-8, p, 0)
    gc = torch.empty_like(y)
    C.call("ngan_lrelu_pixelnorm_bwd", gy, None, y, rn, gc, B * H * W, Cout, 0.2)
    torch.cuda.synchronize()
    e = lambda a_, b_: float((a_.double() - b_).abs().max() / b_.abs().max())
    ge = (gc.double() - gc_ref).abs()
    idx = ge.flatten().argmax().item()
    print(f"prec{prec}: y {e(y, y_ref):.2e}  rn {e(rn, r_ref):.2e}  gc {e(gc, gc_ref):.2e}  min r_ref {float(r_ref.min()):.3e}; worst gc at r={float(r_ref.flatten()[idx // Cout]):.3e}, |gc_ref| max {float(gc_ref.abs().max()):.3e}")
