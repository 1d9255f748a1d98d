"""Bit-compare conv3x3_tile_kernel against conv3x3_persist_kernel (NGAN_TILE_KERNEL=0) on random data for a list of shapes."""
import os, sys, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # B, H, W, K, N, epi, out_mode, prec
    (32, 128, 128, 16, 16, 1, 0, 0), (32, 64, 64, 32, 32, 1, 0, 0), (4, 64, 64, 32, 32, 1, 0, 0), (32, 64, 64, 32, 32, 0, 0, 0),
    (32, 128, 128, 16, 16, 1, 0, 1), (32, 64, 64, 32, 32, 1, 0, 1), (8, 256, 256, 16, 32, 1, 0, 0), (8, 256, 256, 32, 16, 1, 0, 0),
    (16, 128, 128, 16, 16, 2, 1, 0), (16, 64, 64, 32, 32, 2, 1, 0), (16, 64, 64, 32, 16, 0, 1, 0), (3, 100, 96, 16, 16, 1, 0, 0),
    (2, 512, 512, 16, 16, 3, 0, 0), (2, 512, 512, 16, 16, 3, 0, 1),
]
def run(path):
    from __graft_entry__ import load_package
    pkg = load_package(); C, ops = pkg._C, pkg.ops
    outs = []
    for (B, H, W, K, N, epi, om, pr) in SHAPES:
        torch.manual_seed(B + H + K + N + epi)
        prec = C.conv3x3_algorithm(B, H, W, K, N, 0, pr)
        x = torch.randn(B, H, W, K, device="cuda"); w = torch.randn(N, K, 3, 3, device="cuda")
        oh, ow = (2 * H, 2 * W) if om else (H, W)
        ay = torch.randn(B, oh, ow, N, device="cuda") if epi == 2 else (torch.randn(N, device="cuda") if epi == 3 else None)
        arn = (torch.rand(B, oh, ow, device="cuda") + 0.5) if epi == 2 else None
        aout = torch.zeros(B, H, W, device="cuda") if epi == 3 else None
        bias = torch.randn(N, device="cuda") if epi in (0, 1, 3) else None
        packed = ops._packed(w, 0, 0.1, prec)
        y = torch.zeros(B, oh, ow, N, device="cuda"); rn = torch.zeros(B, H, W, device="cuda")
        C.call("ngan_conv3x3_fwd_ex", x, packed, bias, y, rn if epi in (1, 3) else None, ay, arn, aout, B, H, W, K, N, 0, epi, om, 0.2, 1e-8, prec, 0)
        torch.cuda.synchronize()
        outs.append((y.cpu(), rn.cpu(), aout.cpu() if aout is not None else None))
    torch.save(outs, path)
if len(sys.argv) > 1:
    run(sys.argv[1]); sys.exit(0)
subprocess.run([sys.executable, __file__, "/tmp/a.pt"], env=dict(os.environ, NGAN_TILE_KERNEL="0"), check=True)
subprocess.run([sys.executable, __file__, "/tmp/b.pt"], env=dict(os.environ, NGAN_TILE_KERNEL="1"), check=True)
A, Bq = torch.load("/tmp/a.pt"), torch.load("/tmp/b.pt")
for s, a, b in zip(SHAPES, A, Bq):
    d = [float((u - v).abs().max()) if u is not None else 0.0 for u, v in zip(a, b)]
    bad = (a[0] - b[0]).abs() > 1e-5
    msg = ""
    if bad.any():
        i = bad.nonzero()
        msg = f" first bad {i[0].tolist()} rows {sorted(set(i[:,1].tolist()))[:6]} cols {sorted(set(i[:,2].tolist()))[:8]} ch {sorted(set(i[:,3].tolist()))[:8]} frac {float(bad.float().mean()):.4f}"
    print(s, "max diff y/rn/img:", d, msg, flush=True)
