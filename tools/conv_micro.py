"""Micro-benchmark of ngan_conv3x3_fwd / ngan_conv3x3_wgrad through the C ABI (for kernel tuning and PMC runs).
   python tools/conv_micro.py --op fwd --B 16 --H 512 --W 512 --K 16 --N 16 --res 0 --epi 1 --iters 20"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--op", default="fwd")
ap.add_argument("--B", type=int, default=16)
ap.add_argument("--H", type=int, default=512)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--K", type=int, default=16)
ap.add_argument("--N", type=int, default=16)
ap.add_argument("--res", type=int, default=0)
ap.add_argument("--epi", type=int, default=1)
ap.add_argument("--out", type=int, default=0)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--prec", type=int, default=0)
a = ap.parse_args()
pkg = load_package()
C, ops = pkg._C, pkg.ops
dev = "cuda:0"
hin, win = (2 * a.H, 2 * a.W) if a.res == 1 else ((a.H // 2, a.W // 2) if a.res == 2 else (a.H, a.W))
x = torch.randn(a.B, hin, win, a.K, device=dev)
w = torch.randn(a.N, a.K, 3, 3, device=dev)
flops = 2.0 * 9 * a.K * a.N * a.B * a.H * a.W
if a.op == "fwd":
    prec = C.conv3x3_algorithm(a.B, a.H, a.W, a.K, a.N, a.res, a.prec)
    packed = ops._packed(w, 1 if a.epi == 2 else 0, 0.1, prec)
    oh, ow = (2 * a.H, 2 * a.W) if a.out else (a.H, a.W)
    y = torch.empty(a.B, oh, ow, a.N, device=dev)
    rn = torch.empty(a.B, a.H, a.W, device=dev)
    ay = torch.randn(a.B, oh, ow, a.N, device=dev) if a.epi == 2 else (torch.randn(a.N, device=dev) if a.epi == 3 else None)
    arn = (torch.rand(a.B, oh, ow, device=dev) + 0.5) if a.epi == 2 else None
    aout = torch.empty(a.B, a.H, a.W, device=dev) if a.epi == 3 else None
    def run():
        C.call("ngan_conv3x3_fwd_ex", x, packed, None, y, rn if a.epi in (1, 3) else None, ay, arn, aout, a.B, a.H, a.W, a.K, a.N, a.res, a.epi, a.out,
               0.2, 1e-8, prec, C.CONV_SKIP_BORDER if prec == 3 else 0)
        if prec == 3:      # folded bilinear: the border ring is its own launch in the Python layer's split mode (include/ngan.h)
            C.call("ngan_conv3x3_up2_border", x, packed, None, y, rn if a.epi else None, a.B, a.H, a.W, a.K, a.N, a.epi, 0.2, 1e-8)
else:
    g = torch.randn(a.B, a.H, a.W, a.N, device=dev)
    gw = torch.empty(a.N, a.K, 3, 3, device=dev)
    ws = torch.empty(C.wgrad_workspace_bytes(a.B, a.H, a.W, a.K, a.N) // 4, device=dev)
    run = lambda: C.call("ngan_conv3x3_wgrad", x, g, gw, ws, a.B, a.H, a.W, a.K, a.N, a.res, 0.1, 0, a.prec)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
print(f"prec{a.prec} {a.op} B{a.B} {a.H}x{a.W} K{a.K} N{a.N} res{a.res} epi{a.epi} out{a.out}: {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s")
