"""Per-shape timing of the bf16-storage conv kernels (ngan_bf16_conv3x3_fwd / _wgrad) on the layer shapes of the 512x512 iteration:
us per launch, algorithmic GB/s (input read once, output written once, 2 bytes per activation element) and the fraction of 8 TB/s.
    python tools/bf16_micro.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--graph", type=int, default=1, help="1: time the launches inside a replayed HIP graph (default), 0: an eager loop")
a = ap.parse_args()
pkg = load_package()
C, ops = pkg._C, pkg.ops
ops.set_conv_precision("bf16")
dev = "cuda:0"
BF = torch.bfloat16
# (tag, B, H, W, K, N, resample, epilogue, out_mode)
SHAPES = [
    ("G last 16->16 @512 fwd", 16, 512, 512, 16, 16, 0, 1, 0), ("G up2 16->16 @512 fwd", 16, 512, 512, 16, 16, 2, 1, 0),
    ("dgrad+pnbwd 16->16 @512", 16, 512, 512, 16, 16, 0, 2, 0), ("plain 16->16 @512", 16, 512, 512, 16, 16, 0, 0, 0),
    ("D 16->16 @256 b32 fwd", 32, 256, 256, 16, 16, 0, 1, 0), ("D pooled 16->32 @128 b32", 32, 128, 128, 16, 32, 1, 1, 0),
    ("pool-adjoint+pnbwd 32->16 @128->256 b32", 32, 128, 128, 32, 16, 0, 2, 1),
    ("32->32 @128 b32 fwd", 32, 128, 128, 32, 32, 0, 1, 0), ("32->32 @64 b32 fwd", 32, 64, 64, 32, 32, 0, 1, 0),
    ("G up2 32->16 @256", 16, 256, 256, 32, 16, 2, 1, 0), ("G up2 32->32 @128", 16, 128, 128, 32, 32, 2, 1, 0),
    ("32->64 pooled @32 b32", 32, 32, 32, 32, 64, 1, 1, 0), ("64->64 @32 b32", 32, 32, 32, 64, 64, 0, 1, 0),
    ("64->128 pooled @16 b32", 32, 16, 16, 64, 128, 1, 1, 0), ("128->128 @16 b32", 32, 16, 16, 128, 128, 0, 1, 0),
    ("128->128 @16 b16", 16, 16, 16, 128, 128, 0, 1, 0), ("128->64 up2 @32 b16", 16, 32, 32, 128, 64, 2, 1, 0),
    ("64->32 up2 @64 b16", 16, 64, 64, 64, 32, 2, 1, 0), ("128->128 dgrad+pnbwd @16 b32", 32, 16, 16, 128, 128, 0, 2, 0),
]
WG = [("wgrad 16x16 @512", 16, 512, 512, 16, 16, 0), ("wgrad 16x16 @256 b32", 32, 256, 256, 16, 16, 0), ("wgrad 32x32 @128 b32", 32, 128, 128, 32, 32, 0),
      ("wgrad 16->16 up2 @512", 16, 512, 512, 16, 16, 2), ("wgrad 128x128 @16 b32", 32, 16, 16, 128, 128, 0), ("wgrad 64x64 @32 b32", 32, 32, 32, 64, 64, 0)]


def timeit(run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    if a.graph:          # the launches captured into one HIP graph: kernel-bound timing (an eager loop is bound by ~10 us of Python per call)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(a.iters):
                run()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.iters
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.iters


for tag, B, H, W, K, N, res, epi, out in SHAPES:
    if a.only and a.only not in tag:
        continue
    hin, win = (2 * H, 2 * W) if res == 1 else ((H // 2, W // 2) if res == 2 else (H, W))
    x = torch.randn(B, hin, win, K, device=dev).to(BF)
    w = torch.randn(N, K, 3, 3, device=dev)
    packed = ops._packed(w, 1 if epi == 2 else 0, 0.1, 5)
    oh, ow = (2 * H, 2 * W) if out else (H, W)
    y = torch.empty(B, oh, ow, N, device=dev, dtype=BF)
    rn = torch.empty(B, H, W, device=dev)
    ay = torch.randn(B, oh, ow, N, device=dev).to(BF) if epi == 2 else None
    arn = (torch.rand(B, oh, ow, device=dev) + 0.5) if epi == 2 else None
    us = timeit(lambda: C.call("ngan_bf16_conv3x3_fwd", x, packed, None, y, rn if epi == 1 else None, ay, arn, None, B, H, W, K, N, res, epi, out, 0.2, 1e-8))
    nbytes = 2.0 * (x.numel() + y.numel()) + (4.0 * rn.numel() if epi == 1 else 0) + ((2.0 * ay.numel() + 4.0 * arn.numel()) if epi == 2 else 0)
    name = C.conv3x3_kernel_name(B, H, W, K, N, res, epi, out, 5)
    print(f"{tag:42s} {name:44s} {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s  {nbytes / us / 8e6:5.2f} of 8 TB/s   {2.0 * 9 * K * N * B * H * W / us / 1e6:7.1f} TF", flush=True)
for tag, B, H, W, K, N, res in WG:
    if a.only and a.only not in tag:
        continue
    hin, win = (2 * H, 2 * W) if res == 1 else ((H // 2, W // 2) if res == 2 else (H, W))
    x = torch.randn(B, hin, win, K, device=dev).to(BF)
    g = torch.randn(B, H, W, N, device=dev).to(BF)
    gw = torch.empty(N, K, 3, 3, device=dev)
    ws = torch.empty(C.wgrad_workspace_bytes(B, H, W, K, N) // 4, device=dev)
    us = timeit(lambda: C.call("ngan_bf16_conv3x3_wgrad", x, g, gw, ws, B, H, W, K, N, res, 0.1, 0))
    nbytes = 2.0 * (x.numel() + g.numel())
    print(f"{tag:42s} {C.conv3x3_wgrad_kernel_name(B, H, W, K, N, res, 5):44s} {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s  {nbytes / us / 8e6:5.2f} of 8 TB/s   (incl. the slab reduction)", flush=True)
