"""Phase-by-phase wave time of wgrad_f32_kernel (phase-timer build: `make -C neuron-gan_amd/csrc phases`, counters in conv3x3_wgrad.hip).
   NGAN_LIB_PATH=build/phases/libngan_hip_phases.so python tools/wgrad_phases.py [--B 16 --H 256 --W 256 --K 16 --N 16]
Prints, per phase, the share of the summed wave lifetimes inside the tile loop and tail (record: profiles/r04_wgrad_phases.txt)."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

NAMES = ["first barrier (waiting for the other waves' MFMAs)", "waiting for this tile's global loads", "LDS writes (transposing store)",
         "second barrier", "issuing the next tile's loads", "operand reads + transforms + MFMAs", "the tail's first barrier (skew of the waves at the end of the loop)",
         "the wave's own back-transform, its 9 taps to LDS, barrier", "sum over the row-group waves + slab store", "(unused)"]

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=16)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=256)
ap.add_argument("--K", type=int, default=16)
ap.add_argument("--N", type=int, default=16)
ap.add_argument("--res", type=int, default=0)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
pkg = load_package()
C = pkg._C
lib = ctypes.CDLL(os.environ["NGAN_LIB_PATH"])
dev = "cuda:0"
hin, win = (2 * a.H, 2 * a.W) if a.res == 1 else ((a.H // 2, a.W // 2) if a.res == 2 else (a.H, a.W))
x = torch.randn(a.B, hin, win, a.K, device=dev)
g = torch.randn(a.B, a.H, a.W, a.N, device=dev)
gw = torch.empty(a.N, a.K, 3, 3, device=dev)
ws = torch.empty(C.wgrad_workspace_bytes(a.B, a.H, a.W, a.K, a.N) // 4, device=dev)
run = lambda: C.call("ngan_conv3x3_wgrad", x, g, gw, ws, a.B, a.H, a.W, a.K, a.N, a.res, 0.1, 0, 0)
for _ in range(3):
    run()
out = (ctypes.c_ulonglong * 11)()
assert lib.ngan_diag_wgrad_phases(out, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    run()
e1.record()
torch.cuda.synchronize()
assert lib.ngan_diag_wgrad_phases(out, 1) == 0
us = e0.elapsed_time(e1) * 1e3 / a.iters
tot = float(sum(out[:10]))
waves = out[10] / a.iters
print(f"wgrad B{a.B} {a.H}x{a.W} K{a.K} N{a.N} res{a.res}: {us:.1f} us per launch (with the stamps), {waves:.0f} waves sampled per launch (one workgroup in eight), "
      f"{tot / out[10]:.0f} shader clocks per wave between the first tile and the end")
for i, n in enumerate(NAMES):
    print(f"  {out[i] / tot * 100:5.1f} %  {out[i] / out[10]:9.0f} clk/wave  {n}")
