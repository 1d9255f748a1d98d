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
ap.add_argument("--op", default="wgrad", choices=["wgrad", "fwd"], help="fwd: conv3x3_tile_kernel (16 -> 16 forward / input gradient on whole 32-pixel tiles)")
ap.add_argument("--epi", type=int, default=1)
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
reader, ncnt = lib.ngan_diag_wgrad_phases, 15
if a.op == "fwd":
    ops = pkg.ops
    NAMES = ["first barrier (waiting for the other waves' MFMAs / epilogues)", "waiting for the tile's loads + LDS writes", "second barrier",
             "issuing the next tile's loads + epilogue scalars / operand requests", "transforms + MFMAs", "epilogue arithmetic + stores"]
    w = torch.randn(a.N, a.K, 3, 3, device=dev)
    prec = C.conv3x3_algorithm(a.B, a.H, a.W, a.K, a.N, 0, 0)
    packed = ops._packed(w, 1 if a.epi == 2 else 0, 0.1, prec)
    y = torch.empty(a.B, a.H, a.W, a.N, device=dev)
    rnb = torch.empty(a.B, a.H, a.W, device=dev)
    ay = torch.randn(a.B, a.H, a.W, a.N, device=dev) if a.epi == 2 else None
    arn = (torch.rand(a.B, a.H, a.W, device=dev) + 0.5) if a.epi == 2 else None
    run = lambda: C.call("ngan_conv3x3_fwd_ex", x, packed, None, y, rnb if a.epi == 1 else None, ay, arn, None, a.B, a.H, a.W, a.K, a.N, 0, a.epi, 0, 0.2, 1e-8, prec, 0)
    reader, ncnt = lib.ngan_diag_tile_phases, 11
for _ in range(3):
    run()
out = (ctypes.c_ulonglong * ncnt)()
assert reader(out, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    run()
e1.record()
torch.cuda.synchronize()
assert reader(out, 1) == 0
us = e0.elapsed_time(e1) * 1e3 / a.iters
nph = 10 if a.op == "wgrad" else 6
tot = float(sum(out[:nph]))
waves = out[nph] / a.iters
print(f"{a.op} B{a.B} {a.H}x{a.W} K{a.K} N{a.N} res{a.res}: {us:.1f} us per launch (with the stamps), {waves:.0f} waves sampled per launch (one workgroup in eight), "
      f"{tot / out[nph]:.0f} shader clocks per wave between the first tile and the end")
for i, n in enumerate(NAMES):
    print(f"  {out[i] / tot * 100:5.1f} %  {out[i] / out[nph]:9.0f} clk/wave  {n}")
# wall-clock picture of ONE launch's sampled waves (s_memrealtime, 100 MHz) next to their shader-clock count: the clock the kernel ran at
out2 = (ctypes.c_ulonglong * ncnt)()
assert reader(out2, 1) == 0
run()
torch.cuda.synchronize()
assert reader(out2, 1) == 0
cn, o = (6, 7) if a.op == "fwd" else (10, 11)          # index of the wave count; of the first realtime counter
n = out2[cn]
life_us = (out2[o + 3] - out2[o + 2]) / n / 100.0
clk = sum(out2[:nph]) / n
print(f"  one launch: {n} sampled waves; first loop entry to last exit {(out2[o + 1] - out2[o]) / 100.0:.1f} us; mean entry +{(out2[o + 2] / n - out2[o]) / 100.0:.1f} us after the first; "
      f"mean life {life_us:.1f} us for {clk:.0f} shader clocks = {clk / life_us / 1e3:.2f} GHz")
