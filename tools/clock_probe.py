"""In-kernel clock of the persistent 3x3 conv kernel (diagnostic build: `make -C neuron-gan_amd/csrc clockprobe`, run with
NGAN_LIB_PATH=neuron-gan_amd/libngan_hip_clockprobe.so):
    NGAN_LIB_PATH=neuron-gan_amd/libngan_hip_clockprobe.so python tools/clock_probe.py --prec 0 --seconds 2
Launches the same conv back to back for `--seconds` (the clock settles under sustained load), then reads the per-workgroup stamps
of the LAST launch: shader-clock cycles (s_memtime) and 100 MHz reference ticks (s_memrealtime) between kernel entry and exit.
Prints the wall time per launch (HIP events), the median in-kernel clock, and the spread of workgroup start / end times."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=16)
ap.add_argument("--H", type=int, default=512)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--K", type=int, default=16)
ap.add_argument("--N", type=int, default=16)
ap.add_argument("--res", type=int, default=0)
ap.add_argument("--epi", type=int, default=1)
ap.add_argument("--prec", type=int, default=0)
ap.add_argument("--seconds", type=float, default=2.0)
a = ap.parse_args()
pkg = load_package()
C, ops = pkg._C, pkg.ops
lib = C.lib()
if not hasattr(lib, "ngan_debug_clock_stamps"):
    raise SystemExit("this library has no clock stamps: build `make -C neuron-gan_amd/csrc clockprobe` and set NGAN_LIB_PATH")
dev = "cuda:0"
hin, win = ((a.H // 2, a.W // 2) if a.res == 2 else (a.H, a.W))
x = torch.randn(a.B, hin, win, a.K, device=dev)
w = torch.randn(a.N, a.K, 3, 3, device=dev)
prec = C.conv3x3_algorithm(a.B, a.H, a.W, a.K, a.N, a.res, a.prec)
packed = ops._packed(w, 0, 0.1, prec)
y = torch.empty(a.B, a.H, a.W, a.N, device=dev)
rn = torch.empty(a.B, a.H, a.W, device=dev)
run = lambda: C.call("ngan_conv3x3_fwd", x, packed, None, y, rn if a.epi else None, a.B, a.H, a.W, a.K, a.N, a.res, a.epi, 0, 0.2, 1e-8, prec,
                     C.CONV_SKIP_BORDER if prec == 3 else 0)
for _ in range(5):
    run()
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < a.seconds:
    for _ in range(200):
        run()
    torch.cuda.synchronize()
    n += 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 10.0
nb = 8192
buf = (ctypes.c_ulonglong * (4 * nb))()
lib.ngan_debug_clock_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.ngan_debug_clock_stamps(buf, nb) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 4).astype(np.float64)
s = s[s[:, 1] > 0]
cyc, ticks = s[:, 2] - s[:, 0], s[:, 3] - s[:, 1]
ghz = cyc / ticks * 0.1
start, end = s[:, 1] - s[:, 1].min(), s[:, 3] - s[:, 1].min()
flops = 2.0 * 9 * a.K * a.N * a.B * a.H * a.W
print(f"prec{prec} B{a.B} {a.H}x{a.W} K{a.K} N{a.N} res{a.res} epi{a.epi}: {us:.1f} us/launch = {flops / us / 1e6:.1f} TFLOP/s after {n} warm launches; "
      f"{len(s)} workgroups; in-kernel clock median {np.median(ghz):.3f} GHz (p5 {np.percentile(ghz, 5):.3f}, p95 {np.percentile(ghz, 95):.3f}); "
      f"workgroup lifetime median {np.median(ticks) / 100:.1f} us; first start -> last end {end.max() / 100:.1f} us; "
      f"start spread {start.max() / 100:.1f} us; end spread {(end.max() - end.min()) / 100:.1f} us")
