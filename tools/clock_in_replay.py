"""The shader clock INSIDE the replayed iteration: the phase-timer build's counters of wgrad_f32_kernel and conv3x3_tile_kernel accumulate over every
launch of a replay (shader cycles and s_memrealtime lifetimes of the sampled waves), so their ratio is the clock those kernels ran at while the
whole graph runs back to back -- next to the same kernels launched alone (tools/wgrad_phases.py).
    make -C neuron-gan_amd/csrc phases ; NGAN_LIB_PATH=build/phases/libngan_hip_phases.so python tools/clock_in_replay.py     (record: profiles/r04_clock_probe.txt)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

pkg = load_package()
lib = ctypes.CDLL(os.environ["NGAN_LIB_PATH"])
dev = torch.device("cuda", 0)
torch.manual_seed(1)
G, D = bench.build_nets(pkg, 512, 1.0, dev)
tr = pkg.train.PGGANTrainer(G, D, learning_rate=1e-4, beta1=0.5, grad_pen_lambda=10.0, drift_epsilon=0.001, device_latents=True)
real = (torch.rand(16, 1, 512, 512) * 2 - 1).to(dev)
tr.capture(real)
for _ in range(20):
    tr.replay(real)
torch.cuda.synchronize()
w, t = (ctypes.c_ulonglong * 15)(), (ctypes.c_ulonglong * 11)()
assert lib.ngan_diag_wgrad_phases(w, 1) == 0 and lib.ngan_diag_tile_phases(t, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    tr.replay(real)
e1.record()
torch.cuda.synchronize()
assert lib.ngan_diag_wgrad_phases(w, 1) == 0 and lib.ngan_diag_tile_phases(t, 1) == 0
print(f"50 replayed iterations, {e0.elapsed_time(e1) / 50:.3f} ms each (phase-timer build)")
WG = ["first barrier", "waiting for the tile's loads", "LDS writes (transposing store)", "second barrier", "issuing the next tile's loads", "operand reads + transforms + MFMAs",
      "the tail's first barrier", "own back-transform + 9 taps to LDS + barrier", "sum over the row-group waves + slab store", "(unused)"]
TL = ["first barrier", "waiting for the tile's loads + LDS writes", "second barrier", "issuing the next tile's loads + epilogue scalars", "transforms + MFMAs", "epilogue arithmetic + stores (incl. the MFMA drain)"]
for name, c, names, cn, o in (("wgrad_f32_kernel (all Winograd weight-gradient launches)", w, WG, 10, 11), ("conv3x3_tile_kernel (all 16 -> 16 forward / input-gradient launches)", t, TL, 6, 7)):
    nph = len(names)
    cyc, life = sum(c[:nph]), (c[o + 3] - c[o + 2]) / 100.0     # shader cycles; microseconds
    print(f"  {name}: {c[cn]} sampled waves, {cyc / c[cn]:.0f} shader cycles and {life / c[cn]:.1f} us per wave on average: {cyc / life / 1e3:.2f} GHz")
    print("      " + "; ".join(f"{n} {c[i] / cyc * 100:.1f} %" for i, n in enumerate(names) if c[i] / cyc > 0.001))
