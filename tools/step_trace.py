"""Per-launch timing of one eager training iteration, bucketed by (entry point, shape arguments).
   python tools/step_trace.py [--res 512] [--batch 16] [--precision bf16x3] [--iters 3]
Every C-ABI launch is bracketed with HIP events on the launch stream (the `_C.set_probe` hook bench.py uses), so the
table shows which LAYER SHAPES the time goes to -- rocprofv3 only names template instances."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--alpha", type=float, default=1.0)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--precision", default="bf16x3")
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--min-us", type=float, default=0.0)
a = ap.parse_args()

pkg = load_package()
pkg.ops.set_conv_precision(a.precision)
dev = torch.device("cuda:0")
G, D = bench.build_nets(pkg, a.res, a.alpha, dev)
tr = pkg.train.PGGANTrainer(G, D, device_latents=True)
x = (torch.rand(a.batch, 1, a.res, a.res) * 2 - 1).to(dev)


class Probe:
    def __init__(self):
        self.rec = []

    def wants(self, name, args):
        return True

    def add(self, name, args, e0, e1):
        key = (name,) + tuple(v for v in args if isinstance(v, int) and not isinstance(v, bool))
        self.rec.append((key, e0, e1))


for _ in range(2):
    tr.train_iteration(x)
torch.cuda.synchronize()
p = Probe()
pkg._C.set_probe(p)
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(a.iters):
    tr.train_iteration(x)
t1.record()
torch.cuda.synchronize()
pkg._C.set_probe(None)
agg = {}
for key, e0, e1 in p.rec:
    d = agg.setdefault(key, [0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values()) / a.iters
print(f"# {a.res}x{a.res} batch {a.batch} {a.precision}: {t0.elapsed_time(t1) / a.iters:.2f} ms/iteration eager (probe on), "
      f"{tot / 1e3:.2f} ms inside ngan launches")
print("# conv3x3_fwd args: B H W K N resample epilogue out_mode prec | wgrad: B H W Cin Cout resample accumulate prec")
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us / a.iters < a.min_us:
        continue
    print(f"{us / a.iters:8.1f} us/iter  {n / a.iters:5.1f} calls  {us / n:8.1f} us  {key[0][5:]:24s} {' '.join(str(v) for v in key[1:])}")
