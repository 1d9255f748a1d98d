"""Shader clock and power of the GPU while the iteration's HIP graph is replayed back to back (rocm-smi sampled from a side thread):
what the chip sustains under THIS workload, next to its 2.4 GHz nominal clock.
    python tools/clock_under_load.py [--seconds 6] [--precision f32|bf16x3|bf16]        (record: profiles/r04_clock_under_load.txt)"""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=6.0)
ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16"])
args = ap.parse_args()
pkg = load_package()
pkg.ops.set_conv_precision(args.precision)
dev = torch.device("cuda", 0)
torch.manual_seed(1)
G, D = bench.build_nets(pkg, 512, 1.0, dev)
tr = pkg.train.PGGANTrainer(G, D, learning_rate=1e-4, beta1=0.5, grad_pen_lambda=10.0, drift_epsilon=0.001, device_latents=True)
real = (torch.rand(16, 1, 512, 512) * 2 - 1).to(dev)
tr.capture(real)
samples, stop = [], False


def smi():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
        except Exception as e:  # noqa: BLE001
            out = repr(e)
        sclk = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", out)
        pw = re.findall(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", out)
        tj = re.findall(r"Temperature \(Sensor junction\) \(C\): ([\d.]+)", out)
        samples.append((time.perf_counter(), sclk[:1], pw[:1], tj[:1]))
        time.sleep(0.25)


idle = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
th = threading.Thread(target=smi)
th.start()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < args.seconds:
    for _ in range(20):
        tr.replay(real)
    n += 20
    torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / n
stop = True
th.join()
print(f"{args.precision}: {n} replayed iterations in {args.seconds:.0f} s: {ms:.3f} ms per iteration")
print("idle before:", " | ".join(l.strip() for l in idle.splitlines() if "sclk" in l or "Power" in l)[:300])
for t, s, p, tj in samples:
    print(f"  t = {t - t0:5.2f} s   sclk {s}   power {p} W   junction {tj} C")
