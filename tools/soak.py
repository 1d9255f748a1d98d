"""Soak run of the step driver at the headline shape: N graph-replayed iterations on changing reals, then the same number of eager
ones; prints ms per iteration per block of 250, the losses at the block ends, allocated device memory and the parameter norms, and
fails if anything is not finite, memory grows after the first block or a block is more than 5 % slower than the first.
    python tools/soak.py [--iters 2000] [--res 512] [--batch 16] [--precision f32|bf16x3|bf16]          (on the GPU box)"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=2000)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16"])
args = ap.parse_args()
pkg = load_package()
pkg.ops.set_conv_precision(args.precision)
dev = torch.device("cuda", 0)
torch.manual_seed(1)
G, D = bench.build_nets(pkg, args.res, 1.0, dev)
tr = pkg.train.PGGANTrainer(G, D, learning_rate=1e-4, beta1=0.5, grad_pen_lambda=10.0, drift_epsilon=0.001, device_latents=True)
pool = [(torch.rand(args.batch, 1, args.res, args.res) * 2 - 1).to(dev) for _ in range(8)]
tr.capture(pool[0])
block = 250
for mode in ("graph replay", "eager"):
    first_ms = first_mem = None
    for b0 in range(0, args.iters, block):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(b0, min(b0 + block, args.iters)):
            stats = tr.replay(pool[i % len(pool)]) if mode == "graph replay" else tr.train_iteration(pool[i % len(pool)])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / (min(b0 + block, args.iters) - b0)
        vals = {k: float(v) for k, v in stats.items()}
        mem = torch.cuda.memory_allocated() / 2**20
        pn = (float(tr.flat_g.flat.norm()), float(tr.flat_d.flat.norm()))
        print(f"{mode:12s} iterations {b0:5d}-{min(b0 + block, args.iters) - 1:5d}: {ms:7.3f} ms/iteration  allocated {mem:8.1f} MiB  |G| {pn[0]:.4f} |D| {pn[1]:.4f}  "
              + "  ".join(f"{k} {v:+.4f}" for k, v in vals.items()), flush=True)
        assert all(v == v and abs(v) < 1e6 for v in vals.values()) and all(p == p for p in pn), "not finite"
        if first_ms is None:
            first_ms, first_mem = ms, mem
        else:
            assert mem <= first_mem + 1.0, f"device memory grew: {first_mem} -> {mem} MiB"
            assert ms <= 1.05 * first_ms, f"slowed down: {first_ms} -> {ms} ms"
print("soak ok")
