"""Which tensors differ between an eager trajectory and a graph-replayed one (same weights, reals, latents, epsilon)?
   python tools/replay_diag.py            (GPU box)"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
from conftest import load_golden  # noqa: E402
import test_gpu_models as T  # noqa: E402

ngan = load_package()
dev = torch.device("cuda:0")
fix = load_golden("small_res16_fade_warm")
batch, latent, res = int(fix["meta"][4]), int(fix["meta"][3]), int(fix["meta"][0])
gen = torch.Generator().manual_seed(77)
steps = []
for _ in range(2):
    z = [torch.randn(batch, latent, generator=gen) for _ in range(3)]
    z = [(v / v.norm(dim=1, keepdim=True)).to(dev) for v in z]
    steps.append(dict(real=(torch.rand(batch, 1, res, res, generator=gen) * 2 - 1).to(dev), z_d=z[0], z_gp=z[1],
                      eps=torch.rand(batch, 1, 1, 1, generator=gen).to(dev), z_g=z[2]))


def make(force):
    G, D = T.build_small(ngan, fix)
    tr = ngan.train.PGGANTrainer(G, D, learning_rate=1e-3)
    if force:
        tr.force_exchange = True
        tr.enable_stem_exchange()
    return tr


def eager(force, n=2):
    tr = make(force)
    for s in steps[:n]:
        tr.train_iteration(s["real"], s["z_d"], s["z_gp"], s["eps"], s["z_g"])
    torch.cuda.synchronize()
    return tr


def replayed(force, n=2):
    tr = make(force)
    static = {k: steps[0][k].clone() for k in ("z_d", "z_gp", "eps", "z_g")}
    tr.capture(steps[0]["real"], warmup=2, draws=static)
    for s in steps[:n]:
        for k, v in static.items():
            v.copy_(s[k])
        tr.replay(s["real"])
    torch.cuda.synchronize()
    return tr


def diff(a, b, what):
    worst = []
    for flat_a, flat_b in ((a.flat_g, b.flat_g), (a.flat_d, b.flat_d)):
        for name, p, q in zip(flat_a.names, flat_a.params, flat_b.params):
            d = float((p - q).abs().max())
            if d > 0:
                worst.append((d, name, int(((p - q) != 0).sum()), p.numel()))
    print(f"{what}: {len(worst)} tensors differ" + "".join(f"\n    {n}: max {d:.3e}, {k}/{tot} elements" for d, n, k, tot in sorted(worst, reverse=True)[:12]), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "dist":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    dist.init_process_group("gloo", rank=0, world_size=1)
    diff(eager(True), eager(True), "eager vs eager (exchange path)")
    diff(eager(True, 1), replayed(True, 1), "eager vs 3-segment replay, 1 step")
    diff(eager(True), replayed(True), "eager vs 3-segment replay, 2 steps")
else:
    diff(eager(False), eager(False), "eager vs eager")
    diff(eager(False, 1), replayed(False, 1), "eager vs single-graph replay, 1 step")
    diff(eager(False), replayed(False), "eager vs single-graph replay, 2 steps")
