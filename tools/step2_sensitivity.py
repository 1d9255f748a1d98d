"""Why the SECOND iteration cannot be pinned at 1e-3 (round-3 review, item 6; record: profiles/r04_step2_sensitivity.txt).

The first Adam step moves every weight by ~lr * sign(g) (bias correction makes m / sqrt(v) = +-1), so a gradient element that is zero
at rounding level takes the other sign in another arithmetic and its weight ends 2 lr away.  Two CPU-only measurements on the oracle
(oracle/pggan_oracle.py, the fixture's weights, reals and draws; nothing of the product is involved):

  (a) torch fp32 against torch fp64 -- the SAME algorithm, differing by fp32 rounding (~1e-7) in the first iteration's gradients:
      how far apart are the second iteration's |grad D| and critic gradients?
  (b) fp32, with Gaussian noise of sigma * max|g| per tensor added to the first iteration's gradients before Adam applies them
      (sigma = 2e-4 ... 1e-3: inside the north star's 1e-3 tolerance on gradients): the same question.

    python tools/step2_sensitivity.py full_C1 full_C2 [full_C3]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
from conftest import load_golden  # noqa: E402
from oracle import pggan_oracle as O  # noqa: E402


def run(ngan, name, dtype=torch.float32, sigma=0.0, seed=0):
    fix = load_golden(name)
    res, alpha, init, latent, batch, lr = fix["meta"]
    res, batch = int(res), int(batch)
    torch.manual_seed(1)
    G = ngan.models.Generator_PG(ngan.config.N_gen_features, image_size_init=16)
    D = ngan.models.Discriminator_PG(ngan.config.N_dis_features, image_size_init=16)
    if res != 16:
        G.set_resolution(res, float(alpha))
        D.set_resolution(res, float(alpha))
    pg = O.as_leaf_params({k: v.detach().clone() for k, v in G.state_dict().items()}, dtype)
    pd = O.as_leaf_params({k: v.detach().clone() for k, v in D.state_dict().items()}, dtype)
    spec = O.NetSpec(image_size_init=16, slope=0.2, alpha=float(alpha))
    og, od = O.make_adam(pg), O.make_adam(pd)
    torch.manual_seed(123)
    x = (torch.rand(batch, 1, res, res) * 2 - 1).to(dtype)
    t = lambda k: torch.from_numpy(fix[k]).to(dtype)
    gen = torch.Generator().manual_seed(seed)

    def noise(params):
        for v in params.values():
            if v.grad is not None and sigma > 0:
                v.grad.add_(torch.randn(v.grad.shape, generator=gen).to(dtype) * (sigma * float(v.grad.abs().max())))

    O.zero_grads(pd)
    d_loss, _, _ = O.d_w_loss(pg, spec, pd, spec, x, t("z_d"), 0.001)
    gp = O.grad_penalty(pg, spec, pd, spec, x, t("z_gp"), t("eps"), 10.0)
    (d_loss + gp).backward()
    noise(pd)
    od.step()
    O.zero_grads(pg)
    O.zero_grads(pd)
    O.g_w_loss(pg, spec, pd, spec, t("z_g")).backward()
    noise(pg)
    og.step()
    O.zero_grads(pd)
    d_loss, s_r, s_f = O.d_w_loss(pg, spec, pd, spec, x, t("s2/z_d"), 0.001)
    gp, norms = O.grad_penalty(pg, spec, pd, spec, x, t("s2/z_gp"), t("s2/eps"), 10.0, return_norms=True)
    (d_loss + gp).backward()
    return norms.detach().double().numpy(), {k: v.grad.detach().double().clone() for k, v in pd.items() if v.grad is not None}


def spread(a, b):
    n = float(np.abs(a[0] - b[0]).max() / b[0].max())
    worst = max((float((a[1][k] - b[1][k]).abs().max() / b[1][k].abs().max()), k) for k in b[1])
    return n, worst


def main():
    ngan = load_package()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    for name in sys.argv[1:] or ["full_C1", "full_C2"]:
        base = run(ngan, name)
        n, w = spread(base, run(ngan, name, torch.float64))
        print(f"{name}: (a) fp32 vs fp64, second iteration: |grad D| {n:.1e}, critic gradient elements {w[0]:.1e} of the tensor's max ({w[1]})", flush=True)
        for sigma in (2e-4, 1e-3):
            n, w = spread(run(ngan, name, sigma=sigma), base)
            print(f"{name}: (b) first-iteration gradients + N(0, ({sigma:g} max|g|)^2): |grad D| {n:.1e}, critic gradient elements {w[0]:.1e} ({w[1]})",
                  flush=True)


if __name__ == "__main__":
    main()
