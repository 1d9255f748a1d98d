"""How much of the post-critic-update generator gradient of the C2 pin (64x64, alpha 0.5, batch 64) is arithmetic noise?
Runs the pinned step in {f32, bf16x3} x {first-order fusion on, off} and prints the relative error of sum|g| per tensor."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402

ngan = load_package()
import test_gpu_models as T  # noqa: E402

fix = dict(np.load(os.path.join(ROOT, "tests", "golden", "full_C2.npz"), allow_pickle=False))
for prec in ("f32", "bf16x3"):
    for fusion in (True, False):
        ngan.ops.set_conv_precision(prec)
        ngan.ops.allow_first_order_fusion(fusion)
        torch.manual_seed(1)
        G = ngan.models.Generator_PG(ngan.config.N_gen_features, image_size_init=16)
        D = ngan.models.Discriminator_PG(ngan.config.N_dis_features, image_size_init=16)
        G.set_resolution(64, 0.5)
        D.set_resolution(64, 0.5)
        torch.manual_seed(123)
        x = torch.rand(64, 1, 64, 64) * 2 - 1
        G.to("cuda:0"); D.to("cuda:0")
        fx = dict(fix); fx["real"] = x.numpy()
        G.zero_grad()
        g_pre, _ = ngan.loss_functions.G_W_loss(G, D)(x.to("cuda:0"), z=torch.from_numpy(fix["z_g"]).to("cuda:0"))
        g_pre.backward()
        pre = {k: abs(float(p.grad.double().abs().sum()) - fix["cs/Ggrad_pre/" + k][1]) / fix["cs/Ggrad_pre/" + k][1]
               for k, p in G.named_parameters() if p.grad is not None}
        wp = sorted(pre.items(), key=lambda kv: -kv[1])[:3]
        print(f"{prec:7s} fusion={fusion!s:5s} G-grad BEFORE the D update, worst: " + ", ".join(f"{k}={v:.2e}" for k, v in wp))
        G.zero_grad(); D.zero_grad()
        scal, norms, dgrads, ggrads = T.run_step_losses(ngan, G, D, fx)
        errs = {k: abs(float(np.abs(g.astype(np.float64)).sum()) - fix["cs/Ggrad/" + k][1]) / fix["cs/Ggrad/" + k][1] for k, g in ggrads.items()}
        derr = {k: abs(float(np.abs(g.astype(np.float64)).sum()) - fix["cs/Dgrad/" + k][1]) / fix["cs/Dgrad/" + k][1] for k, g in dgrads.items()}
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print(f"{prec:7s} fusion={fusion!s:5s} max D-grad err {max(derr.values()):.2e}  G-grad (after D update) worst: " +
              ", ".join(f"{k}={v:.2e}" for k, v in worst))
