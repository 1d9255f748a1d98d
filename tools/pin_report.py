"""Element-wise deviation of the full-width gradient pins (tests/golden/full_C*.npz: first 96 entries + largest entry of every
gradient tensor, from the reference) per tensor, for both conv arithmetic modes:  python tools/pin_report.py  (GPU box)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
from conftest import load_golden
import test_gpu_models as T
ngan = load_package()
DEV = "cuda:0"
for name in ["full_C1", "full_C2", "full_C3", "full_C4", "full_C5"]:
    fix = load_golden(name)
    res, alpha, init, latent, batch, lr = fix["meta"]
    for prec in ("f32", "bf16x3"):
        ngan.ops.set_conv_precision(prec)
        torch.manual_seed(1)
        G = ngan.models.Generator_PG(ngan.config.N_gen_features, image_size_init=16)
        D = ngan.models.Discriminator_PG(ngan.config.N_dis_features, image_size_init=16)
        if int(res) != 16:
            G.set_resolution(int(res), float(alpha)); D.set_resolution(int(res), float(alpha))
        torch.manual_seed(123)
        x = torch.rand(int(batch), 1, int(res), int(res)) * 2 - 1
        G.to(DEV); D.to(DEV)
        fx = dict(fix); fx["real"] = x.numpy()
        G.zero_grad()
        g_pre, _ = ngan.loss_functions.G_W_loss(G, D)(x.to(DEV), z=torch.from_numpy(fix["z_g"]).to(DEV))
        g_pre.backward()
        rows = []
        for k, p in G.named_parameters():
            if p.grad is not None and "sl/Ggrad_pre/" + k in fix:
                flat = p.grad.double().cpu().numpy().reshape(-1); sl = fix["sl/Ggrad_pre/" + k]; mx = fix["mx/Ggrad_pre/" + k]
                rows.append((float(np.abs(flat[:sl.size] - sl).max() / mx[2]), "Ggrad_pre/" + k))
        G.zero_grad(); D.zero_grad()
        scal, norms, dgrads, ggrads = T.run_step_losses(ngan, G, D, fx)
        for k, g in dgrads.items():
            if "sl/Dgrad/" + k in fix:
                flat = g.astype(np.float64).reshape(-1); sl = fix["sl/Dgrad/" + k]; mx = fix["mx/Dgrad/" + k]
                rows.append((float(np.abs(flat[:sl.size] - sl).max() / mx[2]), "Dgrad/" + k))
        rows.sort(reverse=True)
        print(name, prec, "worst element-wise deviation / max-norm:", ", ".join(f"{k} {v:.1e}" for v, k in rows[:5]), flush=True)
ngan.ops.set_conv_precision("f32")
