"""How far do the generator's init-state gradients (config C3: 256x256, batch 32) move between arithmetic variants of the critic's
first layer pair?  Prints, per generator parameter, |sum|g|| deviation from the golden fixture for
{f32, bf16x3} x {FirstBlock fused, unfused}.  Run on the GPU box:  python tools/first_block_sensitivity.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

ngan = importlib.import_module("neuron-gan_amd")
fix = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "full_C3.npz")))
res, alpha, init, latent, batch, lr = fix["meta"]
res, batch = int(res), int(batch)
rows = {}
for prec in ("f32", "bf16x3"):
    for fused in (True, False):
        ngan.ops.set_conv_precision(prec)
        ngan.ops._first_block_allowed = fused
        torch.manual_seed(1)
        G = ngan.models.Generator_PG(ngan.config.N_gen_features, image_size_init=16)
        D = ngan.models.Discriminator_PG(ngan.config.N_dis_features, image_size_init=16)
        G.set_resolution(res, float(alpha)); D.set_resolution(res, float(alpha))
        G.cuda(); D.cuda()
        torch.manual_seed(123)
        x = (torch.rand(batch, 1, res, res) * 2 - 1).cuda()
        loss, _ = ngan.loss_functions.G_W_loss(G, D)(x, z=torch.from_numpy(fix["z_g"]).cuda())
        loss.backward()
        for k, p in G.named_parameters():
            if p.grad is not None:
                cs = fix["cs/Ggrad_pre/" + k]
                rows.setdefault(k, []).append(abs(float(p.grad.double().abs().sum()) - cs[1]) / cs[1])
print(f"{'parameter':40s} f32/fused  f32/unfused  bf16x3/fused  bf16x3/unfused")
for k, v in rows.items():
    print(f"{k:40s} " + "  ".join(f"{e:10.2e}" for e in v))
