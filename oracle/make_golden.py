"""Generate golden vectors under tests/golden/ from the REFERENCE's own models.py.

Runs only in the build container (needs /root/reference).  Never runs on the GPU box; the
fixtures it writes are data (inputs, weights, expected outputs) and are committed.

What is imported from the reference: `models.Generator_PG`, `models.Discriminator_PG`
(/root/reference/models.py:272-616) -- it imports as-is (torch, numpy, configs only).
`loss_functions.py` is NOT imported: its `utils` import needs parse/torchvision/cv2, which are
not installed.  The three loss forwards (loss_functions.py:14-47, 59-74, 157-180) and the inner
loop (train.py:357-385) are therefore replayed by `ref_step` below over the imported reference
modules; an assert in `full_fixture` verifies that this replay reproduces the first-step values that
SURVEY.md 8(c) recorded from the reference's real loss modules (same seeds, same RNG order).

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--full]
"""
import argparse
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import models  # noqa: the reference's models.py
    from configs import config  # noqa
    return models, config


def latent(b, dim):
    z = torch.randn(b, dim).clamp(-5, 5)  # utils.py:77
    return z / z.norm(p=2, dim=1, keepdim=True)  # utils.py:78


def ref_losses(G, D, x, z_d, z_gp, eps, lam=10.0, drift=0.001):
    """D_W_loss + D_grad_pen_loss over reference modules; returns tensors with graph."""
    real_score = D(x)
    s_real = real_score.mean()
    fake = G(z_d).detach()
    fake_score = D(fake)
    s_fake = fake_score.mean()
    d_loss = -s_real + s_fake
    if drift > 0:
        d_loss = d_loss + drift * torch.square(real_score).mean()
    x_tilde = G(z_gp).detach()
    x_hat = eps * x + (1 - eps) * x_tilde
    x_hat.requires_grad_()
    out = D(x_hat)
    g = torch.autograd.grad(outputs=out.sum(), inputs=x_hat, create_graph=True)[0]
    norms = g.norm(2, dim=(1, 2, 3))
    gp = lam * torch.mean((norms - 1) ** 2)
    return dict(real_score=real_score, fake=fake, fake_score=fake_score, s_real=s_real, s_fake=s_fake, d_loss=d_loss,
                gp=gp, norms=norms, x_hat_grad=g)


def ref_step(G, D, oG, oD, x, z_d, z_gp, eps, z_g, lam=10.0, drift=0.001, capture=None):
    """train.py:357-385 with n_critic=1 over the reference modules."""
    if capture is not None:
        # generator gradients BEFORE the critic update: a pin that does not go through Adam's sign-like first step
        G.zero_grad()
        D.zero_grad()
        pre = -D(G(z_g)).mean()
        pre.backward()
        for k, p in G.named_parameters():
            if p.grad is not None:
                capture["Ggrad_pre/" + k] = p.grad.detach().clone().numpy()
        capture["G_loss_pre"] = np.array(float(pre.detach()))
        G.zero_grad()
    D.zero_grad()
    L = ref_losses(G, D, x, z_d, z_gp, eps, lam, drift)
    total = L["d_loss"] + L["gp"]
    total.backward()
    if capture is not None:
        for k, p in D.named_parameters():
            if p.grad is not None:
                capture["Dgrad/" + k] = p.grad.detach().clone().numpy()
    oD.step()
    G.zero_grad()
    D.zero_grad()
    fake = G(z_g)
    g_loss = -D(fake).mean()
    g_loss.backward()
    if capture is not None:
        for k, p in G.named_parameters():
            if p.grad is not None:
                capture["Ggrad/" + k] = p.grad.detach().clone().numpy()
        capture["fake_g"] = fake.detach().numpy()
    oG.step()
    return dict(D_loss=float(total.detach()), score_real=float(L["s_real"].detach()), score_fake=float(L["s_fake"].detach()),
                GP=float(L["gp"].detach()), G_loss=float(g_loss.detach())), L


def state_np(net, prefix):
    return {prefix + k: v.detach().clone().numpy() for k, v in net.state_dict().items()}


# ------------------------------------------------------------------------------------------
# reduced-width fixtures: every weight, input and output stored
# ------------------------------------------------------------------------------------------
SMALL = dict(g_widths=[32, 16, 16], d_widths=[16, 16, 32], image_size_init=4, latent_dim=32)


def small_fixture(models, name, res, alpha, warm_steps, batch=4, seed=11):
    torch.manual_seed(seed)
    G = models.Generator_PG(list(SMALL["g_widths"]), image_size_init=SMALL["image_size_init"], latent_dim=SMALL["latent_dim"])
    D = models.Discriminator_PG(list(SMALL["d_widths"]), image_size_init=SMALL["image_size_init"])
    G.to(torch.device("cpu"), torch.float32)
    D.to(torch.device("cpu"), torch.float32)
    if res != SMALL["image_size_init"]:
        G.set_resolution(res, alpha)
        D.set_resolution(res, alpha)
    lr = 1e-4
    if warm_steps:
        # warm the nets so that |grad D| is O(1) instead of ~1e-2 (last hard part)
        oG = torch.optim.Adam(G.parameters(), lr=2e-3, betas=(0.5, 0.999))
        oD = torch.optim.Adam(D.parameters(), lr=2e-3, betas=(0.5, 0.999))
        for _ in range(warm_steps):
            x = torch.rand(batch, 1, res, res) * 2 - 1
            ref_step(G, D, oG, oD, x, latent(batch, G.latent_dim), latent(batch, G.latent_dim),
                     torch.rand(batch, 1, 1, 1), latent(batch, G.latent_dim))
    oG = torch.optim.Adam(G.parameters(), lr=lr, betas=(0.5, 0.999))
    oD = torch.optim.Adam(D.parameters(), lr=lr, betas=(0.5, 0.999))
    out = {}
    out.update(state_np(G, "G/"))
    out.update(state_np(D, "D/"))
    x = torch.rand(batch, 1, res, res) * 2 - 1
    z_d, z_gp, z_g = (latent(batch, G.latent_dim) for _ in range(3))
    eps = torch.rand(batch, 1, 1, 1)
    out.update(real=x.numpy(), z_d=z_d.numpy(), z_gp=z_gp.numpy(), z_g=z_g.numpy(), eps=eps.numpy())
    with torch.no_grad():
        out["G_of_z_d"] = G(z_d).numpy()
        out["D_of_real"] = D(x).numpy()
    cap = {}
    scal, L = ref_step(G, D, oG, oD, x, z_d, z_gp, eps, z_g, capture=cap)
    out.update(cap)
    out["D_of_fake"] = L["fake_score"].detach().numpy()
    out["grad_norms"] = L["norms"].detach().numpy()
    out["x_hat_grad"] = L["x_hat_grad"].detach().numpy()
    out["scalars"] = np.array([scal[k] for k in ("D_loss", "score_real", "score_fake", "GP", "G_loss")], dtype=np.float64)
    out.update(state_np(G, "G_after/"))
    out.update(state_np(D, "D_after/"))
    out["meta"] = np.array([res, alpha, SMALL["image_size_init"], SMALL["latent_dim"], batch, lr], dtype=np.float64)
    path = os.path.join(OUT, f"small_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: scalars={out['scalars']}, norms={out['grad_norms']}")


# ------------------------------------------------------------------------------------------
# full-width pins (SURVEY.md 8c protocol): weights reconstructed from seed by the consumer
# ------------------------------------------------------------------------------------------
FULL = {  # name: (res, alpha, batch)
    "C1": (16, 1.0, 16), "C2": (64, 0.5, 64), "C3": (256, 1.0, 32), "C4": (512, 1.0, 16),
    "C5": (512, 1.0, 8),      # BASELINE.json's fifth config (512x512, batch 8 per GPU); no SURVEY.md pin exists for it
}
SURVEY_PINS = {  # (D_loss, score_real, score_fake, GP, G_loss) recorded in SURVEY.md 8(c)
    "C1": (9.621342659, -0.006367247, -0.002287695, 9.617262840, 0.009899071),
    "C2": (9.631142616, -0.018096786, -0.012321905, 9.625367165, 0.011696977),
    "C3": (9.966442108, -0.023956211, -0.021933945, 9.964419365, 0.028972290),
    "C4": (9.984798431, -0.018186904, -0.019845556, 9.986456871, 0.028318128),
}


def checksums(net):
    return {k: np.array([float(v.double().sum()), float(v.double().abs().sum())]) for k, v in net.state_dict().items()
            if v.numel() > 1}


def full_fixture(models, config, name):
    res, alpha, batch = FULL[name]
    torch.manual_seed(1)
    G = models.Generator_PG(config.N_gen_features, image_size_init=16)  # train.py:172
    D = models.Discriminator_PG(config.N_dis_features, image_size_init=16)  # train.py:184
    G.to(torch.device("cpu"), torch.float32)
    D.to(torch.device("cpu"), torch.float32)
    if res != 16:
        G.set_resolution(res, alpha)
        D.set_resolution(res, alpha)
    oD = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
    oG = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999))
    out = {}
    for k, v in checksums(G).items():
        out["Ginit_cs/" + k] = v
    for k, v in checksums(D).items():
        out["Dinit_cs/" + k] = v
    torch.manual_seed(123)
    x = torch.rand(batch, 1, res, res) * 2 - 1
    # RNG order of one iteration: z_d, z_gp, eps, z_g
    z_d = latent(batch, 512)
    z_gp = latent(batch, 512)
    eps = torch.rand(batch, 1, 1, 1)
    z_g = latent(batch, 512)
    with torch.no_grad():
        img = G(z_d)
        out["G_of_z_d_slice"] = img[:2, 0, :8, :8].numpy()
        out["G_of_z_d_stats"] = np.array([float(img.double().mean()), float(img.double().std())])
        out["D_of_real"] = D(x).numpy()
    cap = {}
    scal, L = ref_step(G, D, oG, oD, x, z_d, z_gp, eps, z_g, capture=cap)
    out["D_of_fake"] = L["fake_score"].detach().numpy()
    out["grad_norms"] = L["norms"].detach().numpy()
    for k, v in cap.items():
        if k.startswith(("Dgrad/", "Ggrad/", "Ggrad_pre/")):
            out["cs/" + k] = np.array([float(v.astype(np.float64).sum()), float(np.abs(v.astype(np.float64)).sum())])
            # element-wise pins besides the checksums: the first 96 entries of every gradient tensor, its largest entry and the
            # tensor's max-norm (the scale the comparison is made on)
            flat = v.reshape(-1)
            out["sl/" + k] = flat[:96].copy()
            out["mx/" + k] = np.array([float(np.argmax(np.abs(flat))), float(flat[np.argmax(np.abs(flat))]), float(np.abs(flat).max())])
    out["scalars"] = np.array([scal[k] for k in ("D_loss", "score_real", "score_fake", "GP", "G_loss")], dtype=np.float64)
    out["G_loss_pre"] = cap["G_loss_pre"]
    for k, v in checksums(G).items():
        out["Gafter_cs/" + k] = v
    for k, v in checksums(D).items():
        out["Dafter_cs/" + k] = v
    out.update(z_d=z_d.numpy(), z_gp=z_gp.numpy(), z_g=z_g.numpy(), eps=eps.numpy())
    out["real_cs"] = np.array([float(x.double().sum()), float(x.double().abs().sum())])
    out["meta"] = np.array([res, alpha, 16, 512, batch, 1e-4], dtype=np.float64)
    # A SECOND iteration on the same reals (round 4): the next four draws of the same RNG stream (order z_d, z_gp, eps, z_g; the
    # forward / backward passes consume none), through the weights and Adam state the first iteration left.  The first Adam step
    # moves every weight by ~lr * sign(g), which amplifies rounding-level gradient differences into individual flipped steps; the
    # second step's scalars and critic gradients say what is left of that one iteration later.  (Written after everything above, so
    # the keys of the earlier rounds keep their values bit for bit.)
    z_d2 = latent(batch, 512)
    z_gp2 = latent(batch, 512)
    eps2 = torch.rand(batch, 1, 1, 1)
    z_g2 = latent(batch, 512)
    cap2 = {}
    scal2, L2 = ref_step(G, D, oG, oD, x, z_d2, z_gp2, eps2, z_g2, capture=cap2)
    out["s2/scalars"] = np.array([scal2[k] for k in ("D_loss", "score_real", "score_fake", "GP", "G_loss")], dtype=np.float64)
    out["s2/grad_norms"] = L2["norms"].detach().numpy()
    out["s2/G_loss_pre"] = cap2["G_loss_pre"]
    for k, v in cap2.items():
        if k.startswith(("Dgrad/", "Ggrad/")):
            flat = v.reshape(-1)
            out["s2/cs/" + k] = np.array([float(v.astype(np.float64).sum()), float(np.abs(v.astype(np.float64)).sum())])
            out["s2/sl/" + k] = flat[:96].copy()
            out["s2/mx/" + k] = np.array([float(np.argmax(np.abs(flat))), float(flat[np.argmax(np.abs(flat))]), float(np.abs(flat).max())])
    out.update({"s2/z_d": z_d2.numpy(), "s2/z_gp": z_gp2.numpy(), "s2/z_g": z_g2.numpy(), "s2/eps": eps2.numpy()})
    # The same two iterations with the reference modules in fp64 (same weights, reals and draws, cast): how far the reference's OWN fp32
    # arithmetic is from the exact answer one update later.  The first Adam step is ~lr * sign(g), so gradient elements at rounding
    # level take the other sign in another arithmetic and the second iteration sees slightly different weights: this spread is a
    # property of the algorithm, and it is the yardstick for the second-iteration test (tests/test_gpu_models.py).
    torch.manual_seed(1)
    G64 = models.Generator_PG(config.N_gen_features, image_size_init=16)
    D64 = models.Discriminator_PG(config.N_dis_features, image_size_init=16)
    G64.to(torch.device("cpu"), torch.float64)
    D64.to(torch.device("cpu"), torch.float64)
    if res != 16:
        G64.set_resolution(res, alpha)
        D64.set_resolution(res, alpha)
    oD64 = torch.optim.Adam(D64.parameters(), lr=1e-4, betas=(0.5, 0.999))
    oG64 = torch.optim.Adam(G64.parameters(), lr=1e-4, betas=(0.5, 0.999))
    d = lambda t: t.double()
    ref_step(G64, D64, oG64, oD64, d(x), d(z_d), d(z_gp), d(eps), d(z_g))
    cap64 = {}
    scal64, L64 = ref_step(G64, D64, oG64, oD64, d(x), d(z_d2), d(z_gp2), d(eps2), d(z_g2), capture=cap64)
    out["s2/f64/scalars"] = np.array([scal64[k] for k in ("D_loss", "score_real", "score_fake", "GP", "G_loss")], dtype=np.float64)
    out["s2/f64/grad_norms"] = L64["norms"].detach().numpy()
    for k, v in cap64.items():
        if k.startswith(("Dgrad/", "Ggrad/")):
            flat = v.reshape(-1)
            i32 = int(out["s2/mx/" + k][0])                 # the SAME element the fp32 pin names, plus the tensor's own max-norm
            out["s2/f64/cs/" + k] = np.array([float(v.sum()), float(np.abs(v).sum())])
            out["s2/f64/sl/" + k] = flat[:96].copy()
            out["s2/f64/mx/" + k] = np.array([float(i32), float(flat[i32]), float(np.abs(flat).max())])
    if name in SURVEY_PINS:
        pins = np.array(SURVEY_PINS[name])
        rel = np.abs(out["scalars"] - pins) / np.abs(pins)
        print(f"full_{name}: scalars={out['scalars']}  max rel diff vs SURVEY pins = {rel.max():.2e}")
        assert rel.max() < 5e-6, "harness replay disagrees with the reference's own loss modules (SURVEY pins)"
    else:
        print(f"full_{name}: scalars={out['scalars']}  (same replay harness as the pinned configs; no SURVEY pin for this one)")
    np.savez_compressed(os.path.join(OUT, f"full_{name}.npz"), **out)


# ------------------------------------------------------------------------------------------
# checkpoint-format fixtures: files in the reference's dictionary layout (utils.py:160-169), written
# from reference modules, plus what the reference's own from_state_dict (models.py:394-444, 566-616) makes of them
# ------------------------------------------------------------------------------------------
def checkpoint_fixtures(models):
    os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"   # the reference calls torch.load without weights_only (torch 1.13)
    torch.manual_seed(21)
    # latent_dim stays at the config default: the reference's from_state_dict does not pass it to the constructor
    G = models.Generator_PG([16, 16, 16], image_size_init=4)
    D = models.Discriminator_PG([16, 16, 16], image_size_init=4)
    G.set_resolution(8, 1.0)
    D.set_resolution(8, 1.0)
    attrs = lambda m: {a: getattr(m, a) for a in m.saved_attrs}           # utils.get_saved_attrs
    ck = {"epoch": 7, "Generator_state": G.state_dict(), "Generator_attrs": attrs(G), "Discriminator_state": D.state_dict(),
          "Discriminator_attrs": attrs(D), "lr": 1e-4, "Loss_real": np.arange(7.0), "Loss_fake": -np.arange(7.0),
          "Loss_G": np.ones(7), "Loss_D": np.zeros(7)}
    new_path = os.path.join(OUT, "ref_checkpoint_new.pth")
    torch.save(ck, new_path)
    # old layout: merged ToIm_list / conv_block_list (FromIm_list) entries are still present, plus *_prev / *_conv_block modules
    gs, ds = G.state_dict(), D.state_dict()
    old_g = type(gs)()
    for k, v in gs.items():
        if k.startswith("ToIm_list.") or k.startswith("conv_block_list."):
            parts = k.split("."); parts[1] = str(int(parts[1]) + 1); k = ".".join(parts)   # one stale entry in front
        old_g[k] = v
    old_g["ToIm_list.0.layers.0.weight"] = torch.randn(1, 16, 1, 1)
    old_g["conv_block_list.0.1.weight"] = torch.randn(16, 16, 3, 3)
    old_g["conv_block_list.0.4.weight"] = torch.randn(16, 16, 3, 3)
    old_g["ToIm_prev.layers.0.weight"] = torch.randn(1, 16, 1, 1)
    old_g["last_conv_block.1.weight"] = torch.randn(16, 16, 3, 3)
    old_d = type(ds)()
    for k, v in ds.items():
        old_d[k] = v
    n_from = 1 + max(int(k.split(".")[1]) for k in ds if k.startswith("FromIm_list."))
    n_blk = 1 + max(int(k.split(".")[1]) for k in ds if k.startswith("conv_block_list."))
    old_d[f"FromIm_list.{n_from}.conv.weight"] = torch.randn(16, 1, 1, 1)     # one stale entry at the end
    old_d[f"FromIm_list.{n_from}.conv.bias"] = torch.randn(16)
    old_d[f"conv_block_list.{n_blk}.1.weight"] = torch.randn(16, 16, 3, 3)
    old_d[f"conv_block_list.{n_blk}.4.weight"] = torch.randn(16, 16, 3, 3)
    # (the reference's key parser needs a digit in every matched key, models.py:40, so the stale module is written as a list entry)
    old_d["FromIm_prev.0.conv.weight"] = torch.randn(16, 1, 1, 1)
    old_d["first_conv_block.1.weight"] = torch.randn(16, 16, 3, 3)
    ck_old = dict(ck, Generator_state=old_g, Discriminator_state=old_d)
    old_path = os.path.join(OUT, "ref_checkpoint_old.pth")
    torch.save(ck_old, old_path)
    out = {}
    for tag, path in (("new", new_path), ("old", old_path)):
        g2 = models.Generator_PG.from_state_dict(path, verbose=False)
        d2 = models.Discriminator_PG.from_state_dict(path, verbose=False)
        for k, v in g2.state_dict().items():
            out[f"{tag}/G/{k}"] = v.numpy()
        for k, v in d2.state_dict().items():
            out[f"{tag}/D/{k}"] = v.numpy()
        out[f"{tag}/meta"] = np.array([g2.image_size, float(g2.alpha), d2.image_size, float(d2.alpha)])
    torch.manual_seed(3)
    z = latent(3, 512)
    x = torch.rand(3, 1, 8, 8) * 2 - 1
    with torch.no_grad():
        out["z"], out["x"], out["G_of_z"], out["D_of_x"] = z.numpy(), x.numpy(), G(z).numpy(), D(x).numpy()
    np.savez_compressed(os.path.join(OUT, "ref_checkpoint_expected.npz"), **out)
    print("checkpoint fixtures written:", new_path, old_path)


# ------------------------------------------------------------------------------------------
# epoch-level fixture: what the reference's epoch loop does to the nets and the learning rate, epoch by epoch.
# The growth calls are the reference's own (models.py:355-392, 526-564), driven in the order of train.py:318-333; the learning
# rate follows update_lr (train.py:250-265), which lives in the module-level script train.py (not importable: parse/torchvision,
# prompts, dataset) and is restated here on a plain dict.
# ------------------------------------------------------------------------------------------
def epoch_fixture(models):
    transit_sch, n_epochs, alpha_step, base_lr = [3, 6], 9, 0.5, 1e-3
    torch.manual_seed(5)
    G = models.Generator_PG(list(SMALL["g_widths"]), image_size_init=SMALL["image_size_init"], latent_dim=SMALL["latent_dim"])
    D = models.Discriminator_PG(list(SMALL["d_widths"]), image_size_init=SMALL["image_size_init"])
    bounds = [0] + transit_sch + [n_epochs]                                            # train.py:243
    decay = [np.exp(np.log(1 / 100) / ((bounds[i + 1] - bounds[i]) / 2)) for i in range(len(bounds) - 1)]   # train.py:244-247
    group = {"lr": base_lr}

    def update_lr(epoch):                                                              # train.py:250-265
        if epoch in bounds:
            group["lr"] = base_lr
        else:
            phase = sum(epoch > t for t in transit_sch)
            n = bounds[phase + 1] - bounds[phase]
            since = epoch - bounds[phase]
            if since <= n / 2:
                group["lr"] = base_lr * (decay[phase] ** since)

    update_lr(0)                                                                       # train.py:288-289 (epoch_init - 1)
    out = {"meta": np.array([n_epochs, alpha_step, base_lr] + transit_sch, dtype=np.float64)}
    rows, gkeys, dkeys = [], [], []
    for epoch in range(1, n_epochs + 1):                                               # train.py:312
        lr = group["lr"]                                                               # train.py:316: the rate this epoch trains with
        if G.alpha < 1 and D.alpha < 1:                                                # train.py:319-321
            G.advance_transition(alpha_step)
            D.advance_transition(alpha_step)
        if epoch in transit_sch:                                                       # train.py:328-330
            G.increase_resolution()
            D.increase_resolution()
        rows.append([epoch, float(G.alpha), float(D.alpha), G.image_size, D.image_size, G.N_layers, D.N_layers, lr])
        gkeys.append("|".join(G.state_dict().keys()))
        dkeys.append("|".join(D.state_dict().keys()))
        update_lr(epoch)                                                               # train.py:425-426
    out["rows"] = np.array(rows, dtype=np.float64)
    out["G_keys"] = np.array(gkeys)
    out["D_keys"] = np.array(dkeys)
    np.savez_compressed(os.path.join(OUT, "epochs_small.npz"), **out)
    print("epochs_small.npz:", out["rows"][:, [0, 1, 3, 7]].tolist())


# ------------------------------------------------------------------------------------------
# sampling / eval: utils.gen_samples (utils.py:346-355) and the image part of plot_gen_samples (utils.py:568-601)
# over the reference's own Generator_PG.  utils.py does not import here (torchvision), so its seeded latent branch
# (utils.py:57-92: save the global RNG state, manual_seed(seed), draw, restore) is replayed line by line below.
# ------------------------------------------------------------------------------------------
def sampling_fixture(models, seed=3, n_images=6):
    torch.manual_seed(23)
    G = models.Generator_PG(list(SMALL["g_widths"]), image_size_init=SMALL["image_size_init"], latent_dim=SMALL["latent_dim"])
    G.to(torch.device("cpu"), torch.float32)
    G.set_resolution(8, 1.0)                                  # below image_size_max = 16: the enlargement branch runs
    out = state_np(G, "G/")
    torch.manual_seed(99)
    before = torch.get_rng_state()
    rng_state = torch.get_rng_state()                          # utils.py:66-67
    torch.manual_seed(seed)
    z = latent(n_images, G.latent_dim)                         # utils.py:77-78
    torch.set_rng_state(rng_state)                             # utils.py:84
    assert torch.equal(before, torch.get_rng_state())
    was_training = G.training
    G.train(False)                                             # utils.py:571-572
    with torch.no_grad():
        images = G(z).detach()                                 # utils.py:352-353
    G.train(was_training)
    enlarged = torch.nn.functional.interpolate(images.cpu(), size=(G.image_size_max, G.image_size_max))   # utils.py:598-601
    out.update(z=z.numpy(), images=images.numpy(), enlarged=enlarged.numpy(),
               meta=np.array([8, seed, n_images, G.image_size_max, SMALL["image_size_init"], SMALL["latent_dim"]], dtype=np.float64))
    np.savez_compressed(os.path.join(OUT, "sampling_small.npz"), **out)
    print("sampling_small.npz: images", tuple(images.shape), "mean", float(images.mean()), "enlarged", tuple(enlarged.shape))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also regenerate the full-width C3/C4 pins (about a minute of CPU)")
    ap.add_argument("--only", default="", help="comma list of fixture names")
    args = ap.parse_args()
    models, config = import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    small = [("fresh4", 4, 1.0, 0), ("res8_init", 8, 1.0, 0), ("res8_warm", 8, 1.0, 40),
             ("res16_fade_init", 16, 0.5, 0), ("res16_fade_warm", 16, 0.5, 40), ("res16_warm", 16, 1.0, 40)]
    only = set(args.only.split(",")) - {""}
    for name, res, alpha, warm in small:
        if not only or name in only:
            small_fixture(models, name, res, alpha, warm)
    if not only or "checkpoint" in only:
        checkpoint_fixtures(models)
    if not only or "epochs" in only:
        epoch_fixture(models)
    if not only or "sampling" in only:
        sampling_fixture(models)
    for name in (["C1", "C2"] + (["C3", "C4", "C5"] if args.full else [])):
        if not only or name in only:
            full_fixture(models, config, name)


if __name__ == "__main__":
    main()
