"""Model-level parity of the HIP path against golden vectors captured from the reference (tests/golden/, made by
oracle/make_golden.py) -- reduced-width nets with every weight stored, and the full-width BASELINE configs with
weights rebuilt from the seed.  Tolerance (north star): losses, D(x), |grad D| within 1e-3 relative of the fp32 CPU
reference; parameter gradients within 2e-3 of their max-norm."""
import numpy as np
import pytest
import torch

from conftest import load_golden, split_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SMALL = ["small_fresh4", "small_res8_init", "small_res8_warm", "small_res16_fade_init", "small_res16_fade_warm",
         "small_res16_warm"]


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def build_small(ngan, fix):
    res, alpha, init, latent, batch, lr = fix["meta"]
    G = ngan.models.Generator_PG([32, 16, 16], image_size_init=int(init), latent_dim=int(latent))
    D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=int(init))
    if int(res) != int(init):
        G.set_resolution(int(res), float(alpha))
        D.set_resolution(int(res), float(alpha))
    G.load_state_dict({k: torch.from_numpy(v) for k, v in split_state(fix, "G/").items()})
    D.load_state_dict({k: torch.from_numpy(v) for k, v in split_state(fix, "D/").items()})
    assert abs(D.alpha_value() - float(alpha)) < 1e-6
    return G.to(DEV), D.to(DEV)


def run_step_losses(ngan, G, D, fix, lam=10.0, drift=0.001, lr=1e-4):
    """One iteration through the step driver (train.py:357-385): D step incl. fused Adam, then G step."""
    t = lambda k: torch.from_numpy(fix[k]).to(DEV)
    tr = ngan.train.PGGANTrainer(G, D, learning_rate=lr, beta1=0.5, grad_pen_lambda=lam, drift_epsilon=drift)
    sd = tr.d_step(t("real"), z_d=t("z_d"), z_gp=t("z_gp"), eps=t("eps"))
    fd = tr.flat_d
    cur_d = {id(p): n for n, p in D.named_parameters()}   # the fixtures use the state_dict names of the current stage
    cur_g = {id(p): n for n, p in G.named_parameters()}
    dgrads = {cur_d[id(p)]: p.grad.detach().cpu().numpy().copy() for p, a in zip(fd.params, fd.active_host) if a}
    norms = tr.gp_loss.last_grad_norms.cpu().numpy()
    sg = tr.g_step(t("real"), z=t("z_g"))
    tr.materialize_stem_grad()      # g_step applies Adam to the stem from the gradient's factors and stores no gradient for it
    fg = tr.flat_g
    ggrads = {cur_g[id(p)]: p.grad.detach().cpu().numpy().copy() for p, a in zip(fg.params, fg.active_host) if a}
    scal = np.array([float(sd["D_loss"]), float(sd["score_real"]), float(sd["score_fake"]), float(sd["D_grad_pen"]),
                     float(sg["G_loss"])])
    return scal, norms, dgrads, ggrads


def check_slices(fix, key, g, tol):
    """element-wise pins of the full-width fixtures: the first 96 entries and the largest entry of a gradient tensor, on the
    tensor's max-norm (a checksum can hide compensating errors; these cannot)"""
    if "sl/" + key not in fix:
        return
    flat = np.asarray(g, np.float64).reshape(-1)
    sl, mx = fix["sl/" + key].astype(np.float64), fix["mx/" + key]
    scale = float(mx[2])
    assert float(np.abs(flat[:sl.size] - sl).max()) < tol * scale, (key, "slice")
    assert abs(flat[int(mx[0])] - mx[1]) < tol * scale, (key, "largest entry")


def adam_close(got, want, lr):
    """first Adam step moves each weight by ~lr*sign(g): allow a few sign flips where |g| is at rounding level"""
    bad = np.mean(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)) > 0.1 * lr)
    return bad < 2e-3


@pytest.mark.parametrize("name", SMALL)
def test_small_nets_match_reference(ngan, name, conv_precision):
    fix = load_golden(name)
    G, D = build_small(ngan, fix)
    with torch.no_grad():
        img = G(torch.from_numpy(fix["z_d"]).to(DEV)).cpu().numpy()
        score = D(torch.from_numpy(fix["real"]).to(DEV)).cpu().numpy()
    assert img.shape == fix["G_of_z_d"].shape
    assert rel(img, fix["G_of_z_d"]) < 1e-4
    assert rel(score, fix["D_of_real"]) < 1e-3
    scal, norms, dgrads, ggrads = run_step_losses(ngan, G, D, fix)
    assert np.allclose(scal, fix["scalars"], rtol=1e-3, atol=2e-5), (scal, fix["scalars"])
    assert rel(norms, fix["grad_norms"]) < 1e-3
    want_d = split_state(fix, "Dgrad/")
    want_g = split_state(fix, "Ggrad/")
    assert set(dgrads) == set(want_d) and set(ggrads) == set(want_g)
    for k, v in want_d.items():
        # The critic's last bias gradient is sum_b go[b] with go = -1/B + 2*drift*D(x_b)/B for reals and +1/B for fakes: the
        # +-1/B terms cancel exactly in real arithmetic and leave ~2e-3 * mean score (~4e-5 at init), so the fp32 rounding of the
        # individual terms (2B * 2^-24 / B ~ 1e-7, in the reference's own summation order too) is a few 1e-3 of the result.
        # Its information content (the mean real score) is checked above in `scal`; here it gets that absolute slack.
        if v.size == 1 and abs(float(dgrads[k].ravel()[0]) - float(v.ravel()[0])) < 5e-7:
            continue
        assert rel(dgrads[k], v) < 2e-3, ("D", k, rel(dgrads[k], v))
    for k, v in want_g.items():
        assert rel(ggrads[k], v) < 2e-3, ("G", k, rel(ggrads[k], v))
    lr = float(fix["meta"][5])
    after_g, after_d = G.state_dict(), D.state_dict()
    for k, v in split_state(fix, "G_after/").items():
        assert adam_close(after_g[k].cpu().numpy(), v, lr), ("G_after", k)
    for k, v in split_state(fix, "D_after/").items():
        if k != "alpha":
            assert adam_close(after_d[k].cpu().numpy(), v, lr), ("D_after", k)


FULL = ["full_C1", "full_C2", "full_C3", "full_C4", "full_C5"]   # C5: BASELINE.json's fifth shape (512x512, batch 8) in the arithmetic this repo offers


@pytest.mark.parametrize("name", FULL)
def test_full_width_pins(ngan, name, conv_precision):
    """BASELINE.json configs C1..C4: weights from torch.manual_seed(1) (same constructors, same order as the
    reference), reals from seed 123, latents / epsilon from the fixture."""
    fix = load_golden(name)
    res, alpha, init, latent, batch, lr = fix["meta"]
    res, batch = int(res), int(batch)
    cfg = ngan.config
    torch.manual_seed(1)
    G = ngan.models.Generator_PG(cfg.N_gen_features, image_size_init=16)
    D = ngan.models.Discriminator_PG(cfg.N_dis_features, image_size_init=16)
    if res != 16:
        G.set_resolution(res, float(alpha))
        D.set_resolution(res, float(alpha))
    for k, v in G.state_dict().items():
        if v.numel() > 1:
            cs = fix["Ginit_cs/" + k]
            assert abs(float(v.double().abs().sum()) - cs[1]) <= 1e-9 * cs[1], k
    torch.manual_seed(123)
    x = torch.rand(batch, 1, res, res) * 2 - 1
    assert abs(float(x.double().sum()) - fix["real_cs"][0]) < 1e-6 * fix["real_cs"][1]
    G.to(DEV)
    D.to(DEV)
    fx = dict(fix)
    fx["real"] = x.numpy()
    with torch.no_grad():
        img = G(torch.from_numpy(fix["z_d"]).to(DEV))
        assert rel(img[:2, 0, :8, :8].cpu().numpy(), fix["G_of_z_d_slice"]) < 1e-3
        score = D(x.to(DEV)).cpu().numpy()
    assert rel(score, fix["D_of_real"]) < 1e-3
    # generator gradients before the critic moves (no Adam in between): tight
    G.zero_grad()
    g_pre, _ = ngan.loss_functions.G_W_loss(G, D)(x.to(DEV), z=torch.from_numpy(fix["z_g"]).to(DEV))
    g_pre.backward()
    assert abs(float(g_pre) - float(fix["G_loss_pre"])) < 1e-3 * abs(float(fix["G_loss_pre"])) + 2e-5
    for k, p in G.named_parameters():
        if p.grad is not None:
            cs = fix["cs/Ggrad_pre/" + k]
            got = float(p.grad.double().abs().sum())
            # sum|g| of a 16-entry tensor (ToImage) is a heavily cancelling sum over 2M pixels: single LeakyReLU ties resolved
            # differently move it by up to 2.3e-3 in split-bf16 mode while every operator agrees with fp64 to < 1e-5
            # (tools/first_block_sensitivity.py prints the four arithmetic variants); exact-fp32 mode stays below 1.2e-4
            assert abs(got - cs[1]) < (4e-3 if k.startswith("ToIm") else 2e-3) * cs[1], ("G pre", k, got, cs[1])
            # element-wise, of the tensor's max-norm: 3e-3 in exact fp32 (measured worst over C1-C5: 2.2e-3, layers.8.1.weight of C5;
            # init-state generator gradients are ~1e-5 and sums over 2M pixels whose LeakyReLU ties fall either way), 5e-3 in the
            # split-bf16 mode.  DESIGN.md section 2 quotes these two numbers.
            check_slices(fix, "Ggrad_pre/" + k, p.grad.cpu().numpy(), 3e-3 if conv_precision == "f32" else 5e-3)
    G.zero_grad()
    D.zero_grad()
    scal, norms, dgrads, ggrads = run_step_losses(ngan, G, D, fx)
    assert np.allclose(scal, fix["scalars"], rtol=1e-3, atol=2e-5), (scal, fix["scalars"])
    assert rel(norms, fix["grad_norms"]) < 1e-3
    for k, g in dgrads.items():
        cs = fix["cs/Dgrad/" + k]
        assert abs(float(np.abs(g.astype(np.float64)).sum()) - cs[1]) < 2e-3 * cs[1], ("D", k)
        # element-wise (tools/pin_report.py prints the worst tensor of every config and mode): exact fp32 stays within 1.7e-3 of the
        # tensor's max-norm; in split-bf16 mode the bias gradient of the critic's last 3x3 conv reaches 9.8e-3 (its 4e-6 forward
        # error is amplified by the cancellation in the PixelNorm backward of a nearly radial gradient), everything else 1.4e-3
        tol = 2e-3 if conv_precision == "f32" else (2e-2 if k.endswith(".bias") else 5e-3)
        check_slices(fix, "Dgrad/" + k, g, tol)
    # after the critic's first Adam step (each weight moves by ~lr*sign(g), so a rounding-level critic gradient can flip a
    # step): the generator gradients seen through the updated critic scatter by 1e-3 .. 1.1e-2 between arithmetic variants
    # whose critic gradients all agree with the reference to 1e-4 (tools/c2_sensitivity.py prints the four variants: exact
    # fp32 / split-bf16, fused / unfused backward).  2e-2 bounds that amplification; the tight statement about the
    # generator's gradients is the pre-update check above (3e-3 / 5e-3).
    for k, g in ggrads.items():
        cs = fix["cs/Ggrad/" + k]
        assert abs(float(np.abs(g.astype(np.float64)).sum()) - cs[1]) < 2e-2 * cs[1], ("G", k)


def check_slices_s2(fix, key, g, tol):
    """as check_slices, on the second iteration's pins (`s2/` keys)"""
    flat = np.asarray(g, np.float64).reshape(-1)
    sl, mx = fix["s2/sl/" + key].astype(np.float64), fix["s2/mx/" + key]
    scale = float(mx[2])
    worst = max(float(np.abs(flat[:sl.size] - sl).max()), abs(flat[int(mx[0])] - mx[1])) / scale
    return worst < tol, worst


@pytest.mark.parametrize("name", FULL)
def test_full_width_second_iteration(ngan, name, conv_precision):
    """What follows the first update (round-3 review: the only statement about it was the 2e-2 slack on the post-Adam generator
    checksums).  Two iterations of the step driver on the fixture's reals with the reference's draws (RNG order z_d, z_gp, eps, z_g per
    iteration, oracle/make_golden.py), then the SECOND iteration's scalars, |grad D| and critic gradients against the reference's.

    It does matter, and not through anything this repo does: the first Adam step moves every weight by ~lr * sign(g), so a gradient
    element at rounding level takes the other sign in another arithmetic.  Measured on the CPU oracle alone (tools/step2_sensitivity.py,
    profiles/r04_step2_sensitivity.txt): torch fp32 against torch fp64 -- the same algorithm -- are 3.8e-3 apart on the second
    iteration's |grad D| and 2.3e-2 on its critic gradient elements at C2 (the fixtures carry that fp64 run as `s2/f64/*`), and noise of
    2e-4 of a tensor's maximum on the first iteration's gradients (inside the north star's 1e-3) moves them by 1.1e-2 / 5.7e-2.  A
    1e-3 pin on the second iteration is therefore not a property any fp32 implementation has.  The bounds here are what the kernels
    measured with margin (worst over C1-C5: |grad D| 5.9e-3, critic gradient elements 5.2e-3, scalars 2.4e-3) and stay below that
    sensitivity: scalars 5e-3, |grad D| 1e-2, critic gradient elements 1e-2 of the tensor's max-norm, generator checksums 5e-2 in exact
    fp32 (measured 3.3e-2 on C4's layers.7.1.weight: the generator's second-iteration gradients come through a critic that has taken TWO
    sign-like steps; split-bf16, whose first-iteration gradients differ by up to 5e-3: 3e-2 / 8e-2 -- measured 2.3e-2 on C2's layers.4.weight,
    the very tensor on which torch fp32 and fp64 are 2.3e-2 apart); and the result must be no further from the fp64 answer than
    1.5e-2 on |grad D|."""
    fix = load_golden(name)
    res, alpha, init, latent, batch, lr = fix["meta"]
    res, batch = int(res), int(batch)
    cfg = ngan.config
    torch.manual_seed(1)
    G = ngan.models.Generator_PG(cfg.N_gen_features, image_size_init=16)
    D = ngan.models.Discriminator_PG(cfg.N_dis_features, image_size_init=16)
    if res != 16:
        G.set_resolution(res, float(alpha))
        D.set_resolution(res, float(alpha))
    torch.manual_seed(123)
    x = (torch.rand(batch, 1, res, res) * 2 - 1).to(DEV)
    G.to(DEV)
    D.to(DEV)
    t = lambda k: torch.from_numpy(fix[k]).to(DEV)
    tr = ngan.train.PGGANTrainer(G, D, learning_rate=float(lr), beta1=0.5, grad_pen_lambda=10.0, drift_epsilon=0.001)
    tr.train_iteration(x, t("z_d"), t("z_gp"), t("eps"), t("z_g"))
    sd = tr.d_step(x, z_d=t("s2/z_d"), z_gp=t("s2/z_gp"), eps=t("s2/eps"))
    cur_d = {id(p): n for n, p in D.named_parameters()}
    dgrads = {cur_d[id(p)]: p.grad.detach().cpu().numpy().copy() for p, a in zip(tr.flat_d.params, tr.flat_d.active_host) if a}
    norms = tr.gp_loss.last_grad_norms.cpu().numpy()
    sg = tr.g_step(x, z=t("s2/z_g"))
    tr.materialize_stem_grad()
    cur_g = {id(p): n for n, p in G.named_parameters()}
    ggrads = {cur_g[id(p)]: p.grad.detach().cpu().numpy().copy() for p, a in zip(tr.flat_g.params, tr.flat_g.active_host) if a}
    scal = np.array([float(sd["D_loss"]), float(sd["score_real"]), float(sd["score_fake"]), float(sd["D_grad_pen"]), float(sg["G_loss"])])
    want = fix["s2/scalars"]
    assert np.allclose(scal, want, rtol=5e-3, atol=5e-5), (scal, want)
    assert rel(norms, fix["s2/grad_norms"]) < 1e-2, rel(norms, fix["s2/grad_norms"])
    assert rel(norms, fix["s2/f64/grad_norms"]) < 1.5e-2, rel(norms, fix["s2/f64/grad_norms"])
    worst = {}
    for k, g in dgrads.items():
        ok, w = check_slices_s2(fix, "Dgrad/" + k, g, 1e-2 if conv_precision == "f32" else 3e-2)
        worst[k] = w
        assert ok, ("D step 2", k, w)
    for k, g in ggrads.items():
        cs = fix["s2/cs/Ggrad/" + k]
        assert abs(float(np.abs(g.astype(np.float64)).sum()) - cs[1]) < (5e-2 if conv_precision == "f32" else 8e-2) * cs[1], ("G step 2", k)


@pytest.mark.parametrize("n_colors,res,alpha,widths", [(3, 16, 0.5, None), (3, 16, 1.0, None), (1, 32, 1.0, None),
                                                       # widths that are not multiples of 16: the reference's presets 0004-0006 end in
                                                       # 8-channel blocks (configs/config.py:86-92); zero-padded contraction path
                                                       (1, 16, 0.5, ([32, 8], [8, 32])), (1, 32, 1.0, ([16, 8, 8], [8, 8, 16])),
                                                       # wide blocks shaped like the presets 0006-0008 (configs/config.py:90-98): more
                                                       # than 128 channels per conv run as output-channel chunks (ops._n_chunks)
                                                       (1, 32, 1.0, ([256, 128, 64], [64, 128, 256])), (1, 16, 0.5, ([256, 128], [128, 256])),
                                                       # 512 / 1024 channels (presets 0004-0008) and widths whose quarter is not a power of two:
                                                       # per-pixel operators through csrc/wide.hip
                                                       (1, 16, 1.0, ([1024, 512], [256, 512])), (1, 16, 0.5, ([96, 48], [48, 96]))])
def test_losses_and_gradients_match_oracle_on_the_fly(ngan, n_colors, res, alpha, widths, conv_precision):
    """Configurations the committed fixtures do not hold (RGB images: the reference's N_colors constructor argument; a 32x32 stable
    stage of a three-block net): one critic loss + gradient penalty + generator loss against the CPU oracle evaluated here on the
    same weights and draws.  The oracle itself is pinned by tests/test_oracle_golden.py."""
    from oracle import pggan_oracle as O
    torch.manual_seed(11 + n_colors + res)
    gw, dw = widths if widths is not None else (([32, 16], [16, 32]) if res == 16 else ([32, 16, 16], [16, 16, 32]))
    G = ngan.models.Generator_PG(gw, image_size_init=8, latent_dim=64, N_colors=n_colors)
    D = ngan.models.Discriminator_PG(dw, image_size_init=8, N_colors=n_colors)
    G.set_resolution(res, alpha)
    D.set_resolution(res, alpha)
    pg = O.as_leaf_params({k: v.detach().clone() for k, v in G.state_dict().items()})
    pd = O.as_leaf_params({k: v.detach().clone() for k, v in D.state_dict().items()})
    spec = O.NetSpec(image_size_init=8, slope=0.2, alpha=alpha)
    G.to(DEV)
    D.to(DEV)
    b = 4
    x = torch.rand(b, n_colors, res, res) * 2 - 1
    z1, z2, z3 = (O.sample_latent_vec((b, 64)) for _ in range(3))
    eps = torch.rand(b, 1, 1, 1)
    d_loss, s_r, s_f = O.d_w_loss(pg, spec, pd, spec, x, z1, 0.001)
    gp, norms = O.grad_penalty(pg, spec, pd, spec, x, z2, eps, 10.0, return_norms=True)
    (d_loss + gp).backward()
    LF = ngan.loss_functions
    Dl, Gp, Gl = LF.D_W_loss(G, D, 0.001), LF.D_grad_pen_loss(G, D, 10.0), LF.G_W_loss(G, D)
    xd = x.to(DEV)
    d2, sr2, sf2 = Dl(xd, z=z1.to(DEV))
    gp2 = Gp(xd, z=z2.to(DEV), epsilon=eps.to(DEV))
    (d2 + gp2).backward()
    got = np.array([float(d2.detach()), float(sr2.detach()), float(sf2.detach()), float(gp2.detach())])
    want = np.array([float(d_loss.detach()), float(s_r.detach()), float(s_f.detach()), float(gp.detach())])
    assert np.allclose(got, want, rtol=1e-3, atol=2e-5), (got, want)
    assert np.allclose(Gp.last_grad_norms.cpu().numpy(), norms.detach().numpy(), rtol=1e-3)
    # (the 256-channel nets: 5e-3 and one outlier per 256-element tensor -- measured 2.1e-3 on the critic's final 8x8 conv and 2.8e-3
    # with one outlier on the bias of its last 3x3 conv, whose init-state gradient is the PixelNorm backward of a nearly radial
    # gradient: a cancellation that amplifies a 1e-6 forward difference)
    wide = max(gw + dw) > 128
    tol_l2, tol_out = (5e-3, 5e-3) if wide else (2e-3, 1e-3)

    def close(got, ref, scale):
        # relative L2 error, plus a cap on the number of outliers: in nets this small ONE LeakyReLU tie that falls differently
        #  moves a single gradient element by several 1e-3 of the tensor's maximum
        d = (got.cpu().double() - ref.double())
        l2 = float(d.norm() / (ref.double().norm() + 1e-2 * scale))
        outliers = float((d.abs() > 1e-2 * (float(ref.abs().max()) + 1e-2 * scale)).double().mean())
        return l2 < tol_l2 and outliers <= tol_out, (l2, outliers)

    gmax = max(float(v.grad.abs().max()) for v in pd.values() if v.grad is not None)
    for k, p in D.named_parameters():
        if p.grad is not None:
            ok, info = close(p.grad, pd[k].grad, gmax)
            assert ok, ("D", k, info)
    g_ref = O.g_w_loss(pg, spec, pd, spec, z3)
    g_ref.backward()
    g2, _ = Gl(xd, z=z3.to(DEV))
    g2.backward()
    assert abs(float(g2.detach()) - float(g_ref.detach())) < 1e-3 * abs(float(g_ref.detach())) + 2e-5
    gmax = max(float(v.grad.abs().max()) for v in pg.values() if v.grad is not None)
    for k, p in G.named_parameters():
        if p.grad is not None:
            ok, info = close(p.grad, pg[k].grad, gmax)
            assert ok, ("G", k, info)


@pytest.mark.parametrize("name", ["small_res8_warm", "small_res16_fade_warm", "small_res16_warm"])
def test_reference_inner_loop_verbatim_over_the_drop_in_modules(ngan, name, conv_precision, monkeypatch):
    """INTEGRATION.md claims the reference's inner loop runs unchanged over these modules.  This is train.py:357-385 line for line
    -- zero_grad(), the loss modules called with the images ONLY, `D_loss_val += gp`, backward(), torch.optim.Adam.step() over
    net.parameters() -- with nothing of the step driver (no PGGANTrainer, no flat buffers, no fused Adam).  The latents and epsilon
    the loss modules draw are the fixture's, injected where the reference draws them: `sample_latent_vec` (looked up at call time,
    loss_functions.py:25, 63, 166) and `torch.rand` (loss_functions.py:170)."""
    fix = load_golden(name)
    Generator_net, Discriminator_net = build_small(ngan, fix)
    LF = ngan.loss_functions
    lr = float(fix["meta"][5])
    optimizer_gen = torch.optim.Adam(Generator_net.parameters(), lr=lr, betas=(0.5, 0.999))      # train.py:224-225
    optimizer_dis = torch.optim.Adam(Discriminator_net.parameters(), lr=lr, betas=(0.5, 0.999))
    G_loss = LF.G_W_loss(Generator_net, Discriminator_net)                                       # train.py:228-230
    D_loss = LF.D_W_loss(Generator_net, Discriminator_net, drift_epsilon=0.001)
    D_grad_loss = LF.D_grad_pen_loss(Generator_net, Discriminator_net, Lambda=10.0)
    draws = iter([fix["z_d"], fix["z_gp"], fix["z_g"]])                                          # call order of one iteration
    monkeypatch.setattr(LF, "sample_latent_vec", lambda size, *a, **k: torch.from_numpy(next(draws)).to(DEV))
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *a, **k: torch.from_numpy(fix["eps"]).to(DEV) if tuple(a[0]) == tuple(fix["eps"].shape) else real_rand(*a, **k))
    images = torch.from_numpy(fix["real"]).to(DEV)                                               # train.py:352-353
    # ---- train.py:356-385, n_critic = 1, sim_loss off ----
    Discriminator_net.zero_grad()
    D_loss_val, score_real, score_fake = D_loss(images)
    D_grad_pen_loss = D_grad_loss(images)
    D_loss_val += D_grad_pen_loss
    D_loss_val.backward()
    d_grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in Discriminator_net.named_parameters() if p.grad is not None}
    optimizer_dis.step()
    Generator_net.zero_grad()
    G_loss_val, z = G_loss(images)
    G_loss_val.backward()
    g_grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in Generator_net.named_parameters() if p.grad is not None}
    optimizer_gen.step()
    monkeypatch.undo()
    # ---- against the reference's numbers ----
    got = np.array([D_loss_val.item(), score_real.item(), score_fake.item(), D_grad_pen_loss.item(), G_loss_val.item()])
    assert np.allclose(got, fix["scalars"], rtol=1e-3, atol=2e-5), (got, fix["scalars"])
    assert rel(D_grad_loss.last_grad_norms.cpu().numpy(), fix["grad_norms"]) < 1e-3
    assert np.array_equal(z.cpu().numpy(), fix["z_g"])
    want_d, want_g = split_state(fix, "Dgrad/"), split_state(fix, "Ggrad/")
    assert set(d_grads) == set(want_d) and set(g_grads) == set(want_g)       # exactly the reference's set of .grad-not-None parameters
    scale_d = max(float(np.abs(v).max()) for v in want_d.values())
    for k, v in want_d.items():
        assert float(np.abs(d_grads[k] - v).max()) < 2e-3 * (float(np.abs(v).max()) + 1e-2 * scale_d), ("D", k)
    scale_g = max(float(np.abs(v).max()) for v in want_g.values())
    for k, v in want_g.items():
        assert float(np.abs(g_grads[k] - v).max()) < 2e-3 * (float(np.abs(v).max()) + 1e-2 * scale_g), ("G", k)
    for k, v in split_state(fix, "D_after/").items():
        if v.ndim:
            assert adam_close(Discriminator_net.state_dict()[k].cpu().numpy(), v, lr), ("D after Adam", k)
    for k, v in split_state(fix, "G_after/").items():
        if v.ndim:
            assert adam_close(Generator_net.state_dict()[k].cpu().numpy(), v, lr), ("G after Adam", k)


def test_seeded_samples_match_the_reference_generator(ngan):
    """utils.gen_samples / plot_gen_samples (reference utils.py:346-355, 568-601) on the HIP generator against
    tests/golden/sampling_small.npz (images the reference's Generator_PG made from the same seeded latents, eval mode, no_grad, and
    their nearest-neighbour enlargement to image_size_max).  Images within 1e-4 (tanh outputs)."""
    fix = load_golden("sampling_small")
    res, seed, n, size_max, init, latent = (int(v) for v in fix["meta"])
    G = ngan.models.Generator_PG([32, 16, 16], image_size_init=init, latent_dim=latent)
    G.set_resolution(res, 1.0)
    G.load_state_dict({k: torch.from_numpy(v) for k, v in split_state(fix, "G/").items()})
    G = G.to(DEV)
    ngan.utils.Latent_vecs_memo.clear()
    G.train(True)
    images, z = ngan.utils.gen_samples(G, N_images=n, seed=seed)
    assert np.array_equal(z.cpu().numpy(), fix["z"]) and not images.requires_grad
    assert np.abs(images.cpu().numpy() - fix["images"]).max() < 1e-4
    grid = ngan.utils.plot_gen_samples(G, N_images=n, seed=seed)
    assert G.training                                                        # the caller's mode is restored (utils.py:571-583)
    want = ngan.utils.make_image_grid(torch.from_numpy(fix["enlarged"]), nrow=int(np.round(np.sqrt(n))))
    assert grid.shape == want.shape and float((grid - want).abs().max()) < 2e-4
