"""GPU tests of the rows next to the hot path: checkpoint format, multi-iteration trajectories of the step
driver against the CPU oracle, and the epoch-level driver across growth transitions (graph re-capture, resume, samples)."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, split_state
from oracle import pggan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_reference_checkpoint_runs_on_gpu(ngan):
    exp = load_golden("ref_checkpoint_expected")
    G = ngan.models.Generator_PG([16, 16, 16], image_size_init=4).to(DEV)
    D = ngan.models.Discriminator_PG([16, 16, 16], image_size_init=4).to(DEV)
    ck = ngan.utils.Checkpointer(G, D, 1e-4, "unused.pth", verbose=False, device=torch.device(DEV))
    ck.load_state(os.path.join(GOLDEN, "ref_checkpoint_new.pth"))
    assert G.image_size == 8 and next(G.parameters()).is_cuda
    with torch.no_grad():
        img = G(torch.from_numpy(exp["z"]).to(DEV)).cpu().numpy()
        score = D(torch.from_numpy(exp["x"]).to(DEV)).cpu().numpy()
    assert rel(img, exp["G_of_z"]) < 1e-4 and rel(score, exp["D_of_x"]) < 1e-3


@pytest.mark.parametrize("name", ["small_res8_warm", "small_res16_fade_warm"])
def test_three_iterations_follow_the_oracle(ngan, name, conv_precision):
    """Same weights, same injected latents / epsilon / reals for 3 consecutive iterations: the fused-Adam trajectory (flat
    buffers, per-tensor step counts, packed-weight refresh) must track torch.optim.Adam on the oracle."""
    fix = load_golden(name)
    res, alpha, init, latent, batch, _ = fix["meta"]
    res, batch, latent = int(res), int(batch), int(latent)
    G = ngan.models.Generator_PG([32, 16, 16], image_size_init=int(init), latent_dim=latent)
    D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=int(init))
    G.set_resolution(res, float(alpha))
    D.set_resolution(res, float(alpha))
    G.load_state_dict({k: torch.from_numpy(v) for k, v in split_state(fix, "G/").items()})
    D.load_state_dict({k: torch.from_numpy(v) for k, v in split_state(fix, "D/").items()})
    pg, pd = O.as_leaf_params(split_state(fix, "G/")), O.as_leaf_params(split_state(fix, "D/"))
    spec = O.NetSpec(image_size_init=int(init), slope=0.2, alpha=float(alpha))
    lr = 1e-3
    og, od = O.make_adam(pg, lr), O.make_adam(pd, lr)
    tr = ngan.train.PGGANTrainer(G.to(DEV), D.to(DEV), learning_rate=lr)
    torch.manual_seed(99)
    for it in range(3):
        x = torch.rand(batch, 1, res, res) * 2 - 1
        z = [O.sample_latent_vec((batch, latent)) for _ in range(3)]
        eps = torch.rand(batch, 1, 1, 1)
        want = O.train_step(pg, spec, pd, spec, og, od, x, z[0], z[1], eps, z[2])
        got = tr.train_iteration(x.to(DEV), z_d=z[0].to(DEV), z_gp=z[1].to(DEV), eps=eps.to(DEV), z_g=z[2].to(DEV))
        for k_w, k_g in (("D_loss", "D_loss"), ("score_real", "score_real"), ("score_fake", "score_fake"), ("GP", "D_grad_pen"), ("G_loss", "G_loss")):
            assert abs(float(got[k_g]) - want[k_w]) < 2e-3 * abs(want[k_w]) + 2e-4, (it, k_w, float(got[k_g]), want[k_w])
    # after three steps of lr = 1e-3 the weights have moved by ~3e-3; the two trajectories stay together
    for k, p in G.named_parameters():
        if k in pg and pg[k].grad is not None:
            assert float((p.detach().cpu() - pg[k].detach()).abs().max()) < 2e-4, k
    for k, p in D.named_parameters():
        if k in pd and pd[k].grad is not None:
            assert float((p.detach().cpu() - pd[k].detach()).abs().max()) < 2e-4, k


def test_epoch_driver_grows_checkpoints_and_resumes(ngan, tmp_path):
    models, train, utils = ngan.models, ngan.train, ngan.utils
    cfg = types.SimpleNamespace(adapt_critic=False, sim_loss_lambda=0.0, n_critic=1, batch_size=4, transit_sch=[3, 6], N_epochs=9,
                                alpha_step=0.5, learning_rate=1e-3, checkpointing_period=4, ID="t001")
    torch.manual_seed(5)
    G = models.Generator_PG([32, 16, 16], image_size_init=4, latent_dim=32).to(DEV)
    D = models.Discriminator_PG([16, 16, 32], image_size_init=4).to(DEV)
    data = train.TensorImageDataset.synthetic(8, 16, device=DEV)
    tr = train.PGGANTrainer(G, D, learning_rate=cfg.learning_rate, alpha_step=cfg.alpha_step, device_latents=True)
    f = str(tmp_path / "GenDisc_t001.pth")
    ck = utils.Checkpointer(G, D, cfg.learning_rate, f, N_epochs=cfg.N_epochs, verbose=False, device=torch.device(DEV), trainer=tr,
                            extra_checkpoint_period=1e3)
    lines = []
    series = train.pggan_train(tr, data, cfg, checkpoint=ck, epoch_final=cfg.N_epochs + 1, log=lines.append, samples_dir=str(tmp_path))
    assert all(len(v) == 9 and np.isfinite(v).all() for v in series.values())
    assert G.image_size == 16 and D.image_size == 16 and G.alpha_value() >= 1 and D.alpha_value() >= 1
    assert [k for k in G.state_dict() if k.startswith("conv_block_list")] == [] and len(G.layers) == 9
    assert os.path.exists(f) and os.path.exists(str(tmp_path / "Samples_t001_8.png"))
    saved = ngan.utils.load_checkpoint_dict(f)
    assert saved["epoch"] == 8 and "optimizer_state" in saved and len(saved["Loss_real"]) == 8
    # the blocks that joined late have fewer Adam steps than the stem (torch skips .grad=None tensors)
    steps = dict(zip(saved["optimizer_state"]["D"]["names"], saved["optimizer_state"]["D"]["step"].tolist()))
    # (names are the construction-time ones: the critic's conv_block_list.1 joined at 8x8, conv_block_list.0 at 16x16)
    assert steps["layers.0.weight"] > steps["conv_block_list.1.1.weight"] > steps["conv_block_list.0.1.weight"] > 0
    # resume into fresh nets + trainer: same weights, same optimiser moments, and training continues
    G2 = models.Generator_PG([32, 16, 16], image_size_init=4, latent_dim=32).to(DEV)
    D2 = models.Discriminator_PG([16, 16, 32], image_size_init=4).to(DEV)
    tr2 = train.PGGANTrainer(G2, D2, learning_rate=cfg.learning_rate, alpha_step=cfg.alpha_step, device_latents=True)
    ck2 = utils.Checkpointer(G2, D2, cfg.learning_rate, f, N_epochs=cfg.N_epochs, verbose=False, device=torch.device(DEV), trainer=tr2)
    ck2.load_state()
    assert ck2.epoch == 8 and G2.image_size == 16
    for (k, a), b in zip(saved["Generator_state"].items(), G2.state_dict().values()):
        assert torch.equal(a, b.cpu()), k
    st2 = tr2.optimizer_state()
    assert torch.equal(st2["D"]["step"], saved["optimizer_state"]["D"]["step"])
    assert torch.equal(st2["G"]["exp_avg"]["layers.4.weight"], saved["optimizer_state"]["G"]["exp_avg"]["layers.4.weight"])
    more = train.pggan_train(tr2, data, cfg, checkpoint=ck2, epoch_init=9, epoch_final=10, log=lines.append)
    assert len(more["G_loss"]) == 1 and np.isfinite(more["G_loss"][0])
    grid = utils.plot_gen_samples(G2, N_images=4, seed=0)
    assert grid.shape[0] == 1 and grid.shape[1] > 2 * 16


def test_one_update_per_batch_with_a_ragged_last_batch(ngan):
    """10 images in batches of 4 -> batches of 4, 4, 2 per epoch.  The reference makes one optimiser step per batch
    (train.py:350-385).  The graph path must too: capture() trains nothing (its warm-up runs on a snapshot), graphs are kept per
    input shape, so three epochs cost two captures and exactly nine Adam steps."""
    models, train = ngan.models, ngan.train
    cfg = types.SimpleNamespace(adapt_critic=False, sim_loss_lambda=0.0, n_critic=1, batch_size=4, transit_sch=[], N_epochs=3,
                                alpha_step=0.5, learning_rate=1e-3, checkpointing_period=100, ID="t002")
    torch.manual_seed(6)
    G = models.Generator_PG([32, 16], image_size_init=8, latent_dim=32).to(DEV)
    D = models.Discriminator_PG([16, 32], image_size_init=8).to(DEV)
    data = train.TensorImageDataset.synthetic(10, 8, device=DEV)
    tr = train.PGGANTrainer(G, D, learning_rate=cfg.learning_rate, device_latents=True)
    captures = []
    real_capture = tr.capture
    tr.capture = lambda x, *a, **k: (captures.append(tuple(x.shape)), real_capture(x, *a, **k))[1]
    w0 = tr.flat_d.flat.clone()
    real_capture(data.full[:4])
    assert torch.equal(w0, tr.flat_d.flat) and float(tr.flat_d.seg_step.sum()) == 0.0        # capturing is not training
    tr._graphs.clear()
    series = train.pggan_train(tr, data, cfg, epoch_final=4, log=lambda *_: None)
    assert len(series["G_loss"]) == 3 and np.isfinite(series["G_loss"]).all()
    assert sorted(captures) == [(2, 1, 8, 8), (4, 1, 8, 8)], captures
    for flat in (tr.flat_g, tr.flat_d):
        steps = flat.seg_step.cpu().tolist()
        assert all(s == (9.0 if a else 0.0) for s, a in zip(steps, flat.active_host)), steps
    assert not torch.equal(w0, tr.flat_d.flat)


def test_cached_graphs_of_two_shapes_replay_the_eager_trajectory(ngan):
    """Graphs are cached per input shape.  Batch 16 and batch 8 at 64x64 choose DIFFERENT kernels and packed-weight formats for the
    16 -> 16 layers (ngan_conv3x3_algorithm: the persistent / Winograd forms need >= 256 tiles), so the second capture registers
    packed copies the first shape's re-pack table does not list.  Replaying A, B, A, B from the cache must stay bit-equal to eager
    `train_iteration` (round-2 advisor findings: stale packed weights, re-pack tables freed under a live graph)."""
    torch.manual_seed(11)
    def make():
        torch.manual_seed(11)
        G = ngan.models.Generator_PG([32, 16, 16, 16], image_size_init=8, latent_dim=32)
        D = ngan.models.Discriminator_PG([16, 16, 16, 32], image_size_init=8)
        G.set_resolution(64, 1.0)
        D.set_resolution(64, 1.0)
        return ngan.train.PGGANTrainer(G.to(DEV), D.to(DEV), learning_rate=1e-3)
    C = ngan._C
    assert C.conv3x3_algorithm(16, 64, 64, 16, 16, 0, 0) != C.conv3x3_algorithm(8, 64, 64, 16, 16, 0, 0)
    gen = torch.Generator().manual_seed(3)
    def draw(b):
        z = [torch.randn(b, 32, generator=gen) for _ in range(3)]
        z = [(v / v.norm(dim=1, keepdim=True)).to(DEV) for v in z]
        return dict(real=(torch.rand(b, 1, 64, 64, generator=gen) * 2 - 1).to(DEV), z_d=z[0], z_gp=z[1],
                    eps=torch.rand(b, 1, 1, 1, generator=gen).to(DEV), z_g=z[2])
    seq = [draw(16), draw(8), draw(16), draw(8)]
    eager, tr = make(), make()
    statics = {}
    for s in seq[:3]:
        if not tr.has_graph(s["real"].shape):
            st = statics.setdefault(s["real"].shape[0], {k: s[k].clone() for k in ("z_d", "z_gp", "eps", "z_g")})
            tr.capture(s["real"], draws=st)
    assert tr.has_graph(seq[0]["real"].shape) and tr.has_graph(seq[1]["real"].shape)
    for s in seq:
        eager.train_iteration(s["real"], s["z_d"], s["z_gp"], s["eps"], s["z_g"])
        for k, v in statics[s["real"].shape[0]].items():
            v.copy_(s[k])
        tr.replay(s["real"])
    torch.cuda.synchronize()
    for name, p, pe in zip(tr.flat_g.names + tr.flat_d.names, tr.flat_g.params + tr.flat_d.params, eager.flat_g.params + eager.flat_d.params):
        assert torch.equal(p, pe), f"{name}: {float((p - pe).abs().max())}"


def test_wide_nets_through_the_step_driver(ngan):
    """Blocks wider than 128 channels (the reference's presets 0004-0008) run as output-channel chunks with one-shot packed weight
    slices (ops._n_chunks): the whole step driver -- flat buffers, fused Adam, graph capture and replay -- must take them, and the
    replayed trajectory must equal the eager one bit for bit."""
    def make():
        torch.manual_seed(21)
        G = ngan.models.Generator_PG([256, 128], image_size_init=8, latent_dim=32)
        D = ngan.models.Discriminator_PG([128, 256], image_size_init=8)
        G.set_resolution(16, 0.5)
        D.set_resolution(16, 0.5)
        return ngan.train.PGGANTrainer(G.to(DEV), D.to(DEV), learning_rate=1e-3)
    gen = torch.Generator().manual_seed(5)
    def draw(b):
        z = [torch.randn(b, 32, generator=gen) for _ in range(3)]
        z = [(v / v.norm(dim=1, keepdim=True)).to(DEV) for v in z]
        return dict(real=(torch.rand(b, 1, 16, 16, generator=gen) * 2 - 1).to(DEV), z_d=z[0], z_gp=z[1],
                    eps=torch.rand(b, 1, 1, 1, generator=gen).to(DEV), z_g=z[2])
    seq = [draw(4) for _ in range(3)]
    eager, tr = make(), make()
    static = {k: seq[0][k].clone() for k in ("z_d", "z_gp", "eps", "z_g")}
    tr.capture(seq[0]["real"], draws=static)
    for s in seq:
        stats = eager.train_iteration(s["real"], s["z_d"], s["z_gp"], s["eps"], s["z_g"])
        for k, v in static.items():
            v.copy_(s[k])
        tr.replay(s["real"])
    torch.cuda.synchronize()
    assert all(np.isfinite(float(v)) for v in stats.values())
    for name, p, pe in zip(tr.flat_g.names + tr.flat_d.names, tr.flat_g.params + tr.flat_d.params, eager.flat_g.params + eager.flat_d.params):
        assert torch.equal(p, pe), f"{name}: {float((p - pe).abs().max())}"


def test_small_gradients_added_in_place_equal_autograds_accumulation(ngan, monkeypatch):
    """Biases and the FromImage / ToImage / head weights get their gradients added into .grad by the kernel that computes them
    (ngan_*_acc, ops._small_grads_in_place) instead of by autograd's AccumulateGrad.  Same contributions in the same order; the
    kernels' `grad + scale * sum` is one fused multiply-add where autograd rounds the product first, so the two trajectories agree
    to the last bits, not bit for bit (measured: 3e-6 on weights of size 0.3 after three Adam steps of 1e-3).  Fading stage, so the
    faded-in branches accumulate too."""
    def make():
        torch.manual_seed(31)
        G = ngan.models.Generator_PG([32, 16, 16], image_size_init=8, latent_dim=32)
        D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=8)
        G.set_resolution(32, 0.4)
        D.set_resolution(32, 0.4)
        return ngan.train.PGGANTrainer(G.to(DEV), D.to(DEV), learning_rate=1e-3)
    gen = torch.Generator().manual_seed(9)
    def draw(b):
        z = [torch.randn(b, 32, generator=gen) for _ in range(3)]
        z = [(v / v.norm(dim=1, keepdim=True)).to(DEV) for v in z]
        return ((torch.rand(b, 1, 32, 32, generator=gen) * 2 - 1).to(DEV), z[0], z[1], torch.rand(b, 1, 1, 1, generator=gen).to(DEV), z[2])
    seq = [draw(8) for _ in range(3)]
    assert ngan.ops._small_grads_in_place
    a, b = make(), make()
    for s in seq:
        a.train_iteration(*s)
    monkeypatch.setattr(ngan.ops, "_small_grads_in_place", False)
    for s in seq:
        b.train_iteration(*s)
    torch.cuda.synchronize()
    for name, p, pe in zip(a.flat_g.names + a.flat_d.names, a.flat_g.params + a.flat_d.params, b.flat_g.params + b.flat_d.params):
        # Adam normalises every element's step to ~lr, so an element whose gradient is ~0 can move by up to lr per step on a last-bit
        # difference: bound the worst element by a fraction of the 3 * lr the weights moved, and the tensor as a whole tightly
        assert float((p - pe).abs().max()) < 3e-4, f"{name}: {float((p - pe).abs().max())}"
        assert float((p - pe).norm()) <= 2e-5 * float(pe.norm()) + 1e-7, f"{name}: {float((p - pe).norm() / pe.norm())}"
    assert all(float(p.grad.abs().max()) > 0 for p in a.flat_d.params if p.numel() < 64 and a.flat_d.active_host[a.flat_d.index[id(p)]])


def test_flat_adam_against_torch_adam(ngan):
    """FusedAdam (ngan_adam_step over the flat buffers) next to torch.optim.Adam on copies of the same tensors: five steps of random
    gradients, one tensor without a gradient for the first two steps (torch skips `.grad is None` tensors, so its bias correction
    starts later -- the per-tensor step count), sizes that are not multiples of the 4096-element chunk.  Same fp32 formulas: the
    parameters agree to a few 1e-7 of a step's size."""
    torch.manual_seed(17)
    net = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(n, device=DEV) * 0.3) for n in (5000, 37, 4096, 12289)])
    ref = [torch.nn.Parameter(p.detach().clone()) for p in net]
    lr = 2e-3
    opt = torch.optim.Adam(ref, lr=lr, betas=(0.5, 0.999), eps=1e-8, foreach=False)
    flat = ngan.train.FlatParams(net)
    fa = ngan.train.FusedAdam(flat, lr, (0.5, 0.999))
    for it in range(5):
        late = it < 2                                      # tensor 1 joins at the third step
        flat.set_active([p for i, p in enumerate(flat.params) if not (late and i == 1)])
        flat.zero_grad()
        for i, (p, r) in enumerate(zip(flat.params, ref)):
            if late and i == 1:
                r.grad = None
                continue
            g = torch.randn_like(p) * (10.0 ** (i - 2))    # gradient scales from 1e-2 to 10
            p.grad.copy_(g)
            r.grad = g.clone()
        fa.step()
        opt.step()
    torch.cuda.synchronize()
    assert flat.seg_step.cpu().tolist() == [5.0, 3.0, 5.0, 5.0]
    for i, (p, r) in enumerate(zip(flat.params, ref)):
        d = float((p.detach() - r.detach()).abs().max())
        assert d < 1e-4 * lr, (i, d)                       # five steps of ~lr each; measured: 3e-5 lr (two ulps of the parameter) at most,
        assert float((p.detach() - r.detach()).abs().mean()) < 2e-6 * lr, i                            # 2.7e-7 lr on average


def test_penalty_switched_off_draws_no_second_latent_batch(ngan):
    """grad_pen_lambda = 0 is the reference CLI's argparse default: D_grad_pen_loss returns 0 (loss_functions.py:179) and draws
    nothing.  The step driver must run (round 1 stacked a CPU scalar with device tensors) and must not generate the unused fakes."""
    G = ngan.models.Generator_PG([32, 16], image_size_init=8, latent_dim=32).to(DEV)
    D = ngan.models.Discriminator_PG([16, 32], image_size_init=8).to(DEV)
    tr = ngan.train.PGGANTrainer(G, D, grad_pen_lambda=0.0, device_latents=True)
    seen = []
    inner = G.forward
    G.forward = lambda z: (seen.append(z.shape[0]), inner(z))[1]
    x = (torch.rand(4, 1, 8, 8) * 2 - 1).to(DEV)
    stats = tr.train_iteration(x)
    assert seen == [4, 4]                                   # one detached pass for the critic step, one pass in the generator step
    assert float(stats["D_grad_pen"]) == 0.0 and stats["D_grad_pen"].device.type == "cuda"
    cfg = types.SimpleNamespace(adapt_critic=False, sim_loss_lambda=0.0, n_critic=1, batch_size=4, transit_sch=[], N_epochs=1,
                                alpha_step=0.5, learning_rate=1e-3, checkpointing_period=100, ID="t003")
    G.forward = inner
    series = ngan.train.pggan_train(tr, ngan.train.TensorImageDataset.synthetic(4, 8, device=DEV), cfg, epoch_final=2, log=lambda *_: None)
    assert series["D_grad_pen"] == [0.0]


def test_epoch_driver_follows_the_reference_epoch_by_epoch(ngan):
    """tests/golden/epochs_small.npz (reference growth calls driven in train.py's order + update_lr): alpha, image size, layer
    count, learning rate and state_dict key lists of all nine epochs, observed INSIDE pggan_train through its on_epoch hook, graph
    replay on -- the epoch driver's growth / re-capture / LR logic against the reference, not against itself."""
    fix = load_golden("epochs_small")
    n_epochs, alpha_step, base_lr = int(fix["meta"][0]), float(fix["meta"][1]), float(fix["meta"][2])
    transit = [int(v) for v in fix["meta"][3:]]
    cfg = types.SimpleNamespace(adapt_critic=False, sim_loss_lambda=0.0, n_critic=1, batch_size=4, transit_sch=transit, N_epochs=n_epochs,
                                alpha_step=alpha_step, learning_rate=base_lr, checkpointing_period=100, ID="t004")
    torch.manual_seed(5)
    G = ngan.models.Generator_PG([32, 16, 16], image_size_init=4, latent_dim=32).to(DEV)
    D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=4).to(DEV)
    tr = ngan.train.PGGANTrainer(G, D, learning_rate=base_lr, alpha_step=alpha_step, device_latents=True)
    lr0 = ngan.train.lr_schedule(0, base_lr, transit, n_epochs)
    tr.opt_g.set_lr(lr0)
    tr.opt_d.set_lr(lr0)
    seen = []

    def on_epoch(epoch, trainer):
        seen.append(([epoch, G.alpha_value(), D.alpha_value(), G.image_size, D.image_size, G.N_layers, D.N_layers,
                      trainer.opt_g.param_groups[0]["lr"]], list(G.state_dict().keys()), list(D.state_dict().keys())))
    data = ngan.train.TensorImageDataset.synthetic(8, 16, device=DEV)
    series = ngan.train.pggan_train(tr, data, cfg, epoch_final=n_epochs + 1, log=lambda *_: None, on_epoch=on_epoch)
    assert len(seen) == n_epochs and np.isfinite(series["D_loss"]).all()
    for (row, gk, dk), want, wg, wd in zip(seen, fix["rows"], fix["G_keys"], fix["D_keys"]):
        assert np.allclose(row[:7], want[:7], rtol=0, atol=1e-7), (row, want.tolist())
        assert abs(row[7] - want[7]) <= 1e-12 * want[7], (row[0], row[7], want[7])
        assert gk == str(wg).split("|") and dk == str(wd).split("|"), row[0]


def test_stem_factor_exchange_matches_plain_gradient(ngan):
    """The data-parallel path forms the stem's weight gradient from (gathered) factors after the backward pass; with one rank
    that must give exactly the weights of the plain path.  So must the default path on the GPU, which hands the factors to Adam and
    never stores that gradient (PGGANTrainer.fused_stem)."""
    fix = load_golden("small_res8_warm")
    res, alpha, init, latent, batch, _ = fix["meta"]
    out = []
    for mode in ("plain", "factors", "adam_epilogue"):
        G = ngan.models.Generator_PG([32, 16, 16], image_size_init=int(init), latent_dim=int(latent))
        D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=int(init))
        G.set_resolution(int(res), float(alpha))
        D.set_resolution(int(res), float(alpha))
        G.load_state_dict({k: torch.from_numpy(v) for k, v in split_state(fix, "G/").items()})
        D.load_state_dict({k: torch.from_numpy(v) for k, v in split_state(fix, "D/").items()})
        tr = ngan.train.PGGANTrainer(G.to(DEV), D.to(DEV), learning_rate=1e-3, fused_stem=(mode == "adam_epilogue"))
        if mode == "factors":
            tr.enable_stem_exchange()
            assert tr.stem is not None
        assert tr.fused_stem == (mode == "adam_epilogue")
        t = lambda k: torch.from_numpy(fix[k]).to(DEV)
        for _ in range(3):      # three updates: the per-tensor step count and both moments take part
            tr.train_iteration(t("real"), z_d=t("z_d"), z_gp=t("z_gp"), eps=t("eps"), z_g=t("z_g"))
        if mode == "adam_epilogue":
            # the third form never stores the stem's gradient (ngan_linear_wgrad_adam: Adam in the epilogue of the factor product)
            assert float(tr.stem.weight.grad.abs().max()) == 0.0
        out.append({k: v.detach().cpu().clone() for k, v in G.state_dict().items()})
        out[-1]["exp_avg"], out[-1]["exp_avg_sq"] = tr.flat_g.exp_avg.cpu().clone(), tr.flat_g.exp_avg_sq.cpu().clone()
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k
        assert torch.equal(out[0][k], out[2][k]), k


def test_eager_iterations_do_not_leak_device_memory(ngan):
    """The first-order hand-off objects (ops.PNLink) are held by autograd nodes; they must not hold those nodes' own outputs (a
    cycle through C++ that gc cannot break leaked every iteration's activations once).  Allocated memory must be flat."""
    torch.manual_seed(2)
    G = ngan.models.Generator_PG([32, 16, 16], image_size_init=8, latent_dim=32)
    D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=8)
    G.set_resolution(32, 1.0)
    D.set_resolution(32, 1.0)
    tr = ngan.train.PGGANTrainer(G.to(DEV), D.to(DEV), device_latents=True)
    x = (torch.rand(4, 1, 32, 32) * 2 - 1).to(DEV)
    seen = []
    for i in range(8):
        tr.train_iteration(x)
        torch.cuda.synchronize()
        seen.append(torch.cuda.memory_allocated())
    assert seen[-1] <= seen[3], seen


def test_pooled_side_output_is_consumed_and_never_stale(ngan):
    """The 2x2-averaged copy a Winograd conv writes next to its output travels to the pooled conv of the next block as an attribute of
    the tensor object (ops._run_conv / ops._pooled_side).  (a) On the headline critic (512x512, fp32) it IS consumed: one feature-map
    pooling pass is left in a forward pass (behind the 64-channel block, whose producer is not a Winograd kernel); with the side output
    switched off there are four.  (b) A copy whose tensor was written after the producer stored it is not used."""
    ops, C = ngan.ops, ngan._C
    ops.set_conv_precision("f32")

    class Count:
        def __init__(self):
            self.n = 0

        def wants(self, name, args):
            if name == "ngan_pool2_fwd" and args[-1] >= 16:        # (the one-channel image pool of the first block is not a feature map)
                self.n += 1
            return False

        def add(self, *a):
            pass

    torch.manual_seed(1)
    D = ngan.models.Discriminator_PG(ngan.config.N_dis_features, image_size_init=16)
    D.set_resolution(512, 1.0)
    D.to(DEV)
    x = torch.rand(16, 1, 512, 512, device=DEV) * 2 - 1          # the headline batch: at 64x64 it still gives the 256 tiles a Winograd launch wants
    counts = []
    for allowed in (True, False):
        ops._pool_out_allowed = allowed
        probe = Count()
        C.set_probe(probe)
        try:
            with torch.no_grad(), ops.first_order_only():
                score = D(x)
        finally:
            C.set_probe(None)
            ops._pool_out_allowed = True
        counts.append((probe.n, score.clone()))
    assert counts[0][0] == 1 and counts[1][0] == 4, [c[0] for c in counts]
    assert torch.equal(counts[0][1], counts[1][1])                  # same bits either way
    # (b) staleness
    w = torch.randn(16, 16, 3, 3, device=DEV)
    xin = torch.randn(2, 128, 256, 16, device=DEV)
    y, _ = ops._run_conv(xin, w, None, 0, 0.1, 1, 0.2, pool_out=True)
    side = ops._pooled_side(y)
    assert side is not None and tuple(side.shape) == (2, 64, 128, 16)
    assert torch.equal(side, ops._pooled(y))
    y.mul_(2.0)
    assert ops._pooled_side(y) is None                             # written since: the copy is stale and a pooling pass runs instead
    assert ops._pooled_side(y.detach()) is None                    # another tensor object: no copy travels with it


def test_capture_survives_garbage_left_by_earlier_trainers(ngan):
    """Regression for the abort on record (gpurun_out/r03_a_tests.log: `Fatal Python error: Aborted` under "Garbage-collecting" during
    the third capture() of a process).  A dead trainer whose captured graphs sit in a reference cycle is left behind, the collector is
    set to run at every allocation, and a further capture() must neither abort nor be disturbed: capture() collects BEFORE the capture
    begins and keeps the cyclic collector off until it ends (PGGANTrainer.capture; which finalizer is the dangerous one:
    tools/gc_capture_probe.py, profiles/r04_gc_capture_probe.txt).  Run once; no repeat loop."""
    import gc
    fix = load_golden("small_res16_warm")
    import test_gpu_models as M
    real = torch.from_numpy(fix["real"]).to(DEV)

    def make():
        G, D = M.build_small(ngan, fix)
        return ngan.train.PGGANTrainer(G, D, device_latents=True)

    for _ in range(2):                       # two dead trainers, each holding its graphs in a cycle only the collector can free
        t = make()
        t.capture(real)
        t.replay(real)
        t.me = t
        del t
    old = gc.get_threshold()
    gc.set_threshold(1, 1, 1)
    try:
        t3 = make()
        t3.capture(real)                     # the third capture of the process
        assert gc.isenabled()                # ... and the collector is back on afterwards
        t3.replay(real)
        torch.cuda.synchronize()
    finally:
        gc.set_threshold(*old)
    gc.collect()
    assert all(torch.isfinite(p).all() for p in t3.G.parameters())
