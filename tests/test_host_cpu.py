"""CPU-only checks of the host side: the C-ABI library and its symbols, the module tree / state_dict keys / growth
state machine mirrored from the reference, initialisation parity from a seed, the config module, the latent sampler,
the LR schedule, and loud failure (no fallback) when tensors are not on the GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, split_state


@pytest.fixture(scope="session")
def built(ngan):
    if not os.path.exists(ngan._C.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return ngan


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "ngan.h")).read()
    declared = set(re.findall(r"\b(ngan_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 32
    lib = ctypes.CDLL(built._C.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/ngan.h but not exported"
    assert declared == set(built._C.exported_symbols()), "ctypes signature table and header disagree"
    assert built._C.version().startswith("ngan-hip")


def test_invalid_arguments_return_status_not_crash(built):
    lib = built._C.lib()
    # null pointers / bad shapes are rejected on the host before any launch (no GPU needed)
    assert lib.ngan_conv3x3_fwd(None, None, None, None, None, 1, 8, 8, 16, 16, 0, 0, 0, 0.2, 1e-8, 0, 0, None) < 0
    assert b"null" in lib.ngan_last_error()
    assert built._C.wgrad_workspace_bytes(1, 8, 8, 16, 16) > 0
    assert built._C.wgrad_workspace_bytes(1, 8, 8, 12, 16) == 0


def test_in_place_entry_points_validate_on_the_host(built):
    """the accumulate-into-.grad forms and the stem's Adam-epilogue launch reject bad arguments before anything is launched"""
    lib = built._C.lib()
    assert lib.ngan_channel_sum_acc(None, None, None, 16, 16, 1.0, 1, None) < 0
    assert lib.ngan_final_dot_dw_acc(None, None, None, None, 1, 16, 16, 1.0, 3, None) < 0
    assert lib.ngan_from_image_dw_acc(None, None, None, None, None, 1, 8, 8, 1, 16, 0, 3, None) < 0
    assert lib.ngan_linear_wgrad_adam(None, None, None, None, None, None, None, 9, 16, 512, 256, 128, 1.0, None) < 0
    assert b"null" in lib.ngan_last_error()
    # a contraction the MFMA form does not take (K must be a multiple of 16, at most 512): refused, not mis-launched
    import ctypes
    one = ctypes.c_void_p(16)        # any non-null address: the shape check comes before the launch
    assert lib.ngan_linear_wgrad_adam(one, one, one, one, one, one, one, 9, 16, 520, 256, 128, 1.0, None) < 0
    assert b"K=520" in lib.ngan_last_error()
    # a binding written against the 5-float `hyper` of round 2 is refused by the count, not left to read past its buffer
    assert lib.ngan_linear_wgrad_adam(one, one, one, one, one, one, one, 5, 16, 512, 256, 128, 1.0, None) == -1
    assert lib.ngan_adam_step(one, one, one, one, one, one, one, one, 1, one, one, 1, one, 5, None) == -1
    assert b"hyper holds 5 floats" in lib.ngan_last_error()


def test_wide_layers_are_cut_into_kernel_sized_chunks(ngan):
    """A conv launch takes 16 / 32 / 64 / 128 output channels (include/ngan.h); the reference's wide presets (configs/config.py:87-98)
    and any other multiple of 16 run as chunks of those sizes, largest first, covering every channel exactly once."""
    chunks = ngan.ops._n_chunks
    assert chunks(128) == [(0, 128)] and chunks(16) == [(0, 16)]
    assert chunks(256) == [(0, 128), (128, 128)] and chunks(1024) == [(i * 128, 128) for i in range(8)]
    assert chunks(48) == [(0, 32), (32, 16)] and chunks(240) == [(0, 128), (128, 64), (192, 32), (224, 16)]
    for n in range(16, 1040, 16):
        c = chunks(n)
        assert sum(k for _, k in c) == n and all(k in (16, 32, 64, 128) for _, k in c)
        assert [s0 for s0, _ in c] == [sum(k for _, k in c[:i]) for i in range(len(c))]
    with pytest.raises(RuntimeError):
        chunks(24)


def test_measurement_switches_need_the_diag_flag(ngan, monkeypatch):
    """The Python layer's A/B switches are honoured only with NGAN_DIAG=1: a stray environment variable must not change which
    kernels a run exercises (the kernel library reads no environment variable at all: csrc/conv3x3_internal.h)."""
    monkeypatch.delenv("NGAN_DIAG", raising=False)
    monkeypatch.setenv("NGAN_POOL_FIRST", "0")
    assert ngan.ops._diag_env("NGAN_POOL_FIRST", "1") == "1"
    monkeypatch.setenv("NGAN_DIAG", "1")
    assert ngan.ops._diag_env("NGAN_POOL_FIRST", "1") == "0"
    import glob
    for f in glob.glob(os.path.join(ROOT, "neuron-gan_amd", "csrc", "*")):
        if f.endswith((".hip", ".cpp", ".h")):
            src = open(f).read()
            assert "getenv" not in src or "conv3x3_internal.h" in f, f"{f} reads the environment"


def test_no_cpu_fallback(ngan):
    G = ngan.models.Generator_PG([32, 16], image_size_init=4, latent_dim=32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G(torch.randn(2, 32))
    D = ngan.models.Discriminator_PG([16, 32], image_size_init=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        D(torch.randn(2, 1, 4, 4))


@pytest.mark.parametrize("name", ["small_fresh4", "small_res8_init", "small_res16_fade_init", "small_res16_warm"])
def test_state_dict_keys_match_reference(ngan, name):
    fix = load_golden(name)
    res, alpha, init, latent, _, _ = fix["meta"]
    G = ngan.models.Generator_PG([32, 16, 16], image_size_init=int(init), latent_dim=int(latent))
    D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=int(init))
    if int(res) != int(init):
        G.set_resolution(int(res), float(alpha))
        D.set_resolution(int(res), float(alpha))
    ref_g, ref_d = split_state(fix, "G/"), split_state(fix, "D/")
    assert list(G.state_dict().keys()) == list(ref_g.keys())
    assert list(D.state_dict().keys()) == list(ref_d.keys())
    for k, v in G.state_dict().items():
        assert tuple(v.shape) == ref_g[k].shape, k
    for k, v in D.state_dict().items():
        assert tuple(v.shape) == ref_d[k].shape, k
    G.load_state_dict({k: torch.from_numpy(v) for k, v in ref_g.items()})
    D.load_state_dict({k: torch.from_numpy(v) for k, v in ref_d.items()})
    assert abs(D.alpha_value() - float(alpha)) < 1e-7


def test_full_width_keys_and_init_from_seed(ngan):
    """SURVEY.md 3.5 key lists and 8(a1) initialisation pins (torch.manual_seed(1), G then D)."""
    cfg = ngan.config
    torch.manual_seed(1)
    G = ngan.models.Generator_PG(cfg.N_gen_features, image_size_init=16)
    D = ngan.models.Discriminator_PG(cfg.N_dis_features, image_size_init=16)
    sg, sd = G.state_dict(), D.state_dict()
    assert abs(float(sg["layers.0.weight"].abs().sum()) - 820408.476) < 0.5
    assert abs(float(sg["layers.4.weight"].abs().sum()) - 4802.032) < 0.01
    assert abs(float(sd["layers.0.weight"].abs().sum()) - 4800.991) < 0.01
    assert abs(float(sd["layers.3.weight"].abs().sum()) - 200.369) < 0.001
    assert "alpha" in sd and "alpha" not in sg  # D's alpha is persistent, G's is not (models.py:292, 465)
    assert not any("bias" in k for k in sg)
    assert tuple(sg["ToIm.layers.0.weight"].shape) == (1, 128, 1, 1) and tuple(sd["FromIm.conv.weight"].shape) == (128, 1, 1, 1)
    fix = load_golden("full_C1")
    for k, v in sg.items():
        if v.numel() > 1:
            assert abs(float(v.double().abs().sum()) - fix["Ginit_cs/" + k][1]) < 1e-9 * fix["Ginit_cs/" + k][1], k
    G.set_resolution(256)
    D.set_resolution(256)
    assert list(G.state_dict().keys()) == ["layers.0.weight", "layers.4.weight"] + [f"layers.{i}.{j}.weight" for i in (7, 8, 9, 10) for j in (1, 4)] + \
        ["conv_block_list.0.1.weight", "conv_block_list.0.4.weight", "ToIm_list.0.layers.0.weight", "ToIm.layers.0.weight"]
    assert list(D.state_dict().keys())[:9] == ["alpha"] + [f"layers.{i}.{j}.weight" for i in (0, 1, 2, 3) for j in (1, 4)]
    assert G.saved_attrs == ['LeakyReLU_neg_slope', 'N_colors', 'N_features_per_layer', 'N_layers', 'N_layers_max', 'image_size',
                             'image_size_init', 'image_size_max', 'latent_dim', 'training', 'alpha']


def test_growth_state_machine(ngan):
    G = ngan.models.Generator_PG([32, 16, 16], image_size_init=4, latent_dim=32)
    D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=4)
    assert G.image_size_max == 16 and D.image_size_max == 16 and G.N_layers == 1
    with pytest.raises(AssertionError):
        G.set_resolution(12)
    with pytest.raises(AssertionError):
        G.set_resolution(32)
    G.increase_resolution()
    D.increase_resolution()
    assert G.alpha_value() == 0.0 and G.image_size == 8 and G.N_layers == 2 and len(G.conv_block_list) == 2
    with pytest.raises(AssertionError, match="previous transition"):
        G.increase_resolution()
    # fp32 accumulation of 1e-4 needs exactly 10 000 calls and ends at 1.0000535
    n = 0
    while G.alpha_value() < 1:
        G.advance_transition(0.0001)
        D.advance_transition(0.0001)
        n += 1
    assert n == 10000 and abs(G.alpha_value() - 1.0000535) < 1e-6
    assert float(G.alpha) == pytest.approx(G.alpha_value()) and float(D.alpha) == pytest.approx(D.alpha_value())
    assert len(G.conv_block_list) == 1 and len(G.ToIm_list) == 1 and len(G.layers) == 8
    assert len(D.conv_block_list) == 1 and len(D.FromIm_list) == 1 and isinstance(D.layers[0], ngan.models.Conv2d_scale_block)
    G.increase_resolution()
    n = 0
    while G.alpha_value() < 1:
        G.advance_transition(0.05)
        n += 1
    assert n == 20
    with pytest.raises(AssertionError):
        G.increase_resolution()  # already at the maximum size


def test_config_module(ngan, tmp_path):
    cfg = ngan.config
    assert cfg.latent_dim == 512 and cfg.N_gen_features == [128, 64, 32, 32, 16, 16] and cfg.grad_pen_lambda == 10
    assert cfg.beta1 == 0.5 and cfg.learning_rate == 1e-4 and cfg.drift_epsilon == 0.001 and cfg.n_critic == 1
    with pytest.raises(ValueError, match="not defined"):
        cfg.set_configs(no_such_option=1)
    user = tmp_path / "my_config.py"
    user.write_text("ID = '0042'\ntransit_period = 20\nalpha_step = 0.1\nN_epochs = 200\nbatch_size = 4\n")
    saved = {k: getattr(cfg, k) for k in cfg.configs_name}
    try:
        cfg.import_configs(str(user), {"seed": 9})
        assert cfg.transit_sch == [20, 40, 60, 80, 100] and cfg.seed == 9 and cfg.batch_size == 4
        bad = tmp_path / "bad.py"
        bad.write_text("bogus = 1\n")
        with pytest.raises(ValueError, match="not defined"):
            cfg.import_configs(str(bad))
        tight = tmp_path / "tight.py"
        tight.write_text("ID = '0043'\ntransit_period = 5\nalpha_step = 0.1\nN_epochs = 200\n")
        with pytest.raises(AssertionError, match="separated"):
            cfg.import_configs(str(tight))
    finally:
        for k, v in saved.items():
            setattr(cfg, k, v)


def test_latent_sampler(ngan):
    torch.manual_seed(7)
    z = ngan.utils.sample_latent_vec((4, 512))
    assert np.allclose(z[0, :3].numpy(), [-0.03830601, 0.01847874, 0.04198530], atol=1e-7)
    state = torch.get_rng_state()
    a = ngan.utils.sample_latent_vec((2, 8), seed=5)
    assert torch.equal(state, torch.get_rng_state()), "seeded draw must restore the global RNG state (utils.py:66, 84)"
    assert torch.equal(a, ngan.utils.sample_latent_vec((2, 8), seed=5))  # memoised
    u = ngan.utils.sample_latent_vec((3, 5), mode='rand')
    assert float(u.min()) >= -1 and float(u.max()) < 1
    with pytest.raises(ValueError):
        ngan.utils.sample_latent_vec((1, 2), mode='other')


def test_lr_schedule(ngan):
    f = ngan.train.lr_schedule
    sch, n = [100, 200], 300
    assert f(0, 1e-4, sch, n) == 1e-4 and f(100, 1e-4, sch, n) == 1e-4 and f(300, 1e-4, sch, n) == 1e-4
    gamma = np.exp(np.log(1 / 100) / 50)
    assert f(10, 1e-4, sch, n) == pytest.approx(1e-4 * gamma ** 10)
    assert f(50, 1e-4, sch, n) == pytest.approx(1e-6)
    assert f(51, 1e-4, sch, n) is None          # second half of a phase: the optimiser keeps its value (train.py:260)
    assert f(130, 1e-4, sch, n) == pytest.approx(1e-4 * gamma ** 30)


def test_flat_params_and_active_set(ngan):
    D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=4)
    D.set_resolution(8, 0.5)
    flat = ngan.train.FlatParams(D)
    assert flat.total % 64 == 0 and all(o % 64 == 0 for o in flat.offsets)
    for p, off in zip(flat.params, flat.offsets):
        assert p.data_ptr() == flat.flat.data_ptr() + 4 * off and p.grad.data_ptr() == flat.grad.data_ptr() + 4 * off
    act = {id(p) for p in ngan.train.active_parameters(D)}
    names = {n for n, p in D.named_parameters() if id(p) in act}   # current (stage-dependent) names
    assert names == {"layers.0.weight", "layers.0.bias", "layers.3.weight", "layers.3.bias", "FromIm.conv.weight", "FromIm.conv.bias",
                     "conv_block_list.1.1.weight", "conv_block_list.1.4.weight", "FromIm_list.1.conv.weight", "FromIm_list.1.conv.bias"}
    assert "conv_block_list.1.1.weight" in flat.names and "layers.0.weight" in flat.names   # construction-time names
    sd_before = {k: v.clone() for k, v in D.state_dict().items()}
    D.load_state_dict(sd_before)  # in-place copy keeps the views
    assert flat.params[0].data_ptr() == flat.flat.data_ptr()


def test_adaptive_critic_schedule_and_similarity_monitor():
    """utils.Calculate_D_steps / similarity_loss restate reference utils.py:105-120 and loss_functions.py:185-205"""
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    u = load_package().utils
    assert u.Calculate_D_steps([], [], 0, 5, 100) == 5                          # empty series: maximum
    assert u.Calculate_D_steps([1., 2., 3., 4.], [0., 0., 0., 0.], 0, 5, 100) == 2     # round(std 1.118 / gap 2.5 * 5)
    assert u.Calculate_D_steps([1., 1., 1.], [5., 5., 5.], 1, 5, 10) == 1       # constant real score: clamps to N_min
    assert u.Calculate_D_steps([0., 10.], [0.1, 9.9], 0, 5, 100) == 5           # small gap: clamps to N_max
    assert u.Calculate_D_steps([0.] * 50 + [1., 3.], [0.] * 50 + [1., 1.], 0, 4, 2) == 4   # only the last `Period` entries count
    torch.manual_seed(0)
    x, z = torch.randn(6, 1, 8, 8), torch.randn(6, 32)
    xm = (x.reshape(6, -1).double().numpy())
    zm = z.double().numpy()
    xm = xm / np.linalg.norm(xm, axis=1, keepdims=True)
    zm = zm / np.linalg.norm(zm, axis=1, keepdims=True)
    want = 0.7 * ((zm @ zm.T - xm @ xm.T) ** 2).sum() / 30
    assert abs(float(u.similarity_loss(x, z, 0.7)) - want) < 1e-6


def test_epoch_schedule_follows_the_reference_epoch_by_epoch(ngan):
    """tests/golden/epochs_small.npz was written by oracle/make_golden.py driving the REFERENCE's Generator_PG / Discriminator_PG
    through nine epochs in the order of train.py:312-333 (alpha advance, then growth at the transition epochs) with the learning
    rate of update_lr (train.py:250-265): alpha of both nets, image size, layer count, state_dict key lists and the rate each epoch
    trains with.  The product's host logic (PGGANTrainer.start_epoch + lr_schedule, what pggan_train calls) must reproduce every
    row -- on the CPU: no kernel is involved."""
    fix = load_golden("epochs_small")
    n_epochs, alpha_step, base_lr = int(fix["meta"][0]), float(fix["meta"][1]), float(fix["meta"][2])
    transit = [int(v) for v in fix["meta"][3:]]
    torch.manual_seed(5)
    G = ngan.models.Generator_PG([32, 16, 16], image_size_init=4, latent_dim=32)
    D = ngan.models.Discriminator_PG([16, 16, 32], image_size_init=4)
    tr = ngan.train.PGGANTrainer(G, D, learning_rate=base_lr, alpha_step=alpha_step)
    lr0 = ngan.train.lr_schedule(0, base_lr, transit, n_epochs)            # train.py:288-289
    if lr0 is not None:
        tr.opt_g.set_lr(lr0)
        tr.opt_d.set_lr(lr0)
    for row, gk, dk in zip(fix["rows"], fix["G_keys"], fix["D_keys"]):
        epoch = int(row[0])
        tr.start_epoch(epoch, transit)
        got = [epoch, G.alpha_value(), D.alpha_value(), G.image_size, D.image_size, G.N_layers, D.N_layers, tr.opt_g.param_groups[0]["lr"]]
        assert np.allclose(got[:7], row[:7], rtol=0, atol=1e-7), (got, row.tolist())
        assert abs(got[7] - row[7]) <= 1e-12 * row[7] + 1e-18, (epoch, got[7], row[7])
        assert tr.opt_d.param_groups[0]["lr"] == got[7]
        assert list(G.state_dict().keys()) == str(gk).split("|"), epoch
        assert list(D.state_dict().keys()) == str(dk).split("|"), epoch
        lr = ngan.train.lr_schedule(epoch, base_lr, transit, n_epochs)       # train.py:424-426, as pggan_train applies it
        if lr is not None:
            tr.opt_g.set_lr(lr)
            tr.opt_d.set_lr(lr)


def test_counter_rows_become_bytes_per_launch(tmp_path):
    """bench.py's live HBM-traffic figure: rocprofv3 writes one row per dispatch and XCD instance, in KiB; the kernel's launches are
    summed over the instances, averaged over the dispatches, and other kernels / counters are ignored."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    d = tmp_path / "run" / "host"
    d.mkdir(parents=True)
    full = "void (anonymous namespace)::wgrad_f32_kernel<1, 1, 0, 32, 4, 1, 1>((anonymous namespace)::WgradArgs)"
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
    for disp, per_xcd in ((7, 100.0), (9, 300.0)):
        rows += [f'{disp},"{full}",FETCH_SIZE,{per_xcd}' for _ in range(8)]
    rows += ['11,"void (anonymous namespace)::conv3x3_tile_kernel<1, 1, 0, 0, 2>(ngan::ConvArgs, int)",FETCH_SIZE,999.0',
             f'7,"{full}",WRITE_SIZE,5.0']
    (d / "123_counter_collection.csv").write_text("\n".join(rows) + "\n")
    dom = bench._kernel_short_name(full)
    assert dom == "wgrad_f32_kernel<1, 1, 0, 32, 4, 1, 1>"
    got = bench.counter_bytes_per_launch(str(tmp_path), "FETCH_SIZE", dom)
    assert got == ((8 * 100.0 + 8 * 300.0) / 2 * 1024.0, 2)
    assert bench.counter_bytes_per_launch(str(tmp_path), "WRITE_SIZE", dom) == (5.0 * 1024.0, 1)
    assert bench.counter_bytes_per_launch(str(tmp_path), "FETCH_SIZE", "no_such_kernel") is None
