"""World-size-2 data-parallel checks on the CPU (gloo): the flat-gradient exchange of the step driver, and the
data-parallel equivalence the design relies on: 2 ranks x batch b with summed gradients scaled by 1/2
== 1 rank x batch 2b.  The per-rank gradients come from the CPU oracle; the exchange is the product's code."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fixture):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package
        from conftest import load_golden, split_state
        from oracle import pggan_oracle as O
        pkg = load_package()
        torch.set_num_threads(2)
        fix = load_golden(fixture)
        res, alpha, init, latent, batch, lr = fix["meta"]
        batch = int(batch)
        D = pkg.models.Discriminator_PG([16, 16, 32], image_size_init=int(init))
        D.set_resolution(int(res), float(alpha))
        D.load_state_dict({k: torch.from_numpy(v) for k, v in split_state(fix, "D/").items()})
        flat = pkg.train.FlatParams(D)
        flat.set_active(pkg.train.active_parameters(D))

        # 1. the exchange is one SUM all-reduce over the whole flat buffer
        flat.grad.fill_(float(rank + 1))
        pkg.train.exchange_gradients(flat, world)
        assert torch.all(flat.grad == 3.0)

        # 2. data-parallel equivalence with oracle gradients
        spec = O.NetSpec(image_size_init=int(init), slope=0.2, alpha=float(alpha))
        pg = O.as_leaf_params(split_state(fix, "G/"))
        t = lambda k: torch.from_numpy(fix[k])

        def d_grads(sl):
            pd = O.as_leaf_params(split_state(fix, "D/"))
            loss, _, _ = O.d_w_loss(pg, spec, pd, spec, t("real")[sl], t("z_d")[sl], 0.001)
            gp = O.grad_penalty(pg, spec, pd, spec, t("real")[sl], t("z_gp")[sl], t("eps")[sl], 10.0)
            (loss + gp).backward()
            return {k: v.grad for k, v in pd.items() if v.grad is not None}

        half = batch // world
        mine = d_grads(slice(rank * half, (rank + 1) * half))
        flat.zero_grad()
        for name, p in zip(flat.names, flat.params):
            if name in mine:
                p.grad.add_(mine[name])          # what autograd's AccumulateGrad does into the flat views
        pkg.train.exchange_gradients(flat, world)
        full = d_grads(slice(0, batch))
        gmax = max(float(v.abs().max()) for v in full.values())
        for name, p in zip(flat.names, flat.params):
            if name in full:
                got = p.grad / world            # the 1/world factor the fused Adam applies (grad_scale)
                # (the last bias' gradient is a cancellation of -1 + 1: compare against the overall gradient scale too)
                err = float((got - full[name]).abs().max() / (full[name].abs().max() + 1e-2 * gmax))
                assert err < 1e-5, (name, err)
        # 3. generator stem: gathered rank-B factors give the same gradient as all-reducing the per-rank products
        torch.manual_seed(100 + rank)
        b, k, s2, c, scale = 3, 8, 4, 5, 0.25
        z, gc = torch.randn(b, k), torch.randn(b, 2, 2, c)
        w = torch.nn.Parameter(torch.zeros(c * s2, k))
        w.grad = torch.zeros_like(w)
        ref_fn = lambda zs, gs, out, n, kk, ss, cc, sc: out.copy_(sc * torch.einsum("bpc,bk->cpk", gs.reshape(n, ss, cc), zs).reshape(cc * ss, kk))
        ex = pkg.train.StemGradExchange(w, world, wgrad_fn=ref_fn)
        ex.sink(z, gc, w, s2, c, scale)
        ex.finish()
        mine = torch.zeros_like(w)
        ref_fn(z, gc, mine, b, k, s2, c, scale)
        dist.all_reduce(mine)
        assert torch.allclose(w.grad, mine, atol=1e-5)
        # inactive tensors stay exactly zero on every rank
        for a, p in zip(flat.active_host, flat.params):
            if not a:
                assert float(p.grad.abs().max()) == 0.0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fixture", ["small_res8_warm", "small_res16_fade_warm"])
def test_two_rank_gradient_exchange_equals_big_batch(fixture):
    mp.spawn(_worker, args=(2, _free_port(), fixture), nprocs=2, join=True)


def _replica_worker(rank, world, port, fixture):
    """Three iterations of PGGANTrainer.replay's SEGMENT SEQUENCE -- [critic forward/backward] -> exchange -> [critic Adam, generator
    forward/backward] -> exchange -> [generator Adam] -- over gloo with different data on each rank.  The three segments are stand-ins
    (the CPU oracle's gradients written into the trainer's flat .grad views; a torch restatement of the fused Adam over the flat
    buffers, grad_scale included) because the kernels need a GPU; everything between them is the product's code: replay(), the
    graph cache keyed by input shape, _exchange(), exchange_gradients(), the 1/world factor in the optimiser's hyper-parameters."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package
        from conftest import load_golden, split_state
        from oracle import pggan_oracle as O
        pkg = load_package()
        torch.set_num_threads(2)
        fix = load_golden(fixture)
        res, alpha, init, latent, batch, lr = fix["meta"]
        res, init, latent = int(res), int(init), int(latent)
        spec = O.NetSpec(image_size_init=init, slope=0.2, alpha=float(alpha))
        own = [dist.new_group([r]) for r in range(world)][rank]           # a one-rank group: the single-process reference
        b = 2                                                             # per-rank batch

        def build(group):
            G = pkg.models.Generator_PG([32, 16, 16], image_size_init=init, latent_dim=latent)
            D = pkg.models.Discriminator_PG([16, 16, 32], image_size_init=init)
            if res != init:
                G.set_resolution(res, float(alpha))
                D.set_resolution(res, float(alpha))
            G.load_state_dict({k: torch.from_numpy(v) for k, v in split_state(fix, "G/").items()})
            D.load_state_dict({k: torch.from_numpy(v) for k, v in split_state(fix, "D/").items()})
            tr = pkg.train.PGGANTrainer(G, D, learning_rate=1e-3, process_group=group, fused_stem=False)
            return G, D, tr

        def draws(it, r):           # what rank r sees in iteration it (every rank can rebuild every rank's batch)
            g = torch.Generator().manual_seed(1000 * it + r)
            z = [O.sample_latent_vec((b, latent), generator=g) for _ in range(3)]
            return torch.rand(b, 1, res, res, generator=g) * 2 - 1, z[0], z[1], torch.rand(b, 1, 1, 1, generator=g), z[2]

        def ref_adam(flat, opt):
            lr_, b1, b2, eps_, gscale = opt.hyper_host[:5]
            for i, (p, off, a) in enumerate(zip(flat.params, flat.offsets, flat.active_host)):
                if not a:
                    continue
                n = p.numel()
                flat.seg_step[i] += 1
                t = float(flat.seg_step[i])
                g = flat.grad[off:off + n] * gscale
                m, v, w = flat.exp_avg[off:off + n], flat.exp_avg_sq[off:off + n], flat.flat[off:off + n]
                m.lerp_(g, 1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                w.addcdiv_(m, (v.sqrt() / (1 - b2 ** t) ** 0.5).add_(eps_), value=-lr_ / (1 - b1 ** t))

        class Segment:                     # what a captured HIP graph is to replay(): something with .replay()
            def __init__(self, fn):
                self.fn = fn

            def replay(self):
                self.fn()

        def install(tr, G, D, batch_of, n_samples):
            """batch_of(): the (real, z_d, z_gp, eps, z_g) the segments compute on (set per iteration through `cur`)"""
            pg = dict(G.named_parameters())
            pd = dict(D.named_parameters())
            pd["alpha"] = D.alpha

            def seg_a():
                tr.flat_d.zero_grad()
                x, z1, z2, e, _ = batch_of()
                loss, _, _ = O.d_w_loss(pg, spec, pd, spec, x, z1, 0.001)
                gp = O.grad_penalty(pg, spec, pd, spec, x, z2, e, 10.0)
                tr.flat_g.zero_grad()          # (the oracle's detached generator passes leave nothing, its graph might)
                (loss + gp).backward(inputs=[p for p in D.parameters()])

            def seg_b():
                ref_adam(tr.flat_d, tr.opt_d)
                tr.flat_g.zero_grad()
                _, _, _, _, z3 = batch_of()
                O.g_w_loss(pg, spec, pd, spec, z3).backward(inputs=[p for p in G.parameters()])

            def seg_c():
                ref_adam(tr.flat_g, tr.opt_g)

            static_real = torch.zeros(n_samples, 1, res, res)
            tr._graphs[tuple(static_real.shape)] = ([Segment(seg_a), Segment(seg_b), Segment(seg_c)], static_real, {}, None, [], (False, False))

        cur = {}
        G, D, tr = build(None)
        assert tr.world == world and tr.opt_d.hyper_host[4] == 1.0 / world
        install(tr, G, D, lambda: cur["mine"], b)
        Gr, Dr, ref = build(own)
        assert ref.world == 1
        install(ref, Gr, Dr, lambda: cur["all"], world * b)
        start = [ref.flat_d.flat.clone(), ref.flat_g.flat.clone()]
        for it in range(3):
            per_rank = [draws(it, r) for r in range(world)]
            cur["mine"] = per_rank[rank]
            cur["all"] = tuple(torch.cat([d[i] for d in per_rank]) for i in range(5))
            tr.replay(cur["mine"][0])
            ref.replay(cur["all"][0])
        # replicas: bit-identical parameters and Adam state on both ranks after three updates ...
        for flat in (tr.flat_d, tr.flat_g):
            for buf in (flat.flat, flat.exp_avg, flat.exp_avg_sq, flat.seg_step):
                other = buf.clone()
                dist.broadcast(other, src=0)
                assert torch.equal(other, buf), "replicas diverged"
        # ... and equal to ONE rank training on the concatenated batches (N ranks x b == 1 rank x N b up to summation order)
        for flat, fr, was in ((tr.flat_d, ref.flat_d, start[0]), (tr.flat_g, ref.flat_g, start[1])):
            assert float((fr.flat - was).abs().max()) > 1e-3         # (three steps of lr 1e-3 did move the weights)
            assert float((flat.flat - fr.flat).abs().max()) < 2e-5, float((flat.flat - fr.flat).abs().max())
        assert [int(v) for v in tr.flat_d.seg_step.tolist()] == [3 * int(a) for a in tr.flat_d.active_host]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fixture", ["small_res16_fade_warm"])
def test_replayed_segments_with_exchanges_keep_two_replicas_identical(fixture):
    mp.spawn(_replica_worker, args=(2, _free_port(), fixture), nprocs=2, join=True)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` is the driver's multi-GPU command: with no launcher environment the script itself must start
    the N rank processes (before anything touches a GPU) and rank 0 must print one JSON line whose n_gpus is N and whose observed
    world size -- read back from the process group the ranks formed -- is N too.  `--launch-check` runs that path on the CPU (gloo)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_observed"] == 2 and out["sum_of_ones"] == 2.0
    # a rank count that disagrees with the launcher's world size is refused, not silently accepted
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"],
                         env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)


def test_bench_line_reports_every_rank_and_the_exchanges():
    """With N > 1 rank 0's JSON line carries each rank's own ms / iteration (min / max) and the time inside the two gradient
    exchanges, and every Winograd kernel instance is labelled with the share of its algorithmic flops it executes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rep = bench.rank_report([(7.61, 0.021, 0.093), (7.58, 0.020, 0.095), (7.90, 0.033, 0.101)])
    assert rep["ms_per_step"] == [7.61, 7.58, 7.9] and rep["ms_per_step_min"] == 7.58 and rep["ms_per_step_max"] == 7.90
    assert rep["exchange_ms_per_step"]["critic"] == [0.021, 0.02, 0.033] and len(rep["exchange_ms_per_step"]["generator"]) == 3
    json.dumps(rep)
    assert bench.is_winograd_instance("conv3x3_wino_kernel<2, 2, 8, 8, 1, 0>") and bench.is_winograd_instance("conv3x3_tile_kernel<1, 1, 1, 0, 2>")
    assert bench.is_winograd_instance("wgrad_f32_kernel<1, 1, 0, 32, 4, 1, 1>") and not bench.is_winograd_instance("wgrad_f32_kernel<2, 2, 0, 16, 8, 0, 0>")
    assert not bench.is_winograd_instance("conv3x3_mid_kernel<4, 2, 1, 2, 0, 0, 1, 0>") and not bench.is_split_bf16_instance("conv3x3_wino_kernel<2, 2, 8, 8, 1, 0>")


def test_bench_parent_never_imports_torch():
    """The launcher process must not initialise HIP: it does not even import torch (checked on the source: everything above
    `launch_ranks` is stdlib, torch is imported inside the rank-side functions only)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main()")]
    top_level_imports = [ln for ln in head.splitlines() if ln.startswith("import ") or ln.startswith("from ")]
    assert not any("torch" in ln or "__graft_entry__" in ln for ln in top_level_imports), top_level_imports
