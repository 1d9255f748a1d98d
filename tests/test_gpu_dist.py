"""Two data-parallel ranks on ONE GPU (gloo moves the collectives through the host; the real thing is RCCL, one rank per GPU):
the product's step driver -- HIP kernels, flat gradient buffers, critic all-reduce, generator stem exchanged as gathered rank-B
factors, tail all-reduce -- must give each rank (sum of per-rank gradients) == world x (gradient of the whole batch on one rank),
for the critic step and the generator step, on a fade-in stage.  SURVEY.md 8e."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fixture, q, precision="f32"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package
        from conftest import load_golden
        import test_gpu_models as T
        ngan = load_package()
        ngan.ops.set_conv_precision(precision)       # bf16: the stem factor that is all-gathered (the gradient w.r.t. the stem's output) is a bf16 tensor
        dev = torch.device("cuda:0")
        fix = load_golden(fixture)
        own = [dist.new_group([r]) for r in range(world)][rank]          # a one-rank group: the single-process reference
        t = lambda k: torch.from_numpy(fix[k]).to(dev)
        batch = int(fix["meta"][4])
        half = batch // world
        sl = slice(rank * half, (rank + 1) * half)

        G, D = T.build_small(ngan, fix)
        tr = ngan.train.PGGANTrainer(G, D)                                # world = 2: exchanges on
        assert tr.world == world and tr.stem is not None
        Gr, Dr = T.build_small(ngan, fix)
        ref = ngan.train.PGGANTrainer(Gr, Dr, process_group=own)          # world = 1 on the whole batch
        assert ref.world == 1

        def check(flat, flat_ref, what):
            worst = 0.0
            for name, p, pr, a in zip(flat.names, flat.params, flat_ref.params, flat.active_host):
                if not a:
                    assert float(p.grad.abs().max()) == 0.0, (what, name)
                    continue
                got, want = p.grad / world, pr.grad                      # the fused Adam applies the 1/world factor
                scale = float(want.abs().max()) + 1e-3 * max(float(q.grad.abs().max()) for q in flat_ref.params)
                worst = max(worst, float((got - want).abs().max()) / scale)
            assert worst < 2e-4, (what, worst)

        tr.d_compute(t("real")[sl], t("z_d")[sl], t("z_gp")[sl], t("eps")[sl])
        tr._exchange(tr.flat_d)
        ref.d_compute(t("real"), t("z_d"), t("z_gp"), t("eps"))
        check(tr.flat_d, ref.flat_d, "critic step")
        tr.g_compute(t("real")[sl], t("z_g")[sl])
        tr._exchange(tr.flat_g)
        ref.g_compute(t("real"), t("z_g"))
        check(tr.flat_g, ref.flat_g, "generator step")
        torch.cuda.synchronize()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("fixture", ["small_res16_fade_warm"])
def test_two_ranks_on_one_gpu_match_the_whole_batch(fixture, precision):
    """(bf16: every activation is rounded per element whatever the batch split, so two half batches still sum to the whole batch's gradient
    to fp32 summation order -- the same 2e-4)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fixture, q, precision)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    results = sorted(q.get(timeout=5) for _ in range(2))
    assert results == [(0, "ok"), (1, "ok")], results
    assert all(p.exitcode == 0 for p in procs)


def _capture_worker(port, fixture, q):
    """One rank, a real RCCL process group (watchdog thread alive), force_exchange: the three-segment capture, two replays and a
    re-capture must (a) train nothing while capturing, (b) reproduce the eager `train_iteration` trajectory bit for bit on the same
    reals / latents / epsilon, (c) never trip the watchdog's event poll (the round-1 abort, tools/capture_event_probe.py)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from __graft_entry__ import load_package
        from conftest import load_golden
        import test_gpu_models as T
        ngan = load_package()
        fix = load_golden(fixture)
        batch, latent = int(fix["meta"][4]), int(fix["meta"][3])
        res = int(fix["meta"][0])

        def make():
            G, D = T.build_small(ngan, fix)
            tr = ngan.train.PGGANTrainer(G, D, learning_rate=1e-3)
            tr.force_exchange = True
            tr.enable_stem_exchange()
            assert tr.stem is not None and tr._comm_stream is not None
            return tr

        gen = torch.Generator().manual_seed(77)
        steps = []
        for _ in range(3):
            z = [torch.randn(batch, latent, generator=gen) for _ in range(3)]
            z = [(v / v.norm(dim=1, keepdim=True)).to(dev) for v in z]
            steps.append(dict(real=(torch.rand(batch, 1, res, res, generator=gen) * 2 - 1).to(dev), z_d=z[0], z_gp=z[1],
                              eps=torch.rand(batch, 1, 1, 1, generator=gen).to(dev), z_g=z[2]))
        eager = make()
        for s in steps[:2]:
            eager.train_iteration(s["real"], s["z_d"], s["z_gp"], s["eps"], s["z_g"])
        tr = make()
        static = {k: steps[0][k].clone() for k in ("z_d", "z_gp", "eps", "z_g")}
        before = [t.clone() for t in tr._training_state()]
        tr.capture(steps[0]["real"], warmup=2, draws=static)
        assert len(tr._graph) == 3                                      # segmented: [D fwd/bwd] x [D Adam, G fwd/bwd] x [G Adam]
        for a, b in zip(before, tr._training_state()):
            assert torch.equal(a, b), "capture() must not train"
        for s in steps[:2]:
            for k, v in static.items():
                v.copy_(s[k])
            tr.replay(s["real"])
        torch.cuda.synchronize()
        for name, p, pe in zip(tr.flat_g.names + tr.flat_d.names, tr.flat_g.params + tr.flat_d.params, eager.flat_g.params + eager.flat_d.params):
            assert torch.equal(p, pe), f"{name}: replayed and eager trajectories differ by {float((p - pe).abs().max())}"
        for flat in (tr.flat_g, tr.flat_d):
            st = flat.seg_step.cpu()
            assert all(int(v) == (2 if a else 0) for v, a in zip(st, flat.active_host)), st   # one Adam step per batch, none from capturing
        # a growth event's re-capture directly behind replayed steps (their collectives may not have been polled by the watchdog yet)
        tr.refresh_stage()
        tr.capture(steps[2]["real"], draws=static)
        eager.train_iteration(steps[2]["real"], *(steps[2][k] for k in ("z_d", "z_gp", "eps", "z_g")))
        for k, v in static.items():
            v.copy_(steps[2][k])
        tr.replay(steps[2]["real"])
        torch.cuda.synchronize()
        import time
        time.sleep(0.5)                                                   # several watchdog polls
        assert all(torch.equal(p, pe) for p, pe in zip(tr.flat_d.params, eager.flat_d.params))
        # graphs cached for TWO input shapes (a ragged last batch), replayed A, B, A: each replay's eager stem exchange must read the
        # factor tensors ITS graphs fill, not those of the most recent capture (round-2 advisor finding: the stem is most of G)
        half = batch // 2
        cut = lambda s, n: {k: v[:n].clone() for k, v in s.items()}
        seq = [cut(steps[0], batch), cut(steps[1], half), cut(steps[2], batch)]
        statics = {batch: static}
        for s in (seq[0], seq[1], seq[0]):          # (a capture that registers new packed-weight copies drops the older graphs)
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
            assert torch.equal(p, pe), f"{name}: cached graphs of two shapes (A, B, A) left the eager trajectory by {float((p - pe).abs().max())}"
        q.put("ok")
    except Exception as e:  # noqa: BLE001
        q.put(repr(e))
        raise
    finally:
        dist.destroy_process_group()


def test_segmented_capture_with_a_live_rccl_group_matches_eager():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_capture_worker, args=(_free_port(), "small_res16_fade_warm", q))
    p.start()
    p.join(600)
    assert p.exitcode == 0, f"worker exit code {p.exitcode} (an abort here is the watchdog / capture interaction)"
    assert q.get(timeout=5) == "ok"
