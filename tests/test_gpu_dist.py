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


def _worker(rank, world, port, fixture, q):
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


@pytest.mark.parametrize("fixture", ["small_res16_fade_warm"])
def test_two_ranks_on_one_gpu_match_the_whole_batch(fixture):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fixture, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    results = sorted(q.get(timeout=5) for _ in range(2))
    assert results == [(0, "ok"), (1, "ok")], results
    assert all(p.exitcode == 0 for p in procs)
