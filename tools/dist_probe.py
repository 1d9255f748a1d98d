"""Two-rank rehearsal probe (gloo, both ranks on cuda:0): times the graph segments and the exchanges of the data-parallel step.
   torchrun --nproc-per-node 2 tools/dist_probe.py"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
import bench
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("gloo")
pkg = load_package(); pkg.ops.set_conv_precision("bf16x3")
G, D = bench.build_nets(pkg, 512, 1.0, dev)
tr = pkg.train.PGGANTrainer(G, D, device_latents=True)
x = (torch.rand(16, 1, 512, 512) * 2 - 1).to(dev)
tr.capture(x, warmup=2)
def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
for it in range(int(os.environ.get('PROBE_PHASES', '4'))):
    t = [timed(lambda: tr._graph[0].replay()), timed(lambda: tr._exchange(tr.flat_d)), timed(lambda: tr._graph[1].replay()),
         timed(lambda: tr._exchange(tr.flat_g)), timed(lambda: tr._graph[2].replay())]
    if rank == 0: print("ms: graphA %.2f  exchD %.2f  graphB %.2f  exchG %.2f  graphC %.2f" % tuple(t), flush=True)
pool = [(torch.rand(16, 1, 512, 512) * 2 - 1).to(dev) for _ in range(4)]
for name, fn in (("replay(pool[i%4])", lambda i: tr.replay(pool[i % 4])),):
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    for i in range(6): fn(i)
    torch.cuda.synchronize(); dist.barrier()
    if rank == 0: print("ms: %-24s %.2f per step" % (name, (time.perf_counter() - t0) / 6 * 1e3), flush=True)
dist.destroy_process_group()
