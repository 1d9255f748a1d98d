"""One rank, RCCL group, force_exchange: wall time of each captured segment and each exchange of the data-parallel step, synchronised
one by one, beside the back-to-back step time and the one-graph step (what the segmentation itself costs).   python tools/segment_probe.py"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pkg = load_package()
x = (torch.rand(16, 1, 512, 512) * 2 - 1).to(dev)
def timed(f, n=1):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3 / n
for seg in (True, False):
    G, D = bench.build_nets(pkg, 512, 1.0, dev)
    tr = pkg.train.PGGANTrainer(G, D, device_latents=True)
    if seg:
        tr.force_exchange = True; tr.enable_stem_exchange()
    tr.capture(x, warmup=2)
    for _ in range(3): tr.replay(x)
    if seg:
        for it in range(3):
            t = [timed(lambda: tr._graph[0].replay()), timed(lambda: tr._exchange(tr.flat_d)), timed(lambda: tr._graph[1].replay()),
                 timed(lambda: tr._exchange(tr.flat_g)), timed(lambda: tr._graph[2].replay())]
            print("ms: graphA %.3f  exchD %.3f  graphB %.3f  exchG %.3f  graphC %.3f  sum %.3f" % (*t, sum(t)), flush=True)
    print("segmented" if seg else "one graph", "%.3f ms per step back to back" % timed(lambda: tr.replay(x), 20), flush=True)
    del tr, G, D
dist.destroy_process_group()
