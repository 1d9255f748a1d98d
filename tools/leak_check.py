import sys, gc, torch
sys.path.insert(0, ".")
from __graft_entry__ import load_package
import bench
pkg = load_package(); pkg.ops.set_conv_precision("bf16x3")
dev = torch.device("cuda:0")
G, D = bench.build_nets(pkg, 256, 1.0, dev)
tr = pkg.train.PGGANTrainer(G, D, device_latents=True)
x = (torch.rand(8, 1, 256, 256) * 2 - 1).to(dev)
for i in range(12):
    tr.train_iteration(x)
    torch.cuda.synchronize()
    if i % 3 == 2:
        print(f"iter {i}: allocated {torch.cuda.memory_allocated()/2**20:.0f} MiB, gc objects {len(gc.get_objects())}")
gc.collect(); torch.cuda.synchronize()
print(f"after gc.collect(): allocated {torch.cuda.memory_allocated()/2**20:.0f} MiB")
