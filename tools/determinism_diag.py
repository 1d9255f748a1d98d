"""First-use vs later-use differences: run the same critic / generator half-steps on fresh trainers several times in one process
and report which gradient tensors differ between run 0 and run k (python tools/determinism_diag.py [bf16x3])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
from conftest import load_golden  # noqa: E402
import test_gpu_models as T  # noqa: E402

ngan = load_package()
if len(sys.argv) > 1:
    ngan.ops.set_conv_precision(sys.argv[1])
dev = torch.device("cuda:0")
fix = load_golden("small_res16_fade_warm")
t = lambda k: torch.from_numpy(fix[k]).to(dev)
runs = []
for r in range(4):
    G, D = T.build_small(ngan, fix)
    tr = ngan.train.PGGANTrainer(G, D, learning_rate=1e-3)
    out = {}
    st = tr.d_compute(t("real"), t("z_d"), t("z_gp"), t("eps"))
    out.update({"stat/" + k: v.clone() for k, v in st.items()})
    out.update({"dgrad/" + n: p.grad.clone() for n, p in zip(tr.flat_d.names, tr.flat_d.params)})
    tr.opt_d.step()
    out.update({"dparam/" + n: p.detach().clone() for n, p in zip(tr.flat_d.names, tr.flat_d.params)})
    st = tr.g_compute(t("real"), t("z_g"))
    out.update({"stat/" + k: v.clone() for k, v in st.items()})
    out.update({"ggrad/" + n: p.grad.clone() for n, p in zip(tr.flat_g.names, tr.flat_g.params)})
    tr.opt_g.step()
    out.update({"gparam/" + n: p.detach().clone() for n, p in zip(tr.flat_g.names, tr.flat_g.params)})
    torch.cuda.synchronize()
    runs.append(out)
for r in range(1, 4):
    bad = [(k, float((runs[0][k] - runs[r][k]).abs().max()), float(runs[0][k].abs().max())) for k in runs[0] if not torch.equal(runs[0][k], runs[r][k])]
    print(f"run 0 vs run {r}: {len(bad)} tensors differ" + "".join(f"\n    {k}: max diff {d:.3e} (max |v| {m:.3e})" for k, d, m in bad[:40]), flush=True)
bad = [k for k in runs[1] if not torch.equal(runs[1][k], runs[2][k])]
print("run 1 vs run 2:", len(bad), "tensors differ", bad[:10])
