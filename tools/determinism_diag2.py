import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
from conftest import load_golden
import test_gpu_models as T
ngan = load_package()
dev = torch.device("cuda:0")
fix = load_golden("small_res16_fade_warm")
t = lambda k: torch.from_numpy(fix[k]).to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
runs = []
for r in range(4):
    G, D = T.build_small(ngan, fix)
    tr = ngan.train.PGGANTrainer(G, D, learning_rate=1e-3, grad_pen_lambda=(0.0 if mode == "wloss" else 10.0))
    if mode == "gp":
        tr.flat_d.zero_grad()
        with torch.no_grad():
            fk = G(t("z_gp"))
        gp = tr.gp_loss(t("real"), x_tilde=fk, epsilon=t("eps"))
        with ngan.ops.deferred_wgrad():
            gp.backward()
    else:
        tr.d_compute(t("real"), t("z_d"), t("z_gp"), t("eps"))
    torch.cuda.synchronize()
    runs.append({n: p.grad.clone() for n, p in zip(tr.flat_d.names, tr.flat_d.params)})
for r in range(1, 4):
    bad = [(k, int((runs[0][k] != runs[r][k]).sum()), runs[0][k].numel(), float((runs[0][k] - runs[r][k]).abs().max())) for k in runs[0] if not torch.equal(runs[0][k], runs[r][k])]
    print(mode, f"run 0 vs {r}:", bad, flush=True)
