"""What would fusing the gradient penalty's first-pass PixelNorm backward into the input-gradient kernel save?  (round-3 review, item 4a)
A create_graph pass needs the input gradient BEFORE the PixelNorm backward as an operand of the second-order pass, so the fused kernel has
two outputs.  The DIAGNOSTIC build's 16 -> 16 Winograd tile kernel writes that second output when `aux_out` is given with epilogue 2; this
script times, inside a replayed HIP graph,
    A = input gradient (epilogue 0) + ngan_lrelu_pixelnorm_bwd as its own launch        (what the create_graph pass does today)
    B = input gradient with the PixelNorm-backward epilogue AND the pre-PixelNorm second output  (the fused form)
and checks that B's two outputs equal A's.
    make -C neuron-gan_amd/csrc diag ; NGAN_LIB_PATH=build/diag/libngan_hip_diag.so python tools/gp_fusion_probe.py      (record: profiles/r04_gp_fusion_probe.txt)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
C, ops = pkg._C, pkg.ops
dev = "cuda:0"


def graph_time(run, n=30):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


total = 0.0
for B, H, W in [(16, 256, 256), (16, 512, 512), (16, 128, 128)]:
    torch.manual_seed(B + H)
    K = N = 16
    g = torch.randn(B, H, W, K, device=dev)                       # gradient w.r.t. the conv's pre-activation
    w = torch.randn(K, N, 3, 3, device=dev)                       # (Cout, Cin): the input gradient contracts over Cout
    yprev = torch.randn(B, H, W, N, device=dev)                   # the producer's LeakyReLU -> PixelNorm output and norms
    rn = torch.rand(B, H, W, device=dev) + 0.5
    prec = C.conv3x3_algorithm(B, H, W, K, N, 0, 0)
    assert prec == 4, prec
    packed = ops._packed(w, 1, 0.1, prec)
    pre_a, post_a = torch.empty(B, H, W, N, device=dev), torch.empty(B, H, W, N, device=dev)
    pre_b, post_b = torch.empty(B, H, W, N, device=dev), torch.empty(B, H, W, N, device=dev)

    def run_a():
        C.call("ngan_conv3x3_fwd_ex", g, packed, None, pre_a, None, None, None, None, B, H, W, K, N, 0, 0, 0, 0.2, 1e-8, prec, 0)
        C.call("ngan_lrelu_pixelnorm_bwd", pre_a, None, yprev, rn, post_a, B * H * W, N, 0.2)

    def run_b():
        C.call("ngan_conv3x3_fwd_ex", g, packed, None, post_b, None, yprev, rn, pre_b, B, H, W, K, N, 0, 2, 0, 0.2, 1e-8, prec, 0)

    def run_c():      # the first-order form, for reference: fused epilogue, no second output
        C.call("ngan_conv3x3_fwd_ex", g, packed, None, post_b, None, yprev, rn, None, B, H, W, K, N, 0, 2, 0, 0.2, 1e-8, prec, 0)
    run_a(); run_b()
    torch.cuda.synchronize()
    assert torch.equal(pre_a, pre_b), float((pre_a - pre_b).abs().max())
    d = float((post_a - post_b).abs().max() / post_a.abs().max())
    assert d < 1e-6, d
    ta, tb, tc = graph_time(run_a), graph_time(run_b), graph_time(run_c)
    print(f"16 -> 16 input gradient, batch {B}, {H}x{W}: two launches {ta:7.1f} us | fused with both outputs {tb:7.1f} us | fused, one output {tc:7.1f} us "
          f"| saving {ta - tb:6.1f} us per layer ({(ta - tb) / ta * 100:4.1f} %); post-PixelNorm outputs agree to {d:.1e}", flush=True)
