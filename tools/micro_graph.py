"""Per-launch time of a conv shape INSIDE a replayed HIP graph (50 launches captured, one replay timed): small kernels take ~30 % less
than the eager / rocprofv3 figure, which includes the launch gap.   python tools/micro_graph.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); C, ops = pkg._C, pkg.ops
dev = "cuda:0"
def run(B,H,W,K,N):
    x = torch.randn(B,H,W,K, device=dev); w = torch.randn(N,K,3,3, device=dev)
    packed = ops._packed(w, 0, 0.1, 1); y = torch.empty(B,H,W,N, device=dev)
    f = lambda: C.call("ngan_conv3x3_fwd", x, packed, None, y, None, B,H,W,K,N,0,0,0,0.2,1e-8,1,0)
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): f()
    g.replay(); torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"B{B} {H}x{W} K{K} N{N} now={os.environ.get('NGAN_EXP_NOW','-')}: {e0.elapsed_time(e1)*1e3/50:.2f} us per launch inside a graph")
for cfg in [(16,16,16,128,128),(32,16,16,128,128),(16,32,32,64,64),(16,64,64,32,64)]: run(*cfg)
