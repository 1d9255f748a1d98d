"""The three-way split-bf16 ("bf16x6") form of the 16 -> 16 layer, an experiment in the DIAGNOSTIC build (conv3x3_tile_kernel PREC = 3,
ngan_diag_conv3x3_bf16x6): error against an fp64 convolution of the same operands and in-graph time, next to the exact-fp32 Winograd kernel
(precision code 4) and the two-way split (precision code 1) of the product path.
    make -C neuron-gan_amd/csrc diag ; NGAN_LIB_PATH=build/diag/libngan_hip_diag.so python tools/bf16x6_probe.py      (record: profiles/r04_bf16x6_probe.txt)"""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
C, ops = pkg._C, pkg.ops
lib = ctypes.CDLL(os.environ["NGAN_LIB_PATH"])
dev = "cuda:0"
P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)


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


def errors(y, ref):
    d = (y.double() - ref)
    return float(d.norm() / ref.norm()), float(d.abs().max())


for B, H, W, epi in [(2, 64, 64, 0), (4, 128, 256, 0), (16, 512, 512, 0), (16, 512, 512, 1), (32, 256, 256, 1)]:
    torch.manual_seed(B + H)
    x = torch.randn(B, H, W, 16, device=dev)
    w = torch.randn(16, 16, 3, 3, device=dev)
    scale = 0.0833
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double() * scale, padding=1).permute(0, 2, 3, 1).contiguous()
    if epi:
        a = F.leaky_relu(ref, 0.2)
        ref = a / torch.sqrt((a * a).mean(dim=3, keepdim=True) + 1e-8)
    y6 = torch.empty(B, H, W, 16, device=dev)
    rn = torch.empty(B, H, W, device=dev)
    packed6 = torch.empty(5 * 3 * 512 // 2, device=dev)

    def run6():
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)       # (inside a capture this is the capturing stream)
        st = lib.ngan_diag_conv3x3_bf16x6(P(x), P(w), None, P(y6), P(rn), P(packed6), B, H, W, ctypes.c_float(scale), epi, ctypes.c_float(0.2), ctypes.c_float(1e-8), stream)
        assert st == 0, st
    run6()
    torch.cuda.synchronize()
    line = f"B{B} {H}x{W} 16->16 epilogue {epi}:  bf16x6 rel L2 {errors(y6, ref)[0]:.2e} max {errors(y6, ref)[1]:.2e}, {graph_time(run6):7.1f} us"
    for name, want in (("winograd fp32", 4), ("bf16x3", 1)):
        prec = C.conv3x3_algorithm(B, H, W, 16, 16, 0, 0 if want == 4 else 1)
        packed = ops._packed(w, 0, scale, prec)
        y = torch.empty(B, H, W, 16, device=dev)

        def run():
            C.call("ngan_conv3x3_fwd_ex", x, packed, None, y, rn if epi else None, None, None, None, B, H, W, 16, 16, 0, epi, 0, 0.2, 1e-8, prec, 0)
        run()
        torch.cuda.synchronize()
        line += f" | {name} (code {prec}) rel L2 {errors(y, ref)[0]:.2e} max {errors(y, ref)[1]:.2e}, {graph_time(run):7.1f} us"
    yt = F.conv2d(x.permute(0, 3, 1, 2), w * scale, padding=1).permute(0, 2, 3, 1)
    if not epi:
        line += f" | torch fp32 conv2d rel L2 {errors(yt, ref)[0]:.2e} max {errors(yt, ref)[1]:.2e}"
    print(line, flush=True)
