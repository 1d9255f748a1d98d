"""Benchmark of the PGGAN / WGAN-GP training step on MI355X (contract: see the task brief / DESIGN.md).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full iteration of the reference's inner loop (train.py:357-385): D step (W loss + drift + gradient
penalty, backward, Adam) then G step (loss, backward, Adam), n_critic = 1, at the 512x512 final stage (alpha = 1),
batch 16 per GPU, fp32, synthetic reals 2*U[0,1)-1 already resident in HBM and unit-sphere latents drawn on the GPU.
Rank 0 prints ONE JSON line: images/s over all ranks, plus
  roofline     -- the dominant kernel (the conv template instance with the largest summed time over ngan_conv3x3_fwd and
                  ngan_conv3x3_fwd_ex) timed with HIP events on its launch stream: achieved = algorithmic bytes (split-bf16
                  instances: HBM roof, 8 TB/s) or flops (exact-fp32 instances: fp32 MFMA roof, 157.3 TF) of those launches / their
                  summed duration; traffic = HBM bytes per launch from the committed PMC summary (profiles/, tools/measure_round.sh)
  cpu_baseline -- the CPU oracle (oracle/pggan_oracle.py, a port of the reference's path) timed on this host's cores on a
                  bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

G_WIDTHS = [128, 64, 32, 32, 16, 16]
D_WIDTHS = [16, 16, 32, 32, 64, 128]
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak (about 6.3 TB/s is achievable with a float4 copy)


class ConvProbe:
    """HIP-event timing of ngan_conv3x3_fwd launches, bucketed by kernel template instance."""

    def __init__(self, name_of):
        self.records = []  # (key, flops, e0, e1)
        self.name_of = name_of

    @staticmethod
    def wants(name, args):
        return name in ("ngan_conv3x3_fwd", "ngan_conv3x3_fwd_ex")

    def add(self, name, args, e0, e1):
        o = 8 if name == "ngan_conv3x3_fwd_ex" else 5          # fwd_ex carries three more pointers (include/ngan.h)
        b, h, w, k, n, resample, epilogue, out_mode = args[o:o + 8]
        key = self.name_of(b, h, w, k, n, resample, epilogue, out_mode, args[o + 10])   # the template instance, as rocprofv3 names it
        # algorithmic work of one launch: 2*9*K*N flop per output pixel; bytes = the input read once
        # (K channels per source pixel; 1/4 of the pixels for bilinear input, 4x for pooled), the output written once
        # (4x the pixels for the pool-adjoint store; not at all when the ToImage epilogue runs without a stored activation), the
        # per-pixel norm when the epilogue produces it, the producer's output and norm read by the PixelNorm-backward epilogue,
        # the image written by the ToImage epilogue
        pix = b * h * w
        src = pix * (4 if resample == 1 else 0.25 if resample == 2 else 1)
        opix = pix * (4 if out_mode else 1)
        y_written = args[3] is not None
        nbytes = 4.0 * (src * k + (opix * n if y_written else 0))
        if epilogue == 1 or (epilogue == 3 and y_written):
            nbytes += 4.0 * pix
        if epilogue == 2:
            nbytes += 4.0 * (opix * n + opix)
        if epilogue == 3:
            nbytes += 4.0 * pix
        self.records.append((key, 2.0 * 9 * k * n * pix, nbytes, e0, e1))

    def summary(self):
        per = {}
        for key, flops, nbytes, e0, e1 in self.records:
            d = per.setdefault(key, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += flops
            d[2] += e0.elapsed_time(e1) * 1e-3
            d[3] += nbytes
        return {k: {"launches": v[0], "flops": v[1], "seconds": v[2], "avg_us": v[2] / v[0] * 1e6, "tflops": v[1] / v[2] / 1e12,
                    "gbs": v[3] / v[2] / 1e9} for k, v in per.items() if v[2] > 0}


def build_nets(pkg, res, alpha, device):
    torch.manual_seed(1)  # weights from torch.manual_seed(1)
    G = pkg.models.Generator_PG(G_WIDTHS, image_size_init=16)
    D = pkg.models.Discriminator_PG(D_WIDTHS, image_size_init=16)
    if res != 16:
        G.set_resolution(res, alpha)
        D.set_resolution(res, alpha)
    return G.to(device), D.to(device)


def cpu_baseline(res, alpha, sample_batch, budget_s=12.0):
    """Time the CPU oracle (port of the reference path) on this host; bounded sample of the same workload."""
    from oracle import pggan_oracle as O
    torch.manual_seed(1)
    pkg = load_package()
    G = pkg.models.Generator_PG(G_WIDTHS, image_size_init=16)
    D = pkg.models.Discriminator_PG(D_WIDTHS, image_size_init=16)
    if res != 16:
        G.set_resolution(res, alpha)
        D.set_resolution(res, alpha)
    pg = O.as_leaf_params({k: v.detach().clone() for k, v in G.state_dict().items()})
    pd = O.as_leaf_params({k: v.detach().clone() for k, v in D.state_dict().items()})
    spec = O.NetSpec(image_size_init=16, slope=0.2, alpha=alpha)
    og, od = O.make_adam(pg), O.make_adam(pd)
    # the box gives one GPU's job a 16-core share of the host; os.cpu_count() reports the whole machine
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    torch.manual_seed(123)
    done, t0 = 0, time.perf_counter()
    while True:
        x = torch.rand(sample_batch, 1, res, res) * 2 - 1
        z = [O.sample_latent_vec((sample_batch, 512)) for _ in range(3)]
        O.train_step(pg, spec, pd, spec, og, od, x, z[0], z[1], torch.rand(sample_batch, 1, 1, 1), z[2])
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s or done >= 8:
            break
    return {"value": sample_batch * done / el, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{done} full iteration(s) of the oracle at {res}x{res}, batch {sample_batch}, fp32, {cores} torch threads, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--graph", type=int, default=-1, help="1: replay a captured HIP graph, 0: eager, -1: auto")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3"],
                    help="conv arithmetic: exact fp32 MFMA, or split-bf16 (3 bf16 MFMAs per product, fp32 accumulate) where available")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the gradient all-reduce even with one rank (path test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="do not time the dominant kernel with HIP events")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # rehearsal on a one-GPU box: NGAN_REHEARSAL_BACKEND=gloo runs every rank on cuda:0 with the collectives staged through the
    # host (same step driver, same segmented graph capture; the numbers mean nothing).  The real run is RCCL, one rank per GPU.
    rehearsal = os.environ.get("NGAN_REHEARSAL_BACKEND", "")
    device = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(device)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)          # `python bench.py --force-dist` without a launcher: a one-rank group
        if rehearsal:
            dist.init_process_group(rehearsal)
        else:
            dist.init_process_group("nccl", device_id=device)

    pkg = load_package()
    pkg._C.lib()
    pkg.ops.set_conv_precision(args.precision)
    G, D = build_nets(pkg, args.res, args.alpha, device)
    trainer = pkg.train.PGGANTrainer(G, D, learning_rate=1e-4, beta1=0.5, grad_pen_lambda=10.0, drift_epsilon=0.001,
                                     device_latents=True)
    trainer.force_exchange = args.force_dist
    if args.force_dist:
        trainer.enable_stem_exchange()
    torch.manual_seed(123 + rank)
    pool = [(torch.rand(args.batch, 1, args.res, args.res) * 2 - 1).to(device) for _ in range(4)]
    torch.cuda.manual_seed(1000 + rank)

    # launch mode: HIP-graph replay of the whole iteration on one GPU (captured once), eager under torchrun
    use_graph = args.graph != 0   # one graph on a single GPU; three graphs with eager all-reduces between them under torchrun
    probe = None if args.no_probe else ConvProbe(pkg._C.conv3x3_kernel_name)

    def step(i):
        if use_graph:
            trainer.replay(pool[i % len(pool)])
        else:
            trainer.train_iteration(pool[i % len(pool)])

    if use_graph:
        try:
            trainer.capture(pool[0], warmup=max(1, args.warmup))
        except Exception as e:  # capture is an optimisation, never a requirement
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            use_graph = False
    for i in range(args.warmup):
        step(i)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    if probe is not None and not use_graph:
        pkg._C.set_probe(probe)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    pkg._C.set_probe(None)
    probe_note = "HIP events around each launch during the timed steps"
    if probe is not None and use_graph:
        # kernels inside a replayed graph cannot be bracketed by events: run the same K steps once more, eagerly,
        # with the probe on (same kernels, same shapes, same stream)
        pkg._C.set_probe(probe)
        for i in range(args.steps):
            trainer.train_iteration(pool[i % len(pool)])
        fence()
        pkg._C.set_probe(None)
        probe_note = "HIP events around each launch, eager re-run of the timed steps right after the graph-replayed timing"
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        from oracle import pggan_oracle as O
        images = args.batch * world * args.steps
        value = images / elapsed
        fg, fd = O.forward_flops(G_WIDTHS, D_WIDTHS, 16, args.res, 512, args.alpha)
        w_alg = 5 * fg + 14 * fd  # algorithmic flops per image per iteration
        out = {"metric": "images/sec (G+D step incl. GP) at 512x512", "value": value, "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.precision == "f32" else "f32 storage/accumulate; 3x3 convs on split-bf16 MFMA (bf16x3, hi+lo operands)",
               "data": "synthetic",
               "config": {"workload": f"{args.res}x{args.res} stage, alpha={args.alpha}, batch {args.batch}/GPU, WGAN-GP lambda=10, "
                                      f"drift 0.001, n_critic=1, Adam(1e-4, 0.5, 0.999), widths G{G_WIDTHS} D{D_WIDTHS}",
                          "global_batch": args.batch * world, "resolution": args.res, "parallelism": f"dp{world}",
                          "launch": ("hip-graph replay" + (" (3 segments, eager all-reduce between)" if use_dist else "")) if use_graph else "eager", "conv_precision": args.precision},
               "step_tflops": value * w_alg / 1e12, "step_frac_of_fp32_mfma_peak": value * w_alg / 1e12 / world / PEAK_FP32_MFMA_TFLOPS}
        if probe is not None and probe.records:
            summ = probe.summary()
            dom = max(summ, key=lambda k: summ[k]["seconds"])
            d = summ[dom]
            # HBM bytes per launch of that kernel from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3
            # passes over this same command; committed summary, see profiles/README.md).  null if no summary for this mode.
            traffic, traffic_src = None, None
            tfile = os.path.join(ROOT, "profiles", f"r01_e_traffic_{args.precision}.json")
            if os.path.exists(tfile) and args.res == 512 and args.batch == 16:
                with open(tfile) as fh:
                    tk = json.load(fh)["kernels"].get(dom)
                if tk:
                    traffic, traffic_src = tk["hbm_bytes_per_launch"], os.path.relpath(tfile, ROOT)
            if "mid_kernel" in dom or "up2f" in dom or ("persist" in dom and dom.rstrip(">").endswith(", 1")):   # split-bf16 instance: ~5x the fp32 MFMA rate, so HBM is the binding roof
                out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": d["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": d["gbs"] / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                                   "algorithmic_bytes_per_launch": d["gbs"] * 1e9 * d["avg_us"] * 1e-6, "launches": d["launches"],
                                   "avg_launch_us": d["avg_us"], "timing": probe_note,
                                   "fp32_equivalent_tflops": d["tflops"]}
            else:
                out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": d["tflops"], "peak": PEAK_FP32_MFMA_TFLOPS,
                                   "unit": "TFLOP/s", "frac": d["tflops"] / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                                   "traffic_source": traffic_src, "launches": d["launches"], "avg_launch_us": d["avg_us"], "timing": probe_note}
            tot_f = sum(v["flops"] for v in summ.values())
            tot_s = sum(v["seconds"] for v in summ.values())
            out["conv_family"] = {"tflops": tot_f / tot_s / 1e12, "seconds_per_step": tot_s / args.steps,
                                  "instances": {k: {"avg_us": round(v["avg_us"], 2), "tflops": round(v["tflops"], 2), "gbs": round(v["gbs"], 1),
                                                    "launches_per_step": v["launches"] / args.steps} for k, v in summ.items()}}
        else:
            out["roofline"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.res, args.alpha, sample_batch=min(args.batch, 4))
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
