"""Benchmark of the PGGAN / WGAN-GP training step on MI355X (contract: see the task brief / DESIGN.md section 5).

    python bench.py --gpus N --steps K --warmup W          # starts the N rank processes itself when N > 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full iteration of the reference's inner loop (train.py:357-385): D step (W loss + drift + gradient
penalty, backward, Adam) then G step (loss, backward, Adam), n_critic = 1, at the 512x512 final stage (alpha = 1),
batch 16 per GPU, synthetic reals 2*U[0,1)-1 already resident in HBM and unit-sphere latents drawn on the GPU.

The headline `value` is measured in the reference's arithmetic: fp32 storage, fp32 accumulation and fp32 products on
v_mfma_f32_16x16x4_f32 (the reference computes in the default dtype, train.py:136-144), `dtype: "f32"`.  The 16- and 32-channel
layers on large images run in Winograd F(2x2, 3x3) form -- still fp32 throughout, but 16 products per 2x2 output tile and channel
pair instead of 36, i.e. NOT the same products as the direct form; every such kernel instance is labelled in `conv_family` with
the fraction of its algorithmic flops it actually executes on the matrix pipe (`executed_mfma_frac`).  The faster split-bf16
convolution mode (3 bf16 MFMAs per fp32 product group, 16-bit-mantissa operands) is timed by the same protocol right afterwards
and reported as the labelled sub-record `"bf16x3"` of the same JSON line, with the relative error it shows against the fp32 mode
on identical weights and draws (`max_rel_err`).  It is never the headline.  A second sub-record, `"bf16"`, is BASELINE.json's C2
arithmetic at this workload: bf16 activation storage and conv operands (one bf16 MFMA per product group), fp32 accumulation /
statistics / master weights -- HBM-bound, so its roofline is bytes (E x 2) against 8 TB/s; it has its own, looser tolerance
(tests/test_gpu_bf16.py) and is never the headline either.

Reading `roofline.frac` of a Winograd instance: `achieved` divides the layer's ALGORITHMIC flops (2*9*K*N per output pixel, SURVEY.md
8d) by the launch time, and the F(2x2, 3x3) form executes only 4/9 of those multiply-adds on the matrix pipe -- so the fraction can
legitimately exceed 1.0 (it has, in isolation: 1.03 for 32 -> 32 at 128x128).  It says how the layer's WORK relates to the roof, not
how busy the pipe is.  For pipe occupancy read `executed_mfma_frac` (= frac x 4/9), and for the other roof `gbs / 8000`
(`conv_family.instances[*].hbm_frac`): a Winograd instance is "at its roof" when either of those two is near 1, not when `frac` is.

Rank 0 prints ONE JSON line: images/s over all ranks, plus
  roofline     -- the dominant kernel = the conv template instance (forward / input-gradient / weight-gradient entry points) with the
                  largest summed time, timed with HIP events on its launch stream: exact-fp32 instances against the fp32 MFMA
                  roof (157.3 TFLOP/s: achieved = 2*9*K*N*pixels of its launches / their summed duration), split-bf16 instances
                  against HBM (8 TB/s: achieved = algorithmic bytes / duration); traffic = HBM bytes per launch from the
                  PMC counters, collected INSIDE this invocation at N = 1 (two rocprofv3 passes over a child run of this script,
                  `live_traffic`; `traffic_source` says so) with the committed summary (profiles/, tools/measure_round.sh) as the
                  fallback when the profiler cannot run
  cpu_baseline -- the CPU oracle (oracle/pggan_oracle.py, a port of the reference's path) timed on this host's cores on a
                  bounded sample of the same workload at the same batch (N = 1 only): one untimed warm-up iteration, then >= 2 timed
                  ones (mean and per-iteration times in `sample` / `s_per_iteration`), with the CPU model string.

Multi-GPU: `python bench.py --gpus N` without a launcher environment starts N child processes (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set) BEFORE anything touches the GPU -- the parent never imports torch -- and exits with their worst exit code.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

G_WIDTHS = [128, 64, 32, 32, 16, 16]
D_WIDTHS = [16, 16, 32, 32, 64, 128]
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak (about 6.3 TB/s is achievable with a float4 copy)
PROFILE_TAG = "r04"            # profiles/<tag>_traffic_<precision>.json holds the PMC traffic per kernel


# ---------------------------------------------------------------------------------------------------------------------
# launcher: N rank processes from the bare command
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start n copies of this script, one per rank, and wait.  Nothing in this process has initialised HIP (torch is not even
    imported), so the children are ordinary fork+exec processes; rank 0's stdout is the JSON line."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, NGAN_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    worst = 0
    try:
        for p in procs:
            rc = p.wait()
            if rc != 0 and worst == 0:
                worst = rc
                for q in procs:             # one rank died: the others would wait for it in a collective
                    if q.poll() is None:
                        q.terminate()
    except KeyboardInterrupt:
        for q in procs:
            q.terminate()
        raise
    return worst


# ---------------------------------------------------------------------------------------------------------------------
# per-launch probe
# ---------------------------------------------------------------------------------------------------------------------
class ConvProbe:
    """HIP-event timing of the 3x3-conv entry points (forward / input gradient: ngan_conv3x3_fwd, _fwd_ex; weight gradient:
    ngan_conv3x3_wgrad), bucketed by the kernel template instance each call dispatches to."""

    def __init__(self, C):
        self.records = []  # (key, flops, bytes, e0, e1)
        self.C = C

    @staticmethod
    def wants(name, args):
        return name in ("ngan_conv3x3_fwd", "ngan_conv3x3_fwd_ex", "ngan_conv3x3_wgrad", "ngan_bf16_conv3x3_fwd", "ngan_bf16_conv3x3_wgrad")

    def add(self, name, args, e0, e1):
        bf = name.startswith("ngan_bf16_")       # bf16 activation storage: 2 bytes per activation element (norms and images stay 4)
        esz = 2.0 if bf else 4.0
        if name.endswith("conv3x3_wgrad"):
            # (x, g, gw, workspace, B, H, W, Cin, Cout, resample, scale, accumulate[, precision]): the contraction runs over pixels;
            # bytes = the input read once (1/4 of the pixels behind a bilinear x2, 4x behind an avg-pool) + the output gradient
            b, h, w, cin, cout, resample = args[4:10]
            key = self.C.conv3x3_wgrad_kernel_name(b, h, w, cin, cout, resample, 5 if bf else args[12])
            pix = b * h * w
            src = pix * (4 if resample == 1 else 0.25 if resample == 2 else 1)
            self.records.append((key, 2.0 * 9 * cin * cout * pix, esz * (src * cin + pix * cout), e0, e1))
            return
        o = 5 if name == "ngan_conv3x3_fwd" else 8            # the _ex / bf16 forms carry three more pointers (include/ngan.h)
        b, h, w, k, n, resample, epilogue, out_mode = args[o:o + 8]
        key = self.C.conv3x3_kernel_name(b, h, w, k, n, resample, epilogue, out_mode, 5 if bf else args[o + 10])   # the template instance, as rocprofv3 names it
        # algorithmic work of one launch: 2*9*K*N flop per output pixel; bytes = the input read once
        # (K channels per source pixel; 1/4 of the pixels for bilinear input, 4x for pooled), the output written once
        # (4x the pixels for the pool-adjoint store; not at all when the ToImage epilogue runs without a stored activation), the
        # per-pixel norm when the epilogue produces it, the producer's output and norm read by the PixelNorm-backward epilogue,
        # the image written by the ToImage epilogue
        pix = b * h * w
        src = pix * (4 if resample == 1 else 0.25 if resample == 2 else 1)
        opix = pix * (4 if out_mode else 1)
        y_written = args[3] is not None
        nbytes = esz * (src * k + (opix * n if y_written else 0))
        if epilogue == 1 or (epilogue == 3 and y_written):
            nbytes += 4.0 * pix
        if epilogue == 2:
            nbytes += esz * opix * n + 4.0 * opix
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
                    "gbs": v[3] / v[2] / 1e9, "bytes_per_launch": v[3] / v[0], "flops_per_launch": v[1] / v[0]}
                for k, v in per.items() if v[2] > 0}


def is_split_bf16_instance(name):
    """does this kernel template instance compute its products on the bf16 MFMA (HBM is then the binding roof)?"""
    if "up2f" in name or "wgrad_bf16x3" in name or "conv3x3_bf16_kernel" in name:
        return True
    # last template argument of these three: PREC
    return ("persist" in name or "tile_kernel" in name or "mid_kernel" in name) and name.rstrip(">").endswith(", 1")


def is_winograd_instance(name):
    """exact-fp32 kernels in Winograd F(2x2, 3x3) form: conv (last template argument PREC = 2) and weight gradient (WINO = 1)"""
    if "wgrad_f32_kernel" in name:
        return name.rstrip(">").endswith(", 1") and name.count(",") == 6
    if "conv3x3_wino_kernel" in name:
        return True
    return ("persist" in name or "tile_kernel" in name) and name.rstrip(">").endswith(", 2")


def rank_report(rows):
    """rows[r] = (ms per iteration on rank r's own clock, ms per iteration inside the critic exchange, ... the generator exchange):
    the part of rank 0's JSON line that makes a multi-GPU run readable -- who was slow, and how long the collectives took"""
    return {"ms_per_step": [round(r[0], 4) for r in rows], "ms_per_step_min": min(r[0] for r in rows), "ms_per_step_max": max(r[0] for r in rows),
            "exchange_ms_per_step": {"critic": [round(r[1], 4) for r in rows], "generator": [round(r[2], 4) for r in rows]},
            "exchange_timing": "HIP events on the communication stream around each gradient exchange (critic: one all-reduce; "
                               "generator: the stem's factor all-gathers + the tail all-reduce; the local stem weight gradient "
                               "formed from the gathered factors runs on the compute stream and is not in these numbers)"}


def build_nets(pkg, res, alpha, device):
    import torch
    torch.manual_seed(1)  # weights from torch.manual_seed(1)
    G = pkg.models.Generator_PG(G_WIDTHS, image_size_init=16)
    D = pkg.models.Discriminator_PG(D_WIDTHS, image_size_init=16)
    if res != 16:
        G.set_resolution(res, alpha)
        D.set_resolution(res, alpha)
    return G.to(device), D.to(device)


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(res, alpha, batch, budget_s=45.0, min_timed=2, max_timed=4):
    """Time the CPU oracle (port of the reference path, train.py:357-385) on this host at the bench's own batch: ONE untimed warm-up
    iteration (allocator growth, oneDNN primitive creation and weight re-ordering happen there), then at least `min_timed` timed
    iterations -- more while the next one still fits `budget_s` seconds of timed work.  Reports the mean and every iteration's time."""
    import torch
    from __graft_entry__ import load_package
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

    def one_iteration():
        x = torch.rand(batch, 1, res, res) * 2 - 1
        z = [O.sample_latent_vec((batch, 512)) for _ in range(3)]
        t0 = time.perf_counter()
        O.train_step(pg, spec, pd, spec, og, od, x, z[0], z[1], torch.rand(batch, 1, 1, 1), z[2])
        return time.perf_counter() - t0

    warm = one_iteration()
    times = []
    while len(times) < min_timed or (len(times) < max_timed and sum(times) + max(times) <= budget_s):
        times.append(one_iteration())
    el = sum(times)
    per = ", ".join(f"{t:.2f}" for t in times)
    return {"value": batch * len(times) / el, "unit": "images/s", "cores": cores, "cpu_model": cpu_model_name(), "kind": "port",
            "timed_iterations": len(times), "warmup_iterations": 1, "s_per_iteration": [round(t, 3) for t in times], "warmup_s": round(warm, 3),
            "sample": f"1 untimed warm-up iteration ({warm:.2f} s) + {len(times)} timed full iterations of the oracle at {res}x{res}, batch {batch}, "
                      f"fp32, {cores} torch threads: {per} s each, mean {el / len(times):.2f} s"}


# ---------------------------------------------------------------------------------------------------------------------
# HBM traffic of the dominant kernel, measured in THIS run: two rocprofv3 counter passes over a child run of this script
# ---------------------------------------------------------------------------------------------------------------------
def _kernel_short_name(full):
    """the probe's / tools/traffic_summary.py's spelling of a rocprofv3 Kernel_Name"""
    return full.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


def counter_bytes_per_launch(out_dir, counter, dom):
    """(mean bytes per launch, launches) of `counter` for kernel `dom` from the counter_collection.csv files rocprofv3 left under
    out_dir: one row per dispatch and XCD instance, values in KiB -- summed per dispatch, averaged over the dispatches."""
    import csv
    import glob
    by_dispatch = {}
    for f in glob.glob(out_dir + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter and _kernel_short_name(row["Kernel_Name"]) == dom:
                    by_dispatch[row["Dispatch_Id"]] = by_dispatch.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    if not by_dispatch:
        return None
    return sum(by_dispatch.values()) / len(by_dispatch) * 1024.0, len(by_dispatch)


def live_traffic(dom, precision, args, budget_s=90):
    """(bytes per launch, source note) for kernel `dom`, or (None, reason).  FETCH_SIZE and WRITE_SIZE do not fit one pass: each gets
    its own `rocprofv3 --pmc <counter> -- python3 bench.py --steps 1 --warmup 1 ...` child (eager launches of the same workload,
    counters only for kernels of `dom`'s family, nothing else traced), run from /tmp.  Units and the gfx950 correction as in
    MI355X_MICROARCH.md (HBM / rocprofv3): both counters in KiB, summed over the 8 XCD instances of a dispatch; FETCH_SIZE x 2
    (a wide streaming read is tallied at half its bytes); WRITE_SIZE exact."""
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    family = re.escape(dom.split("<")[0])
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--res", str(args.res), "--batch", str(args.batch),
             "--alpha", str(args.alpha), "--precision", precision, "--graph", "0", "--sub-record", "0", "--no-cpu-baseline", "--no-probe",
             "--live-traffic", "0"]
    per_launch = {}
    t_end = time.perf_counter() + budget_s
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        left = t_end - time.perf_counter()
        if left < 20:
            return None, "time budget of the counter passes exhausted"
        d = tempfile.mkdtemp(prefix="ngan_pmc_", dir="/tmp")
        try:
            # a session of its own, so that a pass that overruns its budget is ended as a whole (profiler AND profiled child)
            proc = subprocess.Popen([exe, "--pmc", counter, "--kernel-include-regex", family, "--output-format", "csv", "-d", d, "--"] + child,
                                    cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                    start_new_session=True)
            try:
                _, err = proc.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)      # exactly the process group started above
                except ProcessLookupError:
                    pass
                proc.communicate()
                return None, f"rocprofv3 --pmc {counter} pass exceeded its time budget"
            if proc.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited with {proc.returncode}: {err.decode(errors='replace')[-200:]}"
            got = counter_bytes_per_launch(d, counter, dom)
            if got is None:
                return None, f"no {counter} rows for {dom}"
            per_launch[counter] = got
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = 2.0 * per_launch["FETCH_SIZE"][0], per_launch["WRITE_SIZE"][0]
    return rd + wr, (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes, {per_launch['FETCH_SIZE'][1]} launches each) over a child "
                     f"run of this command in this invocation; read {rd / 1e6:.1f} MB (FETCH_SIZE x 2, gfx950) + written {wr / 1e6:.1f} MB")


# ---------------------------------------------------------------------------------------------------------------------
# one timed run in one arithmetic mode
# ---------------------------------------------------------------------------------------------------------------------
def run_mode(pkg, args, precision, device, world, rank, use_dist):
    import torch
    import torch.distributed as dist
    pkg.ops.set_conv_precision(precision)
    G, D = build_nets(pkg, args.res, args.alpha, device)
    trainer = pkg.train.PGGANTrainer(G, D, learning_rate=1e-4, beta1=0.5, grad_pen_lambda=10.0, drift_epsilon=0.001,
                                     device_latents=True)
    trainer.force_exchange = args.force_dist
    if args.force_dist:
        trainer.enable_stem_exchange()
    torch.manual_seed(123 + rank)
    pool = [(torch.rand(args.batch, 1, args.res, args.res) * 2 - 1).to(device) for _ in range(4)]
    torch.cuda.manual_seed(1000 + rank)

    # launch mode: HIP-graph replay of the whole iteration on one GPU (captured once); three graphs with the eager gradient
    # exchanges between them when there is a process group
    use_graph = args.graph != 0
    probe = None if args.no_probe else ConvProbe(pkg._C)

    def step(i):
        if use_graph:
            trainer.replay(pool[i % len(pool)])
        else:
            trainer.train_iteration(pool[i % len(pool)])

    if use_graph:
        try:
            trainer.capture(pool[0], warmup=max(1, min(args.warmup, 3)))
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
    if use_dist:
        trainer.comm_timing = []           # (tag, start event, end event) per gradient exchange, on the communication stream
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    pkg._C.set_probe(None)
    own_elapsed = elapsed
    exchange_ms = None
    if use_dist:
        per = {}
        for tag, e0, e1 in trainer.comm_timing:
            per.setdefault(tag, []).append(e0.elapsed_time(e1))
        exchange_ms = {tag: sum(v) / args.steps for tag, v in per.items()}      # ms per iteration spent inside each exchange
        trainer.comm_timing = None
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
    ranks = None
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what makes a first hardware scaling run readable: every rank's own clock and its time inside the two exchanges
        mine = torch.tensor([own_elapsed / args.steps * 1e3, exchange_ms.get("critic", 0.0), exchange_ms.get("generator", 0.0)],
                            device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ranks = rank_report([r.tolist() for r in allr])

    out = {"elapsed": elapsed, "ranks": ranks, "launch": ("hip-graph replay" + (" (3 segments, eager gradient exchange between)" if use_dist else ""))
           if use_graph else "eager", "roofline": None, "conv_family": None}
    if rank == 0 and probe is not None and probe.records:
        summ = probe.summary()
        dom = max(summ, key=lambda k: summ[k]["seconds"])
        d = summ[dom]
        # HBM bytes per launch of that kernel from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3
        # passes over this same command; committed summary, see profiles/README.md).  null if no summary for this mode.
        traffic, traffic_src = None, None
        live_note = None
        if args.live_traffic != 0 and world == 1 and not args.force_dist and precision == args.precision:
            try:
                traffic, traffic_src = live_traffic(dom, precision, args)
            except Exception as e:      # the counters are evidence, never a reason to lose the line
                traffic, traffic_src = None, f"{type(e).__name__}: {e}"
            if traffic is None:
                live_note, traffic_src = traffic_src, None
                print(f"[bench] live counter passes unavailable ({live_note}); falling back to the committed summary", file=sys.stderr)
        tfile = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_traffic_{precision}.json")
        if traffic is None and os.path.exists(tfile) and args.res == 512 and args.batch == 16:
            with open(tfile) as fh:
                tk = json.load(fh)["kernels"].get(dom)
            if tk:
                traffic, traffic_src = tk["hbm_bytes_per_launch"], os.path.relpath(tfile, ROOT)
        common = {"kernel": dom, "traffic": traffic, "traffic_source": traffic_src, "launches": d["launches"], "avg_launch_us": d["avg_us"],
                  "algorithmic_bytes_per_launch": d["bytes_per_launch"], "algorithmic_flops_per_launch": d["flops_per_launch"], "timing": probe_note}
        if is_split_bf16_instance(dom):   # ~5x the fp32 MFMA rate, so HBM is the binding roof
            out["roofline"] = dict({"bound": "hbm", "achieved": d["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": d["gbs"] / PEAK_HBM_GBS,
                                    "fp32_equivalent_tflops": d["tflops"]}, **common)
        else:
            out["roofline"] = dict({"bound": "mfma", "achieved": d["tflops"], "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                    "frac": d["tflops"] / PEAK_FP32_MFMA_TFLOPS, "gbs": d["gbs"]}, **common)
            if is_winograd_instance(dom):
                # `achieved` counts the layer's ALGORITHMIC flops (2 * 9 * K * N per pixel); the Winograd F(2x2, 3x3) form
                # executes 4/9 of those multiply-adds on the MFMA pipe (plus the transforms' additions on the VALU) -- said here so that
                # nobody reads the fraction as matrix-pipe occupancy
                out["roofline"].update({"algorithm": "Winograd F(2x2,3x3), fp32", "executed_mfma_tflops": d["tflops"] * 4.0 / 9.0,
                                        "executed_mfma_frac": d["tflops"] * 4.0 / 9.0 / PEAK_FP32_MFMA_TFLOPS})
        tot_f = sum(v["flops"] for v in summ.values())
        tot_s = sum(v["seconds"] for v in summ.values())

        def label(k, v):
            e = {"avg_us": round(v["avg_us"], 2), "tflops": round(v["tflops"], 2), "gbs": round(v["gbs"], 1),
                 "launches_per_step": v["launches"] / args.steps, "hbm_frac": round(v["gbs"] / PEAK_HBM_GBS, 3)}
            if not is_split_bf16_instance(k):
                e["mfma_frac"] = round(v["tflops"] / PEAK_FP32_MFMA_TFLOPS, 3)          # ALGORITHMIC flops against the fp32 MFMA peak
                if is_winograd_instance(k):                                             # ... of which the Winograd forms execute 4/9
                    e.update({"algorithm": "winograd_f2x2_3x3", "executed_mfma_frac": round(v["tflops"] * 4.0 / 9.0 / PEAK_FP32_MFMA_TFLOPS, 3)})
            return e
        out["conv_family"] = {"tflops": tot_f / tot_s / 1e12, "seconds_per_step": tot_s / args.steps,
                              # time-weighted over every probed conv launch (what "the dominant kernel" cannot say when the top
                              # instances are within 20 % of each other)
                              "frac_of_fp32_mfma_peak": (tot_f / tot_s / 1e12 / PEAK_FP32_MFMA_TFLOPS) if precision == "f32" else None,
                              "instances": {k: label(k, v) for k, v in summ.items()}}
    if args.force_dist and world == 1 and use_graph:
        # what the data-parallel step structure costs before any inter-GPU traffic: the same K steps as ONE captured graph, no exchange
        # (tools/segment_probe.py measured 3.3 % in round 3); printed next to the segmented time so that a first hardware scaling
        # run can separate "three graphs + two eager collectives" from the collectives' own time
        del trainer
        G2, D2 = build_nets(pkg, args.res, args.alpha, device)
        plain = pkg.train.PGGANTrainer(G2, D2, learning_rate=1e-4, beta1=0.5, grad_pen_lambda=10.0, drift_epsilon=0.001, device_latents=True)
        plain.capture(pool[0], warmup=1)
        for i in range(args.warmup):
            plain.replay(pool[i % len(pool)])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            plain.replay(pool[i % len(pool)])
        torch.cuda.synchronize()
        single = (time.perf_counter() - t1) / args.steps * 1e3
        seg = own_elapsed / args.steps * 1e3
        out["segmentation"] = {"segmented_ms_per_step": seg, "single_graph_ms_per_step": single, "overhead_frac": seg / single - 1.0,
                               "exchange_ms_per_step": exchange_ms,
                               "note": "one rank, RCCL group of one: [D fwd/bwd] -> all-reduce -> [D Adam, G fwd/bwd] -> factor all-gathers + "
                                       "tail all-reduce -> [G Adam] against the whole iteration as one graph without a process group"}
        del plain, G2, D2
        trainer = None
    del trainer, G, D, pool
    torch.cuda.empty_cache()
    return out


def precision_gap(pkg, args, device, mode="bf16x3"):
    """Relative error of a reduced-precision mode against the exact-fp32 mode on identical weights, reals, latents and epsilon: the three
    losses, the scores, the gradient penalty and the two nets' flat gradients (relative L2) of one iteration's forward/backward passes."""
    import torch
    out = {}
    got = {}
    for precision in ("f32", mode):
        pkg.ops.set_conv_precision(precision)
        G, D = build_nets(pkg, args.res, args.alpha, device)
        tr = pkg.train.PGGANTrainer(G, D, learning_rate=1e-4, beta1=0.5, grad_pen_lambda=10.0, drift_epsilon=0.001)
        gen = torch.Generator().manual_seed(4242)
        b = min(args.batch, 8)
        real = (torch.rand(b, 1, args.res, args.res, generator=gen) * 2 - 1).to(device)
        zs = []
        for _ in range(3):
            z = torch.randn(b, 512, generator=gen).clamp(-5, 5)
            zs.append((z / z.norm(dim=1, keepdim=True)).to(device))
        eps = torch.rand(b, 1, 1, 1, generator=gen).to(device)
        st = tr.d_compute(real, zs[0], zs[1], eps)
        gd = tr.flat_d.grad.clone()
        st.update(tr.g_compute(real, zs[2]))
        gg = tr.flat_g.grad.clone()
        torch.cuda.synchronize()
        got[precision] = ({k: float(v) for k, v in st.items()}, gd, gg)
        del tr, G, D
    a, b = got["f32"], got[mode]
    for k in a[0]:
        out[k] = abs(a[0][k] - b[0][k]) / max(abs(a[0][k]), 1e-12)
    out["D_grad_rel_l2"] = float((a[1] - b[1]).norm() / a[1].norm())
    out["G_grad_rel_l2"] = float((a[2] - b[2]).norm() / a[2].norm())
    return max(out.values()), out


def launch_check(world, rank):
    """`--launch-check`: the launcher path without a GPU -- a gloo group over the started ranks, one all-reduce, one JSON line."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "world_size_observed": dist.get_world_size(), "sum_of_ones": float(t.item()),
                          "backend": "gloo"}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--graph", type=int, default=-1, help="1: replay a captured HIP graph, 0: eager, -1: auto")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16"],
                    help="arithmetic of the headline: exact fp32 MFMA (the reference's arithmetic), split-bf16 (3 bf16 MFMAs per "
                         "product, fp32 storage and accumulate), or bf16 (bf16 activation storage, one bf16 MFMA per product)")
    ap.add_argument("--sub-record", type=int, default=-1, help="1: also time the reduced-precision modes and report them as the labelled "
                                                               "sub-records 'bf16x3' and 'bf16'; 0: do not; -1: only on one GPU")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the gradient exchange even with one rank (path test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="do not time the dominant kernel with HIP events")
    ap.add_argument("--live-traffic", type=int, default=-1, help="roofline.traffic of the headline's dominant kernel from two rocprofv3 counter "
                    "passes over a child run, inside this invocation (one GPU only); 0: use the committed summary under profiles/")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, all-reduce over gloo on the CPU, print one line (no GPU)")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        # the bare command: become the launcher.  Nothing GPU-related has happened in this process (torch is not imported).
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.launch_check:
        return launch_check(world, rank)

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    # rehearsal on a one-GPU box: NGAN_REHEARSAL_BACKEND=gloo runs every rank on cuda:0 with the collectives staged through the
    # host (same step driver, same segmented graph capture; the numbers mean nothing).  The real run is RCCL, one rank per GPU.
    rehearsal = os.environ.get("NGAN_REHEARSAL_BACKEND", "")
    if not rehearsal and world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()} "
                         f"(NGAN_REHEARSAL_BACKEND=gloo rehearses the multi-rank path on one GPU)")
    device = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(device)
    use_dist = world > 1 or args.force_dist
    backend = None
    if use_dist:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)          # `python bench.py --force-dist` without a launcher: a one-rank group
        if rehearsal:
            dist.init_process_group(rehearsal)
        else:
            dist.init_process_group("nccl", device_id=device)
        backend = dist.get_backend()

    pkg = load_package()
    pkg._C.lib()
    head = run_mode(pkg, args, args.precision, device, world, rank, use_dist)
    want_sub = args.sub_record == 1 or (args.sub_record == -1 and world == 1 and not args.force_dist)
    subs = {}
    if want_sub and args.precision == "f32":
        for mode in ("bf16x3", "bf16"):
            sub = run_mode(pkg, args, mode, device, world, rank, use_dist)
            subs[mode] = (sub, precision_gap(pkg, args, device, mode) if rank == 0 else None)
        pkg.ops.set_conv_precision("f32")

    if rank == 0:
        wm = pkg.workmodel
        images = args.batch * world * args.steps
        value = images / head["elapsed"]
        w_alg = wm.iteration_flops(G_WIDTHS, D_WIDTHS, 16, args.res, 512, args.alpha)   # FLOP per image per iteration
        e_alg = wm.iteration_io_elements(G_WIDTHS, D_WIDTHS, 16, args.res, 512, args.alpha)
        dtype_of = {"f32": "f32", "bf16x3": "f32 storage/accumulate; 3x3 convs on split-bf16 MFMA (bf16x3, hi+lo operands)",
                    "bf16": "bf16 activation storage and conv operands (one bf16 MFMA per product); f32 accumulate, PixelNorm statistics, "
                            "images, master weights, gradients of parameters, Adam"}
        esz_of = {"f32": 4, "bf16x3": 4, "bf16": 2}
        out = {"metric": "images/sec (G+D step incl. GP) at 512x512", "value": value, "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["elapsed"] / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": dtype_of[args.precision], "data": "synthetic",
               "config": {"workload": f"{args.res}x{args.res} stage, alpha={args.alpha}, batch {args.batch}/GPU, WGAN-GP lambda=10, "
                                      f"drift 0.001, n_critic=1, Adam(1e-4, 0.5, 0.999), widths G{G_WIDTHS} D{D_WIDTHS}",
                          "global_batch": args.batch * world, "resolution": args.res, "parallelism": f"dp{world}",
                          "launch": head["launch"], "conv_precision": args.precision,
                          "world_size_observed": dist.get_world_size() if use_dist else 1, "backend": backend},
               "step_tflops": value * w_alg / 1e12,
               "step_frac_of_fp32_mfma_peak": (value * w_alg / 1e12 / world / PEAK_FP32_MFMA_TFLOPS) if args.precision == "f32" else None,
               "step_algorithmic_gbs": value * e_alg * esz_of[args.precision] / 1e9 / world,
               "roofline": head["roofline"], "conv_family": head["conv_family"]}
        if head.get("ranks") is not None:
            out["ranks"] = head["ranks"]
        if head.get("segmentation") is not None:
            out["segmentation"] = head["segmentation"]
        labels = {"bf16x3": "split-bf16 convolution mode (NOT the headline: 16-bit-mantissa operands, narrower than the reference's fp32)",
                  "bf16": "bf16 mode = BASELINE.json's C2 arithmetic at this workload (NOT the headline: bf16 activation storage and conv operands; "
                          "its own tolerance, tests/test_gpu_bf16.py / DESIGN.md section 8; HBM-bound: bytes = E x 2)"}
        for mode, (sub, gap) in subs.items():
            v2 = images / sub["elapsed"]
            out[mode] = {"label": labels[mode], "value": v2, "unit": "images/s", "ms_per_step": sub["elapsed"] / args.steps * 1e3,
                         "dtype": dtype_of[mode], "max_rel_err": gap[0], "rel_err_vs_f32": gap[1],
                         "step_fp32_equivalent_tflops": v2 * w_alg / 1e12,
                         "step_algorithmic_gbs": v2 * e_alg * esz_of[mode] / 1e9,
                         "step_frac_of_hbm_peak": v2 * e_alg * esz_of[mode] / 1e9 / PEAK_HBM_GBS,
                         "roofline": sub["roofline"], "conv_family": sub["conv_family"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.res, args.alpha, batch=args.batch)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
